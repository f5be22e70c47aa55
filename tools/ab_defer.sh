mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gamg.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_ref_dropin.py -m gpu -q -x > gpurun_out/pytest_defer.log 2>&1; grep -E "passed|failed|^FAILED|^E  " gpurun_out/pytest_defer.log | cut -c1-250 | head
for d in 1 0 1 0; do echo "== MI_PCG_DEFER_PSI=$d"; MI_PCG_DEFER_PSI=$d timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['repeat_ms_per_step'])"; done
