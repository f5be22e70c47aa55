mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_ami.py tests/test_ref_dropin.py tests/test_foam_mirror.py -m gpu -q -x > gpurun_out/pytest_fuse.log 2>&1; grep -E "passed|failed|^FAILED|^E  " gpurun_out/pytest_fuse.log | cut -c1-250 | head
for f in 1 0; do echo "== MI_FUSE_PERM=$f"; MI_FUSE_PERM=$f timeout 300 python tools/bench_shuffled.py 2>&1 | grep -E "^lexico|^shuffled" | python -c "
import sys,ast
for l in sys.stdin:
    tag,rest=l.split(' ',1); d=ast.literal_eval(rest); print(tag, 'engine', d['amul_engine_us'], 'caller', d['amul_caller_us'])"; done
MI_FUSE_PERM=1 DIMS=216,216,216 timeout 300 python tools/bench_shuffled.py 2>&1 | grep -E "^lexico" | cut -c1-400
MI_FUSE_PERM=0 DIMS=216,216,216 timeout 300 python tools/bench_shuffled.py 2>&1 | grep -E "^lexico" | cut -c1-400
