#!/bin/bash
# A/B on one box: explicit 32-bit row entries (default) vs compact 16-bit entries (MI_ENTRY16=1), interleaved
# (when profiles/r01_n was measured the compact form was the default and MI_ENTRY32=1 selected the explicit one)
mkdir -p gpurun_out
{ timeout 900 python -m pytest tests -m gpu -x -q ; } > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
{ MI_ENTRY16=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gamg.py -m gpu -x -q ; } > gpurun_out/pytest_gpu_e16.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu_e16.log
for r in 1 2; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu > gpurun_out/ab_e32_$r.json 2>> gpurun_out/ab.err
  MI_ENTRY16=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu > gpurun_out/ab_e16_$r.json 2>> gpurun_out/ab.err
done
timeout 600 python tools/bench_kernels.py > gpurun_out/kernels_e16.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu_e16.log
for f in gpurun_out/ab_e32_1.json gpurun_out/ab_e16_1.json gpurun_out/ab_e32_2.json gpurun_out/ab_e16_2.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"])
PY
done
tail -5 gpurun_out/ab.err
