#!/bin/bash
# round 6, lease q: fused solver prologue -- its bit-for-bit test, the tests of the paths it touches, and the interleaved A/B of the two time steps
mkdir -p gpurun_out
export TMPDIR=/tmp
{ time timeout 900 python -m pytest tests/test_prologue_fused.py tests/test_gpu_parity.py tests/test_gamg.py -m gpu -q -x ; } > gpurun_out/r06_q_tests.log 2>&1
tail -5 gpurun_out/r06_q_tests.log
for rep in 1 2; do for f in 0 1; do
  echo "== MI_FUSE_PROLOGUE=$f"
  MI_FUSE_PROLOGUE=$f STEPS=4 timeout 300 python tools/bench_timestep.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('piso ms', round(d['ms_per_time_step'],3), {k[:40]: round(v,3) for k,v in d['stages_ms'].items()})"
  MI_FUSE_PROLOGUE=$f STEPS=3 timeout 300 python tools/bench_rhopimple.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('rhopimple ms', round(d['ms_per_time_step'],3), {k[:30]: round(v,3) for k,v in d['stages_ms'].items()})"
done; done 2>&1 | tee gpurun_out/r06_q_ab.txt
