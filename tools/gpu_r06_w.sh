#!/bin/bash
# round 6, lease w: the bottom of the V-cycle's way down as one launch (k_gamg_restrict_chain): tests, interleaved A/B of the cycle at 216^3 and 108^3
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_tests.sh r06_w tests/test_gamg.py tests/test_gpu_configs.py -x
for rep in 1 2 3; do for f in 0 1; do for d in 216,216,216 108,108,108; do
  echo "== MI_GAMG_CHAIN=$f DIMS=$d"
  MI_GAMG_CHAIN=$f GAMG_DIMS=$d GAMG_CYCLES=20 timeout 300 python tools/bench_gamg.py 2>&1 | grep -E "ms_per_cycle" | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['ms_per_cycle'],4), d['history'][:3])"
done; done; done 2>&1 | tee gpurun_out/r06_w_chain_ab.txt
