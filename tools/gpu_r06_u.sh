#!/bin/bash
# round 6, lease u: GAMG tests on the tree without the four-wavefront inversion variant, and the V-cycle's HBM traffic by PMC for these sources
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_tests.sh r06_u tests/test_gamg.py tests/test_dense_invert.py tests/test_prologue_fused.py -x
export MI_GAMG_GRAPH=0
SKIP_TRACE=1 GAMG_CYCLES=5 bash tools/pmc_traffic.sh gamg5 tools/bench_gamg.py > /dev/null 2>&1
SKIP_TRACE=1 GAMG_CYCLES=25 bash tools/pmc_traffic.sh gamg25 tools/bench_gamg.py > /dev/null 2>&1
unset MI_GAMG_GRAPH
tail -n 3 gpurun_out/pmc_gamg25/summary.md
