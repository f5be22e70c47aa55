#!/bin/bash
# round 4, lease r: regression subsets behind the threaded host builds (tile layout, GAMG hierarchy creation with level tables built by the level tasks)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
run() { name=$1; shift; timeout 170 python -m pytest "$@" -q -x > $O/r04_r_$name.log 2>&1; echo "$name rc=$? $(grep -E 'passed|failed' $O/r04_r_$name.log | tail -n 1 | cut -c1-160)"; }
run gamg tests/test_gamg.py -m gpu
run parity tests/test_gpu_parity.py -k "gamg or ordered or adopted or amul or layout or decomposed_solver_paths or renumber"
run ami tests/test_ami.py -m gpu
grep -E "FAILED" $O/r04_r_*.log | head -n 10 | cut -c1-300
