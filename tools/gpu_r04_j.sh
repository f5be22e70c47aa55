#!/bin/bash
# round 4, lease j: coarsest inverse on a 16 x 16 thread grid (bit-equality with the host elimination, time per step)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out; R=$(pwd)
timeout 1200 python -m pytest tests/test_gamg.py -x -q > $O/r04_j_gamg.log 2>&1; echo "gamg rc=$?" | tee -a $O/r04_j_gamg.log; tail -n 3 $O/r04_j_gamg.log | cut -c1-300
timeout 900 python tools/bench_selfcomm_solvers.py --dims 108 108 108 --solver timestep --out $O/r04_j_timestep_108.json > $O/r04_j_ts108.log 2>&1
timeout 900 python tools/bench_selfcomm_solvers.py --dims 216 216 216 --solver timestep --steps 3 --out $O/r04_j_timestep_216.json > $O/r04_j_ts216.log 2>&1
(cd /tmp && MI_SELFCOMM_ONLY=local timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04_j_prof_ts108 -o t -- python $R/tools/bench_selfcomm_solvers.py --dims 108 108 108 --solver timestep --steps 4 > $R/$O/r04_j_prof_ts108.log 2>&1)
find $O/r04_j_prof_ts108 -name "*.db" -delete; rm -f $O/r04_j_prof_ts108/t_kernel_trace.csv
for f in $O/r04_j_ts108.log $O/r04_j_ts216.log; do tail -n 1 $f | cut -c1-2500; done
grep -E "dense_invert" $O/r04_j_prof_ts108/t_kernel_stats.csv | cut -c1-200
