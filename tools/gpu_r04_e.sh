#!/bin/bash
# round 4, lease e: self-validating pairs in the halo window of the one-launch tile operators
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out; R=$(pwd)
run() { name=$1; shift; timeout 1200 python -m pytest "$@" -x -q > $O/r04_e_$name.log 2>&1; echo "$name rc=$?" | tee -a $O/r04_e_$name.log; tail -n 3 $O/r04_e_$name.log | cut -c1-300; }
run gamg tests/test_gamg.py -k "coupled or straight"
run parity tests/test_gpu_parity.py -k "decomposed_solver_paths or attached_comm or distributed_matrix_single or fused_distributed or persistent_distributed"
run dist tests/test_distributed.py -k "entirely_over_peer"
timeout 900 python tools/bench_selfcomm_solvers.py --dims 108 108 108 --solver gamg,timestep --out $O/r04_e_selfcomm_solvers_108.json > $O/r04_e_solvers108.log 2>&1
timeout 900 python tools/bench_selfcomm_solvers.py --dims 216 216 216 --solver gamg,timestep --cycles 10 --steps 3 --out $O/r04_e_selfcomm_solvers_216.json > $O/r04_e_solvers216.log 2>&1
for d in 108; do
  (cd /tmp && MI_SELFCOMM_ONLY=attached timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04_e_prof_gamg_$d -o t -- python $R/tools/bench_selfcomm_solvers.py --dims $d $d $d --solver gamg --cycles 10 > $R/$O/r04_e_prof_gamg_$d.log 2>&1)
  find $O/r04_e_prof_gamg_$d -name "*.db" -delete; rm -f $O/r04_e_prof_gamg_$d/t_kernel_trace.csv
done
for f in $O/r04_e_solvers108.log $O/r04_e_solvers216.log $O/r04_e_solvers108_pull.log; do echo "== $f"; tail -n 1 $f | cut -c1-3000; done
head -n 12 $O/r04_e_prof_gamg_108/t_kernel_stats.csv | cut -c1-200
