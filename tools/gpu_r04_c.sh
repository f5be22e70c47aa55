#!/bin/bash
# round 4, lease c: push fused into the tile kernel (one launch per attached tile operator); kernel tables of the attached V-cycle
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out; R=$(pwd)
run() { name=$1; shift; timeout 1200 python -m pytest "$@" -x -q > $O/r04_c_$name.log 2>&1; echo "$name rc=$?" | tee -a $O/r04_c_$name.log; tail -n 3 $O/r04_c_$name.log | cut -c1-300; }
run parity tests/test_gpu_parity.py -k "decomposed_solver_paths or attached_comm or distributed_matrix_single or persistent"
run dist tests/test_distributed.py -k "entirely_over_peer or transformed"
run gamg tests/test_gamg.py -k "coupled"
run bench tests/test_bench_contract.py -k "two_rank_rehearsal"
timeout 900 python tools/bench_selfcomm_solvers.py --dims 108 108 108 --solver gamg,pbicg,timestep --out $O/r04_c_selfcomm_solvers_108.json > $O/r04_c_solvers108.log 2>&1
timeout 900 python tools/bench_selfcomm_solvers.py --dims 216 216 216 --solver gamg,timestep --cycles 10 --steps 3 --out $O/r04_c_selfcomm_solvers_216.json > $O/r04_c_solvers216.log 2>&1
for d in 108 216; do
  (cd /tmp && MI_SELFCOMM_ONLY=attached timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04_c_prof_gamg_$d -o t -- python $R/tools/bench_selfcomm_solvers.py --dims $d $d $d --solver gamg --cycles 10 > $R/$O/r04_c_prof_gamg_$d.log 2>&1)
  find $O/r04_c_prof_gamg_$d -name "*.db" -delete
  f=$(find $O/r04_c_prof_gamg_$d -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && head -n 40 "$f" > $O/r04_c_kernel_stats_gamg_attached_$d.csv
done
for f in $O/r04_c_solvers108.log $O/r04_c_solvers216.log; do echo "== $f"; tail -n 1 $f | cut -c1-2500; done
head -n 25 $O/r04_c_kernel_stats_gamg_attached_108.csv | cut -c1-200
