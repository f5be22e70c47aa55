#!/bin/bash
# round 4, lease b: the reworked decomposed paths (window-reading tile ops, fused attached GAMG scale / residual, coarsest gather
# window, hipGraph of the attached V-cycle, batched PBiCG on attached matrices) + the spill-free persistent kernel
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
run() { name=$1; shift; timeout 1200 python -m pytest "$@" -x -q > $O/r04_b_$name.log 2>&1; echo "$name rc=$?" | tee -a $O/r04_b_$name.log; tail -n 3 $O/r04_b_$name.log | cut -c1-300; }
run parity tests/test_gpu_parity.py -k "persistent or fused_distributed or decomposed_solver_paths or attached_comm or distributed_matrix_single or native_rccl"
run full -s tests/test_gpu_full_size.py -k "persistent"
run dist tests/test_distributed.py -k "persistent or entirely_over_peer or one_shot or attached_solvers_on_several or transformed"
run gamg tests/test_gamg.py -k "coupled"
run configs tests/test_gpu_configs.py
timeout 300 python tools/bench_persist.py > $O/r04_b_persist.log 2>&1; tail -n 1 $O/r04_b_persist.log > $O/r04_b_persist_single_rank.json
timeout 300 python tools/bench_selfcomm.py --mode peer5,persist --out $O/r04_b_selfcomm_108.json > $O/r04_b_selfcomm.log 2>&1
timeout 900 python tools/bench_selfcomm_solvers.py --dims 108 108 108 --solver gamg,pbicg,timestep --out $O/r04_b_selfcomm_solvers_108.json > $O/r04_b_solvers108.log 2>&1
timeout 900 python tools/bench_selfcomm_solvers.py --dims 216 216 216 --solver gamg,pbicg,timestep --cycles 10 --iters 20 --steps 3 --out $O/r04_b_selfcomm_solvers_216.json > $O/r04_b_solvers216.log 2>&1
for f in $O/r04_b_persist.log $O/r04_b_selfcomm.log $O/r04_b_solvers108.log $O/r04_b_solvers216.log; do echo "== $f"; tail -n 2 $f | cut -c1-1500; done
