#!/bin/bash
# round 4, lease a: persistent-kernel parity at its design point + hardening tests, spill-free kernel timings,
# and the BASELINE of the attached GAMG / PBiCG paths (self-exchange) before they are reworked
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent or fused_distributed" > $O/r04_a_tests_parity.log 2>&1; echo "parity rc=$?" | tee -a $O/r04_a_tests_parity.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -s -k "persistent" > $O/r04_a_tests_full.log 2>&1; echo "full rc=$?" | tee -a $O/r04_a_tests_full.log
timeout 900 python -m pytest tests/test_distributed.py -x -q -k "persistent" > $O/r04_a_tests_dist.log 2>&1; echo "dist rc=$?" | tee -a $O/r04_a_tests_dist.log
timeout 300 python tools/bench_persist.py > $O/r04_a_persist.log 2>&1; tail -1 $O/r04_a_persist.log > $O/r04_a_persist_single_rank.json
timeout 300 python tools/bench_selfcomm.py --mode peer5,persist --out $O/r04_a_selfcomm_108.json > $O/r04_a_selfcomm.log 2>&1
timeout 600 python tools/bench_selfcomm_solvers.py --dims 108 108 108 --solver gamg,pbicg --out $O/r04_a_selfcomm_solvers_108_baseline.json > $O/r04_a_solvers108.log 2>&1
timeout 600 python tools/bench_selfcomm_solvers.py --dims 216 216 216 --solver gamg --cycles 10 --out $O/r04_a_selfcomm_solvers_216_baseline.json > $O/r04_a_solvers216.log 2>&1
tail -3 $O/r04_a_tests_*.log; tail -2 $O/r04_a_persist.log $O/r04_a_selfcomm.log $O/r04_a_solvers108.log $O/r04_a_solvers216.log
