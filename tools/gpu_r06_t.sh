#!/bin/bash
# round 6, lease t: evidence for the tree with the one-pass solver prologue and the new coarsest inversion -- the whole GPU suite with
# durations, smoke, the bench line, rocprofv3 kernel stats + PMC passes of bench.py (tools/prof_round.sh -> tools/summarize_prof.py r06_t),
# the GAMG cycle's kernel trace (tools/prof_gamg.sh) and its HBM traffic by PMC (tools/gpu_r03_g.sh -> tools/summarize_gamg_traffic.py --update)
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/test_durations.tsv gpurun_out/native_solve_timings.tsv
t0=$(date +%s)
{ time timeout 1150 python -m pytest tests -m gpu -q --durations=30 ; } > gpurun_out/r06_t_suite.log 2>&1
echo "pytest exit $?, $(( $(date +%s) - t0 )) s wall" >> gpurun_out/r06_t_suite.log
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r06_t_smoke.log 2>&1
{ time timeout 600 python bench.py ; } > gpurun_out/r06_t_bench.json 2> gpurun_out/r06_t_bench.err
bash tools/prof_round.sh > gpurun_out/r06_t_prof_round.log 2>&1
bash tools/prof_gamg.sh > /dev/null 2>&1; cp gpurun_out/prof_gamg/summary.md gpurun_out/r06_t_gamg_rocprof_summary.md
bash tools/gpu_r03_g.sh > gpurun_out/r06_t_pmc.log 2>&1
rm -rf gpurun_out/prof_gamg/trace gpurun_out/pmc_gamg5/trace gpurun_out/pmc_gamg25/trace
grep -E "passed|failed|pytest exit" gpurun_out/r06_t_suite.log | tail -3; tail -2 gpurun_out/r06_t_smoke.log; cut -c1-600 gpurun_out/r06_t_bench.json; du -sh gpurun_out
