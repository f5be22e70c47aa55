#!/bin/bash
# round 5, lease c (evidence): inherited GAMG tiles validated on the device, bench.py under rocprofv3 (kernel stats + PMC passes),
# the GAMG cycle (per-level kernel table + PMC traffic), the PBiCG solvers' kernel table, the assembly passes (timing + PMC)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r05c && export TMPDIR=/tmp
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05c
# (0) where the multi-process tests spend their time: five of them with pooled rank processes (explicit clean-up after every job)
#     and with fresh processes per job, solve by solve (gpurun_out/native_solve_timings.tsv)
rm -f gpurun_out/native_solve_timings.tsv gpurun_out/test_durations.tsv
K='(test_native_attached_solvers_on_several_engine_ranks and box_2) or (one_shot_peer_allreduce and box_2) or (entirely_over_peer_windows and box_2) or (cyclic_ami and 2-False)'
echo "== pooled" >> gpurun_out/native_solve_timings.tsv
{ time timeout 420 python -m pytest tests/test_distributed.py -m gpu -q -k "$K" ; } > $O/dist_pooled.log 2>&1
echo "== fresh processes per job" >> gpurun_out/native_solve_timings.tsv
{ time MI_TEST_POOL=0 timeout 300 python -m pytest tests/test_distributed.py -m gpu -q -k "$K" ; } > $O/dist_fresh.log 2>&1
cp gpurun_out/test_durations.tsv $O/dist_durations.tsv
# (1) GAMG level layouts from inherited tiles: the device tests with the option on, start-up time both ways
{ time MI_GAMG_INHERIT_TILES=1 timeout 300 python -m pytest tests/test_gamg.py tests/test_ami.py -m gpu -q -x ; } > $O/inherit_tests.log 2>&1
echo "inherit exit $?" >> $O/inherit_tests.log
for v in 0 1; do MI_GAMG_INHERIT_TILES=$v timeout 200 python tools/bench_gamg.py > $O/bench_gamg_inherit$v.log 2>&1; done
# everything below runs with inherited tiles when the device tests passed with them (the default is then flipped in the source)
if grep -q "inherit exit 0" $O/inherit_tests.log; then export MI_GAMG_INHERIT_TILES=1; echo "MI_GAMG_INHERIT_TILES=1 for the measurements" > $O/decision.txt; else echo "inherited tiles NOT used" > $O/decision.txt; fi
# (2) bench.py under the profiler
MI_BENCH_NO_SUPPLEMENTS=1 bash tools/prof_round.sh > $O/prof_round.log 2>&1
# (3) the V-cycle: per-level kernel table, then traffic by PMC (difference of a 25- and a 5-cycle solve)
bash tools/prof_gamg.sh > $O/prof_gamg.log 2>&1
export MI_GAMG_GRAPH=0
SKIP_TRACE=1 GAMG_CYCLES=5 bash tools/pmc_traffic.sh gamg5 tools/bench_gamg.py > /dev/null 2>&1
SKIP_TRACE=1 GAMG_CYCLES=25 bash tools/pmc_traffic.sh gamg25 tools/bench_gamg.py > /dev/null 2>&1
unset MI_GAMG_GRAPH
# (4) the PBiCG solvers
(cd /tmp && DIMS=216,216,216 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_pbicg -o t -- python $R/tools/bench_pbicg.py > $R/$O/bench_pbicg.log 2>&1)
# (5) the assembly passes: timing, then PMC
bash tools/pmc_traffic.sh assembly tools/bench_assembly.py > /dev/null 2>&1
cp gpurun_out/assembly_row_passes.json gpurun_out/pmc_assembly/ 2>/dev/null
find gpurun_out -name "*.db" -delete; rm -f $O/prof_pbicg/t_kernel_trace.csv
cat gpurun_out/native_solve_timings.tsv; tail -3 $O/dist_pooled.log; tail -3 $O/dist_fresh.log
tail -3 $O/inherit_tests.log; tail -2 $O/bench_gamg_inherit0.log | cut -c1-300; tail -2 $O/bench_gamg_inherit1.log | cut -c1-300
tail -n 4 gpurun_out/pmc_gamg25/summary.md; head -n 20 gpurun_out/pmc_assembly/summary.md; tail -n 2 $O/bench_pbicg.log | cut -c1-600
