#!/bin/bash
# round 4 closing evidence: bench.py under rocprofv3 (kernel stats + PMC passes), GAMG cycle traffic by PMC, a rocprof kernel
# table of the PBiCG solvers, the plain bench lines of both modes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r04h && export TMPDIR=/tmp
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
R=$GRAFT_REPO_ROOT
MI_BENCH_NO_SUPPLEMENTS=1 bash tools/prof_round.sh > gpurun_out/r04h/prof_round.log 2>&1
export MI_GAMG_GRAPH=0
SKIP_TRACE=1 GAMG_CYCLES=5 bash tools/pmc_traffic.sh gamg5 tools/bench_gamg.py > /dev/null 2>&1
SKIP_TRACE=1 GAMG_CYCLES=25 bash tools/pmc_traffic.sh gamg25 tools/bench_gamg.py > /dev/null 2>&1
unset MI_GAMG_GRAPH
(cd /tmp && DIMS=216,216,216 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04h/prof_pbicg -o t -- python $R/tools/bench_pbicg.py > $R/gpurun_out/r04h/bench_pbicg.log 2>&1)
find gpurun_out/r04h -name "*.db" -delete; rm -f gpurun_out/r04h/prof_pbicg/t_kernel_trace.csv
{ time timeout 900 python bench.py ; } > gpurun_out/r04h/bench.json 2> gpurun_out/r04h/bench.err
{ time timeout 400 python bench.py --solver gamg ; } > gpurun_out/r04h/bench_gamg.json 2> gpurun_out/r04h/bench_gamg.err
cut -c1-1200 gpurun_out/r04h/bench.json; cut -c1-400 gpurun_out/r04h/bench_gamg.json; tail -n 3 gpurun_out/r04h/bench.err; tail -n 2 gpurun_out/r04h/bench_pbicg.log | cut -c1-600
