#!/bin/bash
# round 5, closing lease: tools/final_round.sh (GPU suite inside the window, smoke, both bench modes), then the V-cycle evidence with
# the default level layouts (per-level kernel table, PMC traffic as the difference of a 25- and a 5-cycle solve) and the assembly
# passes un-profiled
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r05d && export TMPDIR=/tmp
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
SUITE_LIMIT=900 bash tools/final_round.sh > gpurun_out/r05d/final_round.log 2>&1
echo "final_round exit $?" >> gpurun_out/r05d/final_round.log
bash tools/prof_gamg.sh > gpurun_out/r05d/prof_gamg.log 2>&1
export MI_GAMG_GRAPH=0
SKIP_TRACE=1 GAMG_CYCLES=5 bash tools/pmc_traffic.sh gamg5 tools/bench_gamg.py > /dev/null 2>&1
SKIP_TRACE=1 GAMG_CYCLES=25 bash tools/pmc_traffic.sh gamg25 tools/bench_gamg.py > /dev/null 2>&1
unset MI_GAMG_GRAPH
timeout 300 python tools/bench_assembly.py > gpurun_out/r05d/bench_assembly.log 2>&1
cp gpurun_out/assembly_row_passes.json gpurun_out/r05d/
timeout 200 python tools/bench_pbicg.py > gpurun_out/r05d/bench_pbicg.json 2> gpurun_out/r05d/bench_pbicg.err
find gpurun_out -name "*.db" -delete
tail -12 gpurun_out/r05d/final_round.log | cut -c1-900
sort -rn gpurun_out/test_durations.tsv | head -25
tail -n 4 gpurun_out/pmc_gamg25/summary.md
