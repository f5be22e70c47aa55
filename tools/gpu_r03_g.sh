#!/bin/bash
# round 3, items 4(c) and 5: HBM traffic by PMC of the GAMG V-cycle (difference of a 25- and a 5-cycle solve) and of the assembly passes
export MI_GAMG_GRAPH=0
SKIP_TRACE=1 GAMG_CYCLES=5 bash tools/pmc_traffic.sh gamg5 tools/bench_gamg.py > /dev/null 2>&1
GAMG_CYCLES=25 bash tools/pmc_traffic.sh gamg25 tools/bench_gamg.py > /dev/null 2>&1
unset MI_GAMG_GRAPH
bash tools/pmc_traffic.sh assembly tools/bench_assembly.py > /dev/null 2>&1
cp gpurun_out/assembly_row_passes.json gpurun_out/pmc_assembly/ 2>/dev/null
tail -n 4 gpurun_out/pmc_gamg5/summary.md; tail -n 4 gpurun_out/pmc_gamg25/summary.md; head -n 16 gpurun_out/pmc_assembly/summary.md; tail -n 3 gpurun_out/pmc_gamg25/trace.log
