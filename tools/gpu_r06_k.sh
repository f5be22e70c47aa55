#!/bin/bash
# round 6, lease k: HBM traffic by PMC of the GAMG V-cycle alone (difference of a 25- and a 5-cycle solve; tools/summarize_gamg_traffic.py --update)
export MI_GAMG_GRAPH=0
SKIP_TRACE=1 GAMG_CYCLES=5 bash tools/pmc_traffic.sh gamg5 tools/bench_gamg.py > /dev/null 2>&1
SKIP_TRACE=1 GAMG_CYCLES=25 bash tools/pmc_traffic.sh gamg25 tools/bench_gamg.py > /dev/null 2>&1
tail -n 3 gpurun_out/pmc_gamg5/summary.md; tail -n 3 gpurun_out/pmc_gamg25/summary.md
