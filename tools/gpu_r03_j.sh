#!/bin/bash
# round-3 closing evidence: bench.py under rocprofv3 (kernel stats + PMC passes), the plain bench lines of both modes
mkdir -p gpurun_out/r03j
MI_BENCH_NO_SUPPLEMENTS=1 bash tools/prof_round.sh > gpurun_out/r03j/prof_round.log 2>&1
{ time timeout 900 python bench.py ; } > gpurun_out/r03j/bench.json 2> gpurun_out/r03j/bench.err
{ time timeout 400 python bench.py --solver gamg ; } > gpurun_out/r03j/bench_gamg.json 2> gpurun_out/r03j/bench_gamg.err
cut -c1-900 gpurun_out/r03j/bench.json; cut -c1-400 gpurun_out/r03j/bench_gamg.json; tail -n 3 gpurun_out/r03j/bench.err
