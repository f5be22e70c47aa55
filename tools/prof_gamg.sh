#!/bin/bash
# rocprofv3 kernel trace of the GAMG solve on the 216^3 box (BASELINE config 3) -> gpurun_out/prof_gamg/, summary by tools/summarize_gamg_prof.py
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_gamg
cd /tmp
GAMG_CYCLES=${GAMG_CYCLES:-10} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_gamg/trace -o gamg -- python $R/tools/bench_gamg.py > $R/gpurun_out/prof_gamg/bench_gamg.log 2> $R/gpurun_out/prof_gamg/trace.err
cd $R
find gpurun_out/prof_gamg -name "*.db" -delete
python tools/summarize_gamg_prof.py gpurun_out/prof_gamg > gpurun_out/prof_gamg/summary.md 2>&1
tail -60 gpurun_out/prof_gamg/summary.md
