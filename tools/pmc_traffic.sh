#!/bin/bash
# HBM traffic per kernel of any workload script: kernel trace + FETCH_SIZE + WRITE_SIZE in SEPARATE rocprofv3 passes
# (MI355X_MICROARCH.md, HBM section), summarised by tools/summarize_traffic.py.
#   bash tools/pmc_traffic.sh <tag> <script.py> [args...]      -> gpurun_out/pmc_<tag>/{trace,FETCH_SIZE,WRITE_SIZE}/, summary.md
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp
[ -n "$SKIP_TRACE" ] || rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/"$@" > $O/trace.log 2> $O/trace.err
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o t -- python $R/"$@" > $O/$C.log 2> $O/$C.err
done
cd $R
find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
python tools/summarize_traffic.py $O > $O/summary.md 2>&1
head -40 $O/summary.md
