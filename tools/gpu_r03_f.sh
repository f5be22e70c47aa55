#!/bin/bash
# round 3, item 1: the small-sub-domain PCG iteration -- RCCL phase loop / five launches over peer windows / one persistent kernel
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python tools/bench_selfcomm.py --mode rccl,peer5,persist --out $O/selfcomm_108.json > $O/selfcomm.log 2>&1; echo "selfcomm rc=$?"
timeout 300 python tools/bench_persist.py > $O/persist_single.log 2>&1; echo "persist rc=$?"; tail -n 1 $O/persist_single.log > $O/persist_single.json
cd /tmp
for M in peer5 persist; do
  rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$M -o t -- python $R/tools/bench_selfcomm.py --mode $M --iters 200 > $R/$O/trace_$M.log 2>&1
  python $R/tools/prof_db.py $R/$O/trace_$M 12 > $R/$O/kernels_$M.md 2>> $R/$O/trace_$M.log
done
cd $R
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
grep -E "^(rccl|peer5|persist)" $O/selfcomm.log | cut -c1-160; grep "^(" $O/persist_single.log | cut -c1-200; cat $O/kernels_persist.md | head -12; cat $O/kernels_peer5.md | head -12
