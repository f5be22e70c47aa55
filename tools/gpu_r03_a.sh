#!/bin/bash
# round 3, GPU call A: the peer-window path (tests + the 108^3 self-communicator measurement) and a headline sanity run
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused or halo_windows or self_exchange or attached_comm or distributed_matrix" > gpurun_out/r03a/t_parity.log 2>&1; echo "parity rc=$?"
timeout 900 python -m pytest tests/test_distributed.py -x -q -m gpu -k "peer or several_engine_ranks" > gpurun_out/r03a/t_dist.log 2>&1; echo "dist rc=$?"
timeout 600 python tools/bench_selfcomm.py --out gpurun_out/r03a/selfcomm_108.json > gpurun_out/r03a/selfcomm.log 2>&1; echo "selfcomm rc=$?"
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r03a/prof_selfcomm -- python $GRAFT_REPO_ROOT/tools/bench_selfcomm.py --mode peer4 --iters 200 > $GRAFT_REPO_ROOT/gpurun_out/r03a/prof_selfcomm.log 2>&1; echo "prof rc=$?"; cd $GRAFT_REPO_ROOT
for f in t_parity t_dist; do tail -n 4 gpurun_out/r03a/$f.log; done; tail -n 4 gpurun_out/r03a/selfcomm.log
find gpurun_out/r03a/prof_selfcomm -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {}
