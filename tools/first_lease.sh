#!/bin/bash
# First multi-GPU lease: turn it into a DECISION TABLE instead of one number.  Nothing in this repository
# has ever run between two devices (the build box and every gpurun lease have one GPU): this script runs, in order of cost,
#   1. the set-up self-tests between devices: mi_comm_peer_auto (all-reduce windows over hipIpc, fine-grained coherence),
#      halo windows of an attached matrix, the grid-barrier litmus of the persistent kernel, the GAMG gather window;
#   2. the GPU tests that skip on a one-GPU box (RCCL with one device per rank, 2 and 4 ranks), and the window tests of
#      tests/test_distributed.py with one device per rank instead of ranks sharing device 0;
#   3. bench.py --gpus N for N in 2 4 8, once per FORCED path -- persistent kernel over windows / five launches over windows /
#      RCCL phase loop -- each with the path that really ran, the fall-back reason and the wait time-outs in its JSON line.
# Usage (on a node with >= 2 MI355X, from the repository root):   bash tools/first_lease.sh [max_gpus]
# Output: gpurun_out/first_lease/{selftests.log, tests_rccl.log, tests_windows_per_device.log, bench_<N>_<path>.json, table.md}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/first_lease
mkdir -p $O
NG=$(python -c "import torch; print(torch.cuda.device_count())")
MAX=${1:-$NG}
[ "$MAX" -gt "$NG" ] && MAX=$NG
echo "[first_lease] $NG device(s) visible, using up to $MAX" | tee $O/selftests.log
if [ "$MAX" -lt 2 ]; then echo "[first_lease] needs at least 2 GPUs" | tee -a $O/selftests.log; exit 2; fi

# ---- 1. self-tests between devices (tools/first_lease_selftest.py: one rank per device over RCCL)
for N in 2 4 8; do
  [ "$N" -le "$MAX" ] || continue
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      tools/first_lease_selftest.py >> $O/selftests.log 2>&1
  echo "[first_lease] self-tests on $N ranks: rc=$?" | tee -a $O/selftests.log
done

# ---- 2. the tests that skip on one GPU
timeout 1800 python -m pytest tests/test_distributed.py -q -k "over_rccl_one_device_per_rank" > $O/tests_rccl.log 2>&1
echo "[first_lease] RCCL one-device-per-rank tests: rc=$? ($(tail -n 1 $O/tests_rccl.log))" | tee -a $O/selftests.log

# ---- 2b. the window tests with ONE DEVICE PER RANK (host transport gloo, halo / all-reduce / gather windows across xGMI): every
# solver over windows, neighbours in different window forms, cyclicAMI across ranks, the persistent distributed kernel
MI_TEST_DEVICE_PER_RANK=1 timeout 2400 python -m pytest tests/test_distributed.py -q \
    -k "entirely_over_peer_windows or different_window_forms or cyclic_ami_whose or cyclic_ami_side_split or persistent_distributed or transformed_processor" > $O/tests_windows_per_device.log 2>&1
echo "[first_lease] window tests, one device per rank: rc=$? ($(tail -n 1 $O/tests_windows_per_device.log))" | tee -a $O/selftests.log

# ---- 3. bench.py per forced path
run_bench() {  # N path env...
  local N=$1 P=$2; shift 2
  env "$@" timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700 + N)) \
      bench.py --gpus $N --steps 200 --warmup 10 > $O/bench_${N}_${P}.json 2> $O/bench_${N}_${P}.err
  echo "[first_lease] bench --gpus $N path=$P rc=$?" | tee -a $O/selftests.log
}
for N in 2 4 8; do
  [ "$N" -le "$MAX" ] || continue
  run_bench $N auto        MI_ALLREDUCE=auto                                   # what the driver's SCALE run takes
  run_bench $N windows5    MI_ALLREDUCE=auto MI_PCG_PERSIST=0                  # five launches over windows
  run_bench $N rccl        MI_ALLREDUCE=rccl                                   # phase loop over RCCL
  run_bench $N pull        MI_ALLREDUCE=auto MI_WIN_DIRECT=0 MI_GAMG_GRAPH_ATTACHED=0 MI_PCG_PERSIST=0   # round-3 forms of the generic operators
done
python tools/first_lease_table.py $O > $O/table.md 2>> $O/selftests.log
cat $O/table.md
