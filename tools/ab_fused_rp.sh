#!/bin/bash
# A/B on one box (round 6): PCG with the fused residual / direction launch (csrc/pcg_fused.inc) against the separate kernels, and the
# fused launch as a plain launch (default) against a cooperative one; first the bit-for-bit tests
timeout 900 python -m pytest tests/test_pcg_fused.py -m gpu -x -q 2>&1 | tail -5
for rep in 1 2 3; do
for V in "0 0" "1 0" "1 1"; do
  set -- $V
  echo "== MI_PCG_FUSE_RP=$1 MI_PCG_FUSE_COOP=$2"
  MI_PCG_FUSE_RP=$1 MI_PCG_FUSE_COOP=$2 MI_BENCH_NO_SUPPLEMENTS=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu 2>/tmp/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'amul_us', round(d['roofline']['avg_launch_us'],2), '|', d['config']['host_loop'][:90])" || tail -5 /tmp/ab_err.txt
done; done
