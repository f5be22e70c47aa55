#!/bin/bash
# tile-cap sweep of the headline (the tile count against the 1 024 resident workgroup slots: 9 842 tiles = 9.6 dispatch waves at the default cap)
export TMPDIR=/tmp MI_BENCH_NO_SUPPLEMENTS=1
for rep in 1 2; do for tc in 1024 992 960 928 896; do
  MI_TILE_CELLS=$tc timeout 300 python bench.py --no-cpu 2>gpurun_out/tc.err | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('MI_TILE_CELLS=$tc', round(d['value'],1), 'it/s  amul', round(d['roofline']['avg_launch_us'],2), 'us')"
  grep -o "'tiles': [0-9]*" gpurun_out/tc.err | head -1
done; done
