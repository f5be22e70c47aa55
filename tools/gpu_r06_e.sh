#!/bin/bash
# round 6: fused GAMG transfers (MI_GAMG_FUSE) against the separate kernels, interleaved on ONE box, 216^3 and the 108^3 share of the 8-GPU run
mkdir -p gpurun_out
out=gpurun_out/r06_e_gamg_fuse_ab.txt
: > $out
python -m pytest tests/test_gamg.py -m gpu -x -q -k "fused or history or replay" 2>&1 | tail -2 >> $out
for rep in 1 2 3; do
  for dims in 216,216,216 108,108,108; do
    for fuse in 1 0; do
      echo "dims $dims MI_GAMG_FUSE=$fuse rep $rep: $(GAMG_DIMS=$dims GAMG_CYCLES=40 MI_GAMG_FUSE=$fuse python tools/bench_gamg.py 2>/dev/null | grep ms_per_cycle | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_cycle"])')" >> $out
    done
  done
done
cat $out
