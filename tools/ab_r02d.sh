#!/bin/bash
# round 2 A/B (d): cache-policy choices at steady state (box warmed first by tools/exp/warm_probe.py)
cd "$(dirname "$0")/.."
timeout 300 python tools/exp/warm_probe.py > gpurun_out/r02d_warm.log 2>&1
out=gpurun_out/r02d_ab.jsonl; : > $out
run() { label=$1; shift
  line=$(env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu 2>> gpurun_out/r02d_ab.err)
  echo "{\"label\": \"$label\", \"line\": $line}" >> $out; }
V=$PWD/tools/exp
for rep in 1 2 3; do
run "base" A=1
run "flags1" MI_TILE_FLAGS=1
run "rb512_flags1" MI_ENGINE_LIB=$V/libvar_rb512_nt0_rd0.so MI_TILE_FLAGS=1
run "rb512_flags9" MI_ENGINE_LIB=$V/libvar_rb512_nt0_rd0.so MI_TILE_FLAGS=9
run "rb512_rdnt_flags1" MI_ENGINE_LIB=$V/libvar_rb512_nt0_rd1.so MI_TILE_FLAGS=1
run "rb512_rdnt_flags9" MI_ENGINE_LIB=$V/libvar_rb512_nt0_rd1.so MI_TILE_FLAGS=9
run "rb512_rdnt_flags11" MI_ENGINE_LIB=$V/libvar_rb512_nt0_rd1.so MI_TILE_FLAGS=11
run "rb512_rdnt_flags15" MI_ENGINE_LIB=$V/libvar_rb512_nt0_rd1.so MI_TILE_FLAGS=15
done
python - <<'PY'
import json
for l in open("gpurun_out/r02d_ab.jsonl"):
    d = json.loads(l); b = d["line"]
    print(f'{d["label"]:28s} {b["value"]:8.1f} it/s  {b["ms_per_step"]*1e3:7.1f} us/it  amul {b["roofline"]["avg_launch_us"]:6.1f} us  frac {b["roofline"]["frac"]:.3f}')
PY
head -3 gpurun_out/r02d_warm.log; tail -3 gpurun_out/r02d_warm.log
