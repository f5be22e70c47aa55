#!/bin/bash
# round 2 A/B on one box: vector-kernel block size / non-temporal policy x tile configuration, bench.py in-loop numbers
cd "$(dirname "$0")/.."
out=gpurun_out/r02c_ab.jsonl; : > $out
run() { # label, env...
  label=$1; shift
  line=$(env "$@" timeout 200 python bench.py --steps 300 --warmup 20 --no-cpu 2>> gpurun_out/r02c_ab.err)
  echo "{\"label\": \"$label\", \"line\": $line}" >> $out
}
for rep in 1 2; do
run "base" A=1
run "flags1" MI_TILE_FLAGS=1
run "flags3" MI_TILE_FLAGS=3
run "t512_bs256" MI_TILE_CELLS=512 MI_AMUL_BS=256
run "t512_bs256_flags3" MI_TILE_CELLS=512 MI_AMUL_BS=256 MI_TILE_FLAGS=3
run "t512_bs256_c16" MI_TILE_CELLS=512 MI_AMUL_BS=256 MI_ENTRY16=1
run "t512_bs256_c16_flags3" MI_TILE_CELLS=512 MI_AMUL_BS=256 MI_ENTRY16=1 MI_TILE_FLAGS=3
run "rb512" MI_ENGINE_LIB=$PWD/tools/exp/libvar_rb512_nt0.so
run "rb256_nt" MI_ENGINE_LIB=$PWD/tools/exp/libvar_rb256_nt1.so
run "rb512_nt" MI_ENGINE_LIB=$PWD/tools/exp/libvar_rb512_nt1.so
run "rb512_nt_flags3" MI_ENGINE_LIB=$PWD/tools/exp/libvar_rb512_nt1.so MI_TILE_FLAGS=3
run "rb512_nt_t512_c16_flags3" MI_ENGINE_LIB=$PWD/tools/exp/libvar_rb512_nt1.so MI_TILE_CELLS=512 MI_AMUL_BS=256 MI_ENTRY16=1 MI_TILE_FLAGS=3
run "rb512_t512_c16_flags3" MI_ENGINE_LIB=$PWD/tools/exp/libvar_rb512_nt0.so MI_TILE_CELLS=512 MI_AMUL_BS=256 MI_ENTRY16=1 MI_TILE_FLAGS=3
done
python - <<'PY'
import json
for l in open("gpurun_out/r02c_ab.jsonl"):
    d = json.loads(l); b = d["line"]
    print(f'{d["label"]:28s} {b["value"]:8.1f} it/s  {b["ms_per_step"]*1e3:7.1f} us/it  amul {b["roofline"]["avg_launch_us"]:6.1f} us  frac {b["roofline"]["frac"]:.3f}')
PY
