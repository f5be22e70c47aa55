#!/bin/bash
# A/B on one box: 1024-cell tiles (default) vs 512-cell tiles with 256-thread workgroups
mkdir -p gpurun_out
for r in 1 2; do
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu > gpurun_out/abt_1024_$r.json 2>> gpurun_out/abt.err
  MI_TILE_CELLS=512 MI_AMUL_BS=256 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu > gpurun_out/abt_512_$r.json 2>> gpurun_out/abt.err
done
timeout 400 python tools/bench_pbicg.py > gpurun_out/abt_pbicg_1024.log 2>&1
MI_TILE_CELLS=512 MI_AMUL_BS=256 timeout 400 python tools/bench_pbicg.py > gpurun_out/abt_pbicg_512.log 2>&1
MI_TILE_CELLS=512 timeout 400 python tools/bench_pbicg.py > gpurun_out/abt_pbicg_512auto.log 2>&1
timeout 400 python tools/bench_gamg.py > gpurun_out/abt_gamg_1024.log 2>&1
MI_TILE_CELLS=512 MI_AMUL_BS=256 timeout 400 python tools/bench_gamg.py > gpurun_out/abt_gamg_512.log 2>&1
for f in gpurun_out/abt_1024_1.json gpurun_out/abt_512_1.json gpurun_out/abt_1024_2.json gpurun_out/abt_512_2.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"],1), "it/s  amul us", round(d["roofline"]["avg_launch_us"],1))
PY
done
for f in gpurun_out/abt_pbicg_1024.log gpurun_out/abt_pbicg_512.log gpurun_out/abt_pbicg_512auto.log gpurun_out/abt_gamg_1024.log gpurun_out/abt_gamg_512.log; do echo "== $f"; grep -v "ROCm\|Hostname\|Librccl\|RCCL\|HIP version\|amdgpu.ids" $f | tail -8; done
