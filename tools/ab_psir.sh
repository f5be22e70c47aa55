#!/bin/bash
# A/B: k_pcg_update_psi_r compiled for 8 waves per SIMD (<= 64 VGPRs: four 512-thread blocks per CU) against the default (74 VGPRs, three)
set -e
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread"
S="rapidcfd-dev_amd/csrc/engine.hip rapidcfd-dev_amd/csrc/tiling.cpp rapidcfd-dev_amd/csrc/gamg.cpp"
/opt/rocm/bin/hipcc $FL -DMI_PSIR_WAVES=8 -DMI_PSIR_UNROLL2 $S -o /tmp/lib_psir8.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A6 "k_pcg_update_psi_rILi1ELb0" | grep -E "VGPRs:|Spill|Scratch" | head -4
for rep in 1 2; do
for L in "" /tmp/lib_psir8.so; do
  echo "== lib ${L:-default}"
  MI_ENGINE_LIB=$L MI_BENCH_NO_SUPPLEMENTS=1 timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('it/s', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'amul_us', round(d['roofline']['avg_launch_us'],2))"
done; done
