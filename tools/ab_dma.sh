#!/bin/bash
# A/B on one box: direct-to-LDS staging (the in-tree library) vs the register-staged copies (tools/exp/librapidcfd_amd_regstage.so,
# built with -DMI_STAGE_THROUGH_REGISTERS by:  make -C tools/exp librapidcfd_amd_regstage.so)
mkdir -p gpurun_out
for r in 1 2 3; do
  MI_ENGINE_LIB=$PWD/tools/exp/librapidcfd_amd_regstage.so timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu > gpurun_out/abd_reg_$r.json 2>> gpurun_out/abd.err
  timeout 300 python bench.py --steps 300 --warmup 20 --no-cpu > gpurun_out/abd_dma_$r.json 2>> gpurun_out/abd.err
done
MI_ENGINE_LIB=$PWD/tools/exp/librapidcfd_amd_regstage.so timeout 300 python tools/bench_kernels.py > /dev/null 2>&1; cp gpurun_out/kernel_table.md gpurun_out/kernel_table_reg.md
timeout 300 python tools/bench_kernels.py > /dev/null 2>&1; cp gpurun_out/kernel_table.md gpurun_out/kernel_table_dma.md
for f in gpurun_out/abd_reg_1.json gpurun_out/abd_dma_1.json gpurun_out/abd_reg_2.json gpurun_out/abd_dma_2.json gpurun_out/abd_reg_3.json gpurun_out/abd_dma_3.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d["value"],1), "it/s  amul us", round(d["roofline"]["avg_launch_us"],1))
PY
done
for k in reg dma; do echo "== $k"; grep "Amul\|Tmul\|AINV\|Jacobi\|PCG iteration" gpurun_out/kernel_table_$k.md | cut -c1-90; done
