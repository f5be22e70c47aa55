#!/bin/bash
# round 4, lease t: block-local 16-bit row tables of the assembly row passes (MI_ROW16): parity, then A/B timing at 216^3
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 80 python -m pytest tests/test_assembly.py -m gpu -q -x > $O/r04_t_tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/r04_t_tests.log | tail -n 1)"
for v in 1 0; do
  MI_ROW16=$v timeout 45 python tools/bench_assembly.py > $O/r04_t_bench_row16_$v.log 2>&1; echo "bench MI_ROW16=$v rc=$?"
  cp $O/assembly_row_passes.json $O/r04_t_assembly_row16_$v.json 2>/dev/null
  grep -E "laplacian|fvm::div|negSumDiag|surfaceIntegrate" $O/r04_t_bench_row16_$v.log | cut -c1-170
done
