#!/bin/bash
# one gpurun call at the end of a round: full GPU suite, smoke, both bench modes, per-op tables (outputs under gpurun_out/)
mkdir -p gpurun_out
export TMPDIR=/tmp
{ time timeout 1200 python -m pytest tests -m gpu -q ; } > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
{ time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/smoke.log 2>&1
{ time timeout 600 python bench.py ; } > gpurun_out/bench.json 2> gpurun_out/bench.err
{ time timeout 300 python bench.py --solver gamg ; } > gpurun_out/bench_gamg.json 2> gpurun_out/bench_gamg.err
timeout 400 python tools/bench_assembly.py > gpurun_out/bench_assembly.log 2>&1
timeout 600 python tools/bench_kernels.py > gpurun_out/bench_kernels.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cat gpurun_out/bench.json | cut -c1-600; cat gpurun_out/bench_gamg.json | cut -c1-300
timeout 600 python tools/bench_dropin.py > gpurun_out/dropin.log 2>&1
