#!/bin/bash
# round 3, GPU call B: multi-rhs PBiCG, mirror, transformed processor patches, bench supplements, pbicg timing
mkdir -p gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multi_rhs or pbicg or krylov or asym" > gpurun_out/r03b/t_parity.log 2>&1; echo "parity rc=$?"
timeout 600 python -m pytest tests/test_foam_mirror.py tests/test_ref_dropin.py -x -q -m gpu > gpurun_out/r03b/t_mirror.log 2>&1; echo "mirror rc=$?"
timeout 600 python tools/bench_pbicg.py > gpurun_out/r03b/pbicg.json 2> gpurun_out/r03b/pbicg.err; echo "pbicg rc=$?"

timeout 600 python tools/bench_timestep.py > gpurun_out/r03b/timestep_batched.json 2> gpurun_out/r03b/timestep.err; echo "ts rc=$?"
MI_TIMESTEP_SEGREGATED=1 timeout 600 python tools/bench_timestep.py > gpurun_out/r03b/timestep_segregated.json 2>> gpurun_out/r03b/timestep.err; echo "ts2 rc=$?"

for f in t_parity t_mirror; do tail -n 3 gpurun_out/r03b/$f.log; done
grep -E "ms_per_time_step|PBiCG" gpurun_out/r03b/timestep_*.json
cat gpurun_out/r03b/pbicg.json
