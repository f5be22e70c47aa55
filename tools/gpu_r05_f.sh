#!/bin/bash
# round 5, lease f: the pipelined multi-vector tile pass (csrc/multi_pipe.inc, MI_MULTI_PIPE=1) -- bitwise against the plain kernel,
# against the oracle at 216^3, and timed both ways (PBiCG solvers, the PISO time step)
mkdir -p gpurun_out/r05f
export TMPDIR=/tmp
{ time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "pipelined or multi_rhs or krylov or pbicg or bicg" ; } > gpurun_out/r05f/tests_parity.log 2>&1
echo "exit $?" >> gpurun_out/r05f/tests_parity.log
{ time MI_MULTI_PIPE=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "multi_rhs or krylov or pbicg or bicg or rebind" ; } > gpurun_out/r05f/tests_pipe_on.log 2>&1
echo "exit $?" >> gpurun_out/r05f/tests_pipe_on.log
for p in 0 1; do
  MI_MULTI_PIPE=$p timeout 200 python tools/bench_pbicg.py > gpurun_out/r05f/bench_pbicg_pipe$p.json 2> gpurun_out/r05f/bench_pbicg_pipe$p.err
  MI_MULTI_PIPE=$p timeout 200 python tools/bench_timestep.py > gpurun_out/r05f/timestep_pipe$p.json 2> gpurun_out/r05f/timestep_pipe$p.err
done
tail -4 gpurun_out/r05f/tests_parity.log; tail -4 gpurun_out/r05f/tests_pipe_on.log
for p in 0 1; do grep "three rhs\|paired" gpurun_out/r05f/bench_pbicg_pipe$p.json; grep "ms_per_time_step\|PBiCG" gpurun_out/r05f/timestep_pipe$p.json; done
