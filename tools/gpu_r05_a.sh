#!/bin/bash
# round 5, lease a: the restructured GPU suite with durations, smoke, the PBiCG kernels after the staging change, the time step
mkdir -p gpurun_out
export TMPDIR=/tmp
{ time timeout 1100 python -m pytest tests -m gpu -q --durations=70 ; } > gpurun_out/r05_a_suite.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_a_suite.log
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r05_a_smoke.log 2>&1
timeout 300 python tools/bench_pbicg.py > gpurun_out/r05_a_pbicg.json 2> gpurun_out/r05_a_pbicg.err
timeout 200 python tools/bench_timestep.py > gpurun_out/r05_a_timestep.json 2> gpurun_out/r05_a_timestep.err
grep -E "passed|failed|exit" gpurun_out/r05_a_suite.log | tail -5; tail -2 gpurun_out/r05_a_smoke.log; tail -12 gpurun_out/r05_a_pbicg.json
