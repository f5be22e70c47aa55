#!/bin/bash
# The closing lease of a round, run AFTER the last source commit: the whole GPU suite inside the driver's window (it FAILS when
# the suite needs more than 700 s), smoke, the bench line of both modes.  Outputs under gpurun_out/; profiles/rNN_gpu_suite_summary.txt
# is made from them (tail of the pytest log incl. --durations=30).
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/test_durations.tsv gpurun_out/native_solve_timings.tsv
LIMIT=${SUITE_LIMIT:-700}
t0=$(date +%s)
{ time timeout 1150 python -m pytest tests -m gpu -q --durations=30 ; } > gpurun_out/pytest_gpu.log 2>&1
rc=$?
t1=$(date +%s)
echo "pytest exit $rc, $((t1 - t0)) s wall (limit $LIMIT s)" >> gpurun_out/pytest_gpu.log
{ time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/smoke.log 2>&1
{ time timeout 900 python bench.py ; } > gpurun_out/bench.json 2> gpurun_out/bench.err
{ time timeout 300 python bench.py --solver gamg ; } > gpurun_out/bench_gamg.json 2> gpurun_out/bench_gamg.err
grep -E "passed|failed|pytest exit" gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-700 gpurun_out/bench.json; cut -c1-300 gpurun_out/bench_gamg.json
if [ $rc -ne 0 ] || [ $((t1 - t0)) -gt $LIMIT ]; then echo "FINAL ROUND: the GPU suite failed or exceeded $LIMIT s"; exit 1; fi
