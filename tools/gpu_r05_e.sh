#!/bin/bash
# round 5, last lease: the GPU suite of the closing tree (fresh rank processes per multi-process test, trimmed window variants) + smoke
mkdir -p gpurun_out/r05e
export TMPDIR=/tmp
rm -f gpurun_out/test_durations.tsv gpurun_out/native_solve_timings.tsv
t0=$(date +%s)
{ time timeout 1020 python -m pytest tests -m gpu -q --durations=30 ; } > gpurun_out/r05e/pytest_gpu.log 2>&1
rc=$?
t1=$(date +%s)
echo "pytest exit $rc, $((t1 - t0)) s wall" >> gpurun_out/r05e/pytest_gpu.log
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r05e/smoke.log 2>&1
cp gpurun_out/test_durations.tsv gpurun_out/r05e/
grep -E "passed|failed|pytest exit" gpurun_out/r05e/pytest_gpu.log | tail -4; tail -2 gpurun_out/r05e/smoke.log
sort -rn gpurun_out/test_durations.tsv | head -30
