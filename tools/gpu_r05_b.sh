#!/bin/bash
# round 5, lease b: the GPU suite again (one rank pool at a time, tests grouped by world size), per-test durations as they complete
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/test_durations.tsv
{ time timeout 1250 python -m pytest tests -m gpu -q --durations=50 ; } > gpurun_out/r05_b_suite.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05_b_suite.log
grep -E "passed|failed|exit" gpurun_out/r05_b_suite.log | tail -5
sort -rn gpurun_out/test_durations.tsv | head -40
