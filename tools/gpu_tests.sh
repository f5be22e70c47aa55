#!/bin/bash
# run a selection of the GPU tests on the box:  bash tools/gpu_tests.sh <tag> <pytest arguments...>  -> gpurun_out/<tag>_tests.log
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
{ time timeout 1150 python -m pytest "$@" -m gpu -q ; } > gpurun_out/${TAG}_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_tests.log
grep -E "passed|failed|error|pytest exit" gpurun_out/${TAG}_tests.log | tail -5
