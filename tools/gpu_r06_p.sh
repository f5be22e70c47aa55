#!/bin/bash
# round 6, lease p: the whole GPU suite, smoke and the bench line on the tree with the huge-page host tables
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
{ time timeout 1150 python -m pytest tests -m gpu -q --durations=25 ; } > gpurun_out/r06_p_suite.log 2>&1
echo "pytest exit $?, $(( $(date +%s) - t0 )) s wall" >> gpurun_out/r06_p_suite.log
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r06_p_smoke.log 2>&1
{ time timeout 600 python bench.py ; } > gpurun_out/r06_p_bench.json 2> gpurun_out/r06_p_bench.err
grep -E "passed|failed|pytest exit" gpurun_out/r06_p_suite.log | tail -3; tail -2 gpurun_out/r06_p_smoke.log; cut -c1-900 gpurun_out/r06_p_bench.json; grep "layout\|hierarchy" gpurun_out/r06_p_bench.err | head
