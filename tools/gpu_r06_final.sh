#!/bin/bash
# round 6, lease final: the closing tree once more -- whole GPU suite, smoke, bench line (with the re-tied traffic record)
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/test_durations.tsv
t0=$(date +%s)
{ time timeout 1150 python -m pytest tests -m gpu -x -q ; } > gpurun_out/r06_final_suite.log 2>&1
echo "pytest exit $?, $(( $(date +%s) - t0 )) s wall" >> gpurun_out/r06_final_suite.log
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r06_final_smoke.log 2>&1
{ time timeout 600 python bench.py ; } > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
grep -E "passed|failed|pytest exit" gpurun_out/r06_final_suite.log | tail -3; grep "smoke" gpurun_out/r06_final_smoke.log; cut -c1-300 gpurun_out/r06_final_bench.json
