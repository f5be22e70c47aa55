#!/bin/bash
# round 5, closing lease (pipelined multi-vector pass on by default): GPU suite + smoke, the bench lines, the PBiCG kernel table
mkdir -p gpurun_out/r05g
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -f gpurun_out/test_durations.tsv gpurun_out/native_solve_timings.tsv
t0=$(date +%s)
{ time timeout 700 python -m pytest tests -m gpu -q --durations=30 ; } > gpurun_out/r05g/pytest_gpu.log 2>&1
rc=$?
t1=$(date +%s)
echo "pytest exit $rc, $((t1 - t0)) s wall" >> gpurun_out/r05g/pytest_gpu.log
{ time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r05g/smoke.log 2>&1
cp gpurun_out/test_durations.tsv gpurun_out/r05g/
{ time timeout 400 python bench.py ; } > gpurun_out/r05g/bench.json 2> gpurun_out/r05g/bench.err
{ time timeout 200 python bench.py --solver gamg ; } > gpurun_out/r05g/bench_gamg.json 2> gpurun_out/r05g/bench_gamg.err
(cd /tmp && DIMS=216,216,216 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05g/prof_pbicg -o t -- python $R/tools/bench_pbicg.py > $R/gpurun_out/r05g/bench_pbicg.log 2>&1)
find gpurun_out/r05g -name "*.db" -delete; rm -f gpurun_out/r05g/prof_pbicg/t_kernel_trace.csv
grep -E "passed|failed|pytest exit" gpurun_out/r05g/pytest_gpu.log | tail -4; tail -2 gpurun_out/r05g/smoke.log; cut -c1-400 gpurun_out/r05g/bench.json
