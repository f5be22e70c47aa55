#!/bin/bash
# one gpurun call: tests, smoke, bench, sweep, rocprof (outputs under gpurun_out/)
mkdir -p gpurun_out
export TMPDIR=/tmp
{ time timeout 900 python -m pytest tests -m gpu -x -q ; } > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
{ time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/smoke.log 2>&1
{ time timeout 600 python bench.py --steps 200 --warmup 10 ; } > gpurun_out/bench.json 2> gpurun_out/bench.err
if [ -n "$DO_SWEEP" ]; then { time timeout 900 python tools/sweep.py ; } > gpurun_out/sweep.log 2>&1 ; fi
if [ -n "$DO_PROF" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof -name "*.db" -size +20M -delete 2>/dev/null
  ls -la gpurun_out/prof/* > gpurun_out/prof_ls.txt 2>&1
fi
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
