#!/bin/bash
# round 5, lease h: the multi-process tests again after the rendezvous-port retry (lease g: 287 passed, 2 failed with EADDRINUSE)
mkdir -p gpurun_out/r05h
export TMPDIR=/tmp
rm -f gpurun_out/test_durations.tsv
{ time timeout 400 python -m pytest tests/test_distributed.py tests/test_bench_contract.py tests/test_gpu_configs.py -m gpu -q --durations=10 ; } > gpurun_out/r05h/pytest_multiprocess.log 2>&1
echo "pytest exit $?" >> gpurun_out/r05h/pytest_multiprocess.log
{ time timeout 100 python -c "import __graft_entry__ as g; g.smoke()" ; } > gpurun_out/r05h/smoke.log 2>&1
grep -E "passed|failed|pytest exit" gpurun_out/r05h/pytest_multiprocess.log | tail -3; tail -2 gpurun_out/r05h/smoke.log
