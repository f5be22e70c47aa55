#!/bin/bash
# round 5, lease i (the last GPU seconds): the pipelined multi-vector pass with the interleaved {upper, lower} LDS image -- bitwise against the
# plain kernel and the oracle (targeted tests), then the PBiCG timings
mkdir -p gpurun_out/r05i
export TMPDIR=/tmp
{ time timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -k "pipelined or multi_rhs or krylov or pbicg or bicg or rebind" ; } > gpurun_out/r05i/tests.log 2>&1
echo "exit $?" >> gpurun_out/r05i/tests.log
timeout 60 python tools/bench_pbicg.py > gpurun_out/r05i/bench_pbicg.json 2> gpurun_out/r05i/bench_pbicg.err
tail -3 gpurun_out/r05i/tests.log; grep "three rhs\|paired" gpurun_out/r05i/bench_pbicg.json
