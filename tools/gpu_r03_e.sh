#!/bin/bash
mkdir -p gpurun_out/r03e
bash tools/ab_compact.sh > gpurun_out/r03e/ab_compact.log 2>&1; echo "ab rc=$?"
timeout 900 python tools/bench_dropin.py > gpurun_out/r03e/dropin.json 2> gpurun_out/r03e/dropin.err; echo "dropin rc=$?"
cat gpurun_out/r03e/ab_compact.log; grep -E "us_per_iteration|seconds|PCG" gpurun_out/r03e/dropin.json | head -40
