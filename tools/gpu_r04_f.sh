#!/bin/bash
# round 4, lease f: the whole GPU suite on the current tree
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r04_f_gpu_suite.log 2>&1; echo "suite rc=$?" | tee -a gpurun_out/r04_f_gpu_suite.log
tail -n 15 gpurun_out/r04_f_gpu_suite.log | cut -c1-400
