#!/bin/bash
# round 4, lease i: the rest of the GPU suite after the history-bar fix, then the time-step kernel tables and the closing evidence
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_layout.py tests/test_polymesh.py tests/test_ref_dropin.py tests/test_oracle.py tests/test_golden.py tests/test_abi.py -m gpu -q > gpurun_out/r04_i_rest_of_suite.log 2>&1; echo "rest rc=$?" | tee -a gpurun_out/r04_i_rest_of_suite.log
tail -n 6 gpurun_out/r04_i_rest_of_suite.log | cut -c1-300
bash tools/gpu_r04_h.sh
bash tools/gpu_r04_g.sh | head -n 45
