#!/bin/bash
# round 4, lease n: diagnostics of the cyclicAMI-across-ranks cases (failure details of the first failing case)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
T=tests/test_distributed.py::test_cyclic_ami_whose_halves_live_on_different_ranks
for id in ami_sym-2-False ami_asym-2-auto; do
  timeout 150 python -m pytest "$T[$id]" -q -x > $O/r04_n_ami_$id.log 2>&1; rc=$?
  echo "$id rc=$rc"; grep -E "^E |Error|assert|rank|Traceback" $O/r04_n_ami_$id.log | head -n 30 | cut -c1-400
done
