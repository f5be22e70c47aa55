#!/bin/bash
# round 4, lease l: cyclicAMI whose halves live on different ranks (transport patch), local cyclicAMI regression
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests/test_distributed.py -q -k "cyclic_ami_whose" > $O/r04_l_ami_ranks.log 2>&1; echo "ami ranks rc=$?" | tee -a $O/r04_l_ami_ranks.log
tail -n 30 $O/r04_l_ami_ranks.log | cut -c1-400
timeout 600 python -m pytest tests/test_ami.py -q > $O/r04_l_ami_local.log 2>&1; echo "ami local rc=$?" | tee -a $O/r04_l_ami_local.log; tail -n 3 $O/r04_l_ami_local.log | cut -c1-300
