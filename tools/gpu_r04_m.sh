#!/bin/bash
# round 4, lease m: cyclicAMI across ranks case by case (short timeouts, stop at the first failure), then the regression subsets
# for what changed underneath every attached path (agreed any-AMI flag at attach time, dual-form window pushes)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
T=tests/test_distributed.py::test_cyclic_ami_whose_halves_live_on_different_ranks
ok=1
for id in ami_sym-2-False ami_sym-2-auto ami_asym-2-False ami_asym-2-auto ami_sym-4-False ami_sym-4-auto; do
  timeout 170 python -m pytest "$T[$id]" -q -x > $O/r04_m_ami_$id.log 2>&1; rc=$?
  echo "$id rc=$rc $(grep -E 'passed|failed|Error' $O/r04_m_ami_$id.log | tail -n 1 | cut -c1-200)"
  if [ $rc -ne 0 ]; then ok=0; grep -E "^E |Error|assert" $O/r04_m_ami_$id.log | head -n 20 | cut -c1-300; break; fi
done
run() { name=$1; shift; timeout 900 python -m pytest "$@" -q > $O/r04_m_$name.log 2>&1; echo "$name rc=$? $(grep -E 'passed|failed' $O/r04_m_$name.log | tail -n 1 | cut -c1-200)"; }
run parity tests/test_gpu_parity.py -k "decomposed_solver_paths or attached_comm or distributed_matrix_single or fused_distributed or persistent_distributed or native_rccl"
run gamg tests/test_gamg.py -k "coupled"
run ami tests/test_ami.py
run dist tests/test_distributed.py -k "not cyclic_ami_whose"
run configs tests/test_gpu_configs.py
run bench tests/test_bench_contract.py -k "rehearsal"
grep -E "FAILED" $O/r04_m_*.log | head -n 20 | cut -c1-300
