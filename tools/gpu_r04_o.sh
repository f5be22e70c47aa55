#!/bin/bash
# round 4, lease o: neighbours in different window forms (the exchange of iterations enqueued after convergence), the 4-rank
# cyclicAMI case over windows, the agreed any-factor / any-compact flags under the transformed-patch tests
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out
one() { name=$1; shift; timeout 170 python -m pytest "$@" -q -x > $O/r04_o_$name.log 2>&1; rc=$?; echo "$name rc=$rc $(grep -E 'passed|failed|Error' $O/r04_o_$name.log | tail -n 1 | cut -c1-200)"; if [ $rc -ne 0 ]; then grep -E "^E |Error|assert" $O/r04_o_$name.log | head -n 12 | cut -c1-300; fi; }
T=tests/test_distributed.py
one ami4auto "$T::test_cyclic_ami_whose_halves_live_on_different_ranks[ami_sym-4-auto]"
one mixed2 "$T::test_neighbours_in_different_window_forms[box_2-2]"
one mixed4 "$T::test_neighbours_in_different_window_forms[box_4_asym-4]"
one transf "$T::test_transformed_processor_patches_between_engine_ranks"
one ami2 "$T::test_cyclic_ami_whose_halves_live_on_different_ranks" -k "2-auto or 2-False"
one peerwin "$T::test_native_solvers_entirely_over_peer_windows"
one parity tests/test_gpu_parity.py -k "decomposed_solver_paths or fused_distributed"
grep -E "FAILED" $O/r04_o_*.log | head -n 20 | cut -c1-300
