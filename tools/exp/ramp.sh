#!/bin/bash
# fresh box: (1) 40 short processes back to back, (2) one process with 40 alloc/free cycles, (3) 20 more short processes
cd "$(dirname "$0")"
t0=$(date +%s.%N)
now() { echo "$(date +%s.%N) - $t0" | bc; }
for i in $(seq 1 40); do echo -n "proc $i t=$(now) "; ./exp_ramp 1 20; done
echo "--- one process, 40 cycles, t=$(now)"; ./exp_ramp 40 20
echo "--- t=$(now)"
for i in $(seq 1 20); do echo -n "proc $i t=$(now) "; ./exp_ramp 1 20; done
