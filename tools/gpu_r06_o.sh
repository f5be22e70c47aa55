#!/bin/bash
# round 6, lease o: start-up with the per-tile tables in per-worker bump arenas (csrc/host_tables.hpp), three fresh processes
for rep in 1 2 3; do MI_HOST_THP=1 TAG=r06_o_$rep bash tools/timing_build.sh > gpurun_out/r06_o_$rep.txt 2>&1; done
MI_HOST_THP=0 TAG=r06_o_malloc bash tools/timing_build.sh > gpurun_out/r06_o_malloc.txt 2>&1
grep "== \(rep\|layout\|hierarchy\)\|tiles (threads)" gpurun_out/timing_build_r06_o_*.err | grep -v "threads)  *0.00"
