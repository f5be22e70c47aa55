#!/bin/bash
# round 6, lease n: start-up after the host tables moved to transparent huge pages (the caller's face weights generated outside the timed call),
# then the GAMG V-cycle's HBM traffic by PMC again (gamg_engine.inc's text changed: Table<> for std::vector<>)
for rep in 1 2; do MI_HOST_THP=1 TAG=r06_n_thp_$rep bash tools/timing_build.sh > gpurun_out/r06_n_thp_$rep.txt 2>&1; done
MI_HOST_THP=0 TAG=r06_n_malloc bash tools/timing_build.sh > gpurun_out/r06_n_malloc.txt 2>&1
grep "== \(rep\|layout\|hierarchy\)" gpurun_out/timing_build_r06_n_*.err
bash tools/gpu_r06_k.sh
