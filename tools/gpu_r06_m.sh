#!/bin/bash
# round 6: start-up with the host tables on transparent huge pages (csrc/host_tables.hpp) against plain malloc, same build, same box
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag > gpurun_out/r06_m_thp_settings.txt 2>&1
nproc >> gpurun_out/r06_m_thp_settings.txt
for rep in 1 2; do
  MI_HOST_THP=1 TAG=r06_m_thp_$rep bash tools/timing_build.sh > gpurun_out/r06_m_thp_$rep.txt 2>&1
  MI_HOST_THP=0 TAG=r06_m_malloc_$rep bash tools/timing_build.sh > gpurun_out/r06_m_malloc_$rep.txt 2>&1
done
grep "== \(rep\|layout\|hierarchy\)" gpurun_out/timing_build_r06_m_*.err
