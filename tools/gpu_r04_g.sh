#!/bin/bash
# round 4, lease g: kernel table of one time step (216^3, local and attached forms) under rocprofv3
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
O=gpurun_out; R=$(pwd)
for form in local attached; do
  (cd /tmp && MI_SELFCOMM_ONLY=$form timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04_g_prof_timestep_$form -o t -- python $R/tools/bench_selfcomm_solvers.py --dims 216 216 216 --solver timestep --steps 4 > $R/$O/r04_g_prof_timestep_$form.log 2>&1)
  find $O/r04_g_prof_timestep_$form -name "*.db" -delete; rm -f $O/r04_g_prof_timestep_$form/t_kernel_trace.csv
done
head -n 60 $O/r04_g_prof_timestep_local/t_kernel_stats.csv | cut -c1-150
