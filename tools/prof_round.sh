#!/bin/bash
# rocprofv3 passes of bench.py: kernel trace + PMC passes (each in its own run)
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -o bench -- python $R/bench.py --steps 200 --warmup 10 --no-cpu > $R/gpurun_out/prof/trace_bench.json 2> $R/gpurun_out/prof/trace.err
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof/pmc_$N -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu > $R/gpurun_out/prof/pmc_$N.json 2> $R/gpurun_out/prof/pmc_$N.err
done
cd $R
find gpurun_out/prof -name "*.db" -delete
ls -R gpurun_out/prof | head -50
