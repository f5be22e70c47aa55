#!/bin/bash
# PMC counters for the tile kernel only (bench with few steps)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
cd /tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc/$N -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > /dev/null 2> $R/gpurun_out/pmc/$N.err
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/*/b_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'tile_kernel<0' in r['Kernel_Name']:
            agg['amul'][r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg['amul'].items()):
    print(c, round(sum(v)/len(v)))
PY
