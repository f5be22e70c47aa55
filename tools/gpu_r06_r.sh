#!/bin/bash
# round 6, lease r: kernel trace of one PISO step (GAMG's per-solve fixed cost after the fused prologue / graph-replayed agglomeration)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_piso
cd /tmp
STEPS=3 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_piso/trace -o piso -- python $R/tools/bench_timestep.py > $R/gpurun_out/prof_piso/bench.log 2> $R/gpurun_out/prof_piso/trace.err
cd $R
find gpurun_out/prof_piso -name "*.db" -delete
python - <<'PY' > gpurun_out/r06_r_piso_kernels.txt
import csv, glob, re
rows = []
for f in glob.glob("gpurun_out/prof_piso/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("mi::", ""))[:90]
tail = rows[-520:]
t0 = tail[0][0]
for s, e, n in tail:
    print(f"{(s - t0) * 1e-3:10.1f} us  {(e - s) * 1e-3:8.1f} us  {short(n)}")
PY
rm -rf gpurun_out/prof_piso/trace
tail -3 gpurun_out/prof_piso/bench.log
