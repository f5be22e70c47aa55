#!/bin/bash
# round 6: kernel trace of one rhoPimpleFoam step (where does the energy equation's time go?) -> gpurun_out/r06_h_rhopimple_kernels.txt
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_rhop
cd /tmp
STEPS=3 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_rhop/trace -o rhop -- python $R/tools/bench_rhopimple.py > $R/gpurun_out/prof_rhop/bench.log 2> $R/gpurun_out/prof_rhop/trace.err
cd $R
find gpurun_out/prof_rhop -name "*.db" -delete
python - <<'PY' > gpurun_out/r06_h_rhopimple_kernels.txt
import csv, glob, re
rows = []
for f in glob.glob("gpurun_out/prof_rhop/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("mi::", ""))[:90]
# the last step: from the last upwind-weights kernel... simply print the last 400 kernels in order with durations
tail = rows[-420:]
t0 = tail[0][0]
for s, e, n in tail:
    print(f"{(s - t0) * 1e-3:10.1f} us  {(e - s) * 1e-3:8.1f} us  {short(n)}")
PY
tail -5 gpurun_out/prof_rhop/bench.log
