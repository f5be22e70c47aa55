#!/bin/bash
# kernel trace of the PISO-like time step (2 timed steps): which launches make up the momentum and pressure stages
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O
cd /tmp
STEPS=2 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/bench_timestep.py > $O/timestep.json 2> $O/trace.err
cd $R
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/r03i/trace/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last time step: from the last k_upwind-like first kernel... print the sequence of the final 1/3 of launches compactly
import collections
t_end = int(rows[-1]['End_Timestamp'])
seq = [(r['Kernel_Name'].split('(')[0][:60], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp'])) for r in rows]
# find start of last step: look for last occurrence of 'k_limited' or 'k_upwind' names
idx = max(i for i, s in enumerate(seq) if 'upwind' in s[0] or 'k_face_weights' in s[0] or 'weights' in s[0]) if any('upwind' in s[0] or 'weights' in s[0] for s in seq) else len(seq) * 2 // 3
last = seq[idx:]
t0 = last[0][2]
out = []
prev = None; cnt = 0; tot = 0.0; start = 0
for name, us, st in last:
    if name == prev: cnt += 1; tot += us
    else:
        if prev: out.append((start, prev, cnt, tot))
        prev, cnt, tot, start = name, 1, us, (st - t0) / 1e3
out.append((start, prev, cnt, tot))
for st, n, c, t in out: print(f"{st:9.1f} us  {n:60s} x{c:<4d} {t:9.1f} us")
print("launches in the step:", len(last), " span", (int(rows[-1]['End_Timestamp']) - t0) / 1e3, "us")
PY
