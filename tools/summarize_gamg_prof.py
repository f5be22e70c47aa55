"""Per-kernel and per-level view of one GAMG V-cycle from a rocprofv3 kernel trace (tools/prof_gamg.sh).
Cycles of the timed solve of tools/bench_gamg.py (after its 3 warm-up cycles), delimited by the finest residual kernel."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
files = glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))))
rows.sort()
short = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("mi::", ""))[:60]
log = open(os.path.join(d, "bench_gamg.log")).read() if os.path.exists(os.path.join(d, "bench_gamg.log")) else ""
print("# GAMG V-cycle on the 216^3 box: rocprofv3 kernel trace\n")
print("```\n" + "\n".join(l for l in log.splitlines() if l.startswith(("addr", "{", "solve", "PCG")))[:1500] + "\n```\n")
# find the solve with tolerance 0 / GAMG_CYCLES cycles: cycles are delimited by k_gamg_scale-free marker: the finest residual reduce is followed by a D2H copy; use the finest prolong
fin = [i for i, r in enumerate(rows) if "k_fold_final" in r[2]]   # the finest residual (GAMGSolverSolve.C:146-160): once per cycle
ncyc = int(os.environ.get("GAMG_CYCLES", "10"))
# cycles 4 .. 3+ncyc belong to the timed solve (3 warm-up cycles first); take the middle ones
sel = fin[3 + 1: 3 + ncyc - 1]
a, b = sel[0], sel[-1]
seg = rows[a:b]
nc = len(sel) - 1
span = (rows[b][0] - rows[a][0]) * 1e-3
busy = sum(e - s for s, e, *_ in seg) * 1e-3
print(f"{nc} consecutive cycles of the timed solve: {span / nc:.1f} us per cycle wall (first kernel to first kernel), {busy / nc:.1f} us of kernel time, "
      f"{(span - busy) / nc:.1f} us of gaps between kernels, {len(seg) / nc:.1f} launches per cycle\n")
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g, w in seg:
    k = short(n); agg[k][0] += 1; agg[k][1] += (e - s) * 1e-3
print("| kernel | launches / cycle | us / cycle | avg us | % of kernel time |\n|---|---:|---:|---:|---:|")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {c / nc:.1f} | {t / nc:.1f} | {t / c:.2f} | {100 * t / busy:.1f} |")
# per level: tile kernels are identified by their grid (tiles of the level x threads); everything between two restrict/prolong kernels belongs to a level
print("\n## tile kernels by level (grid = tiles x workgroup size)\n\n| grid (workgroups) | launches / cycle | us / cycle | avg us |\n|---:|---:|---:|---:|")
lv = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g, w in seg:
    if "tile_kernel" in n:
        lv[g // max(w, 1)][0] += 1; lv[g // max(w, 1)][1] += (e - s) * 1e-3
for g, (c, t) in sorted(lv.items(), key=lambda kv: -kv[0]):
    print(f"| {g} | {c / nc:.1f} | {t / nc:.1f} | {t / c:.2f} |")
small = sum(t for g, (c, t) in lv.items() if g <= 64)
print(f"\ntile kernels of levels with <= 64 tiles: {small / nc:.1f} us per cycle")
# time line of one cycle (compressed)
print("\n## one cycle, in order (kernel, workgroups, us, gap to the previous kernel's end in us)\n\n```")
one = rows[sel[len(sel) // 2]: sel[len(sel) // 2 + 1]]
prev = None
for s, e, n, g, w in one:
    print(f"{short(n):58s} {g // max(w, 1):7d} {(e - s) * 1e-3:8.2f} {0.0 if prev is None else (s - prev) * 1e-3:7.2f}")
    prev = e
print("```")
