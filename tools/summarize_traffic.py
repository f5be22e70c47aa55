"""Per-kernel HBM traffic table from tools/pmc_traffic.sh's three passes: python tools/summarize_traffic.py gpurun_out/pmc_<tag> [top]
read bytes = 2 x FETCH_SIZE KiB (gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md), write = WRITE_SIZE KiB."""
import collections, csv, glob, os, subprocess, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
def rows(sub, name):
    f = glob.glob(os.path.join(d, sub, "**", name), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
dur = collections.defaultdict(list)
for r in rows("trace", "*kernel_trace.csv"):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in rows(c, "*counter_collection.csv"):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    cnt[c] = acc
tot = sum(sum(v) for v in dur.values())
print(f"| kernel | calls | avg us | % of kernel time | read MB (2 x FETCH_SIZE) | write MB | traffic MB per launch | GB/s |\n|---|---:|---:|---:|---:|---:|---:|---:|")
sum_traffic = 0.0
for k, v in sorted(dur.items(), key=lambda x: -sum(x[1]))[:top]:
    f, w = cnt["FETCH_SIZE"].get(k, []), cnt["WRITE_SIZE"].get(k, [])
    rd = 2 * 1024 * sum(f) / max(len(f), 1); wr = 1024 * sum(w) / max(len(w), 1)
    avg = sum(v) / len(v) / 1e3
    sum_traffic += (rd + wr) * len(v)
    print(f"| `{k.split('(')[0][:70]}` | {len(v)} | {avg:.2f} | {100 * sum(v) / tot:.1f} | {rd / 1e6:.2f} | {wr / 1e6:.2f} | {(rd + wr) / 1e6:.2f} | {(rd + wr) / avg / 1e3 if avg else 0:.0f} |")
allrd = 2 * 1024 * sum(sum(v) for v in cnt["FETCH_SIZE"].values()); allwr = 1024 * sum(sum(v) for v in cnt["WRITE_SIZE"].values())
print(f"\nall dispatches of the run: read {allrd / 1e6:.1f} MB + write {allwr / 1e6:.1f} MB = {(allrd + allwr) / 1e6:.1f} MB")
print(f"\ntotal kernel time {tot / 1e6:.3f} ms in {sum(len(v) for v in dur.values())} launches; traffic of the listed kernels {sum_traffic / 1e6:.1f} MB (calls x per-launch average)")
