"""Per-kernel table (calls, average / total duration) of a rocprofv3 results database: python tools/prof_db.py <dir or .db> [top]"""
import collections, glob, os, sqlite3, sys
path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
db = sqlite3.connect(dbs[-1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
try:
    import subprocess
    names = sorted({r[0] for r in rows})
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True, text=True).stdout.splitlines()
    dm = dict(zip(names, dem))
except Exception:
    dm = {}
agg = collections.defaultdict(list)
for n, s, e in rows:
    agg[dm.get(n, n)].append(e - s)
tot = sum(sum(v) for v in agg.values())
print(f"| kernel | calls | avg us | total ms | % |\n|---|---:|---:|---:|---:|")
for n, v in sorted(agg.items(), key=lambda x: -sum(x[1]))[:top]:
    short = n.split("(")[0]
    print(f"| `{short[:110]}` | {len(v)} | {sum(v) / len(v) / 1e3:.2f} | {sum(v) / 1e6:.3f} | {100.0 * sum(v) / tot:.1f} |")
print(f"\ntotal kernel time {tot / 1e6:.3f} ms in {len(rows)} launches")
