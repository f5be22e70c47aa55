"""decision table of tools/first_lease.sh: one row per (GPUs, forced path) from the bench JSON lines"""
import glob, json, os, sys
d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, "bench_*_*.json"))):
    n, path = os.path.basename(f)[6:-5].split("_", 1)
    try:
        line = [l for l in open(f) if l.strip().startswith("{")][-1]
        j = json.loads(line)
        sup = j["config"].get("supplements") or {}
        gam = next((v["gamg"]["ms_per_v_cycle"] for k, v in sup.items() if k.startswith("decomposed_") and "gamg" in v), None)
        ts = next((v["timestep"]["ms_per_time_step"] for k, v in sup.items() if k.startswith("decomposed_") and "timestep" in v), None)
        rows.append((int(n), path, j["value"], j["ms_per_step"] * 1e3, j["config"]["host_loop"], j["config"]["allreduce"], (j["config"].get("weak_scaling_supplement") or {}).get("iterations_per_s"), gam, ts))
    except Exception as e:
        rows.append((int(n), path, None, None, f"no JSON line ({type(e).__name__}: {e})", "", None, None, None))
print("| GPUs | forced path | PCG it/s (10 M cells, strong) | us/iteration | host loop that ran | all-reduce that ran | weak it/s (216^3 per GPU) | GAMG ms/V-cycle (216^3 per rank) | time step ms (216^3 per rank) |")
print("|---:|---|---:|---:|---|---|---:|---:|---:|")
f = lambda v, p=1: "-" if v is None else f"{v:.{p}f}"
for r in sorted(rows):
    print(f"| {r[0]} | {r[1]} | {f(r[2])} | {f(r[3])} | {r[4]} | {r[5]} | {f(r[6])} | {f(r[7], 3)} | {f(r[8], 2)} |")
