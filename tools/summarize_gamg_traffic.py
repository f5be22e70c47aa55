"""HBM traffic of ONE GAMG V-cycle from two PMC runs of tools/bench_gamg.py that differ only in the number of cycles
(tools/gpu_r03_g.sh): python tools/summarize_gamg_traffic.py gpurun_out/pmc_gamg5 5 gpurun_out/pmc_gamg25 25 [--update]
--update writes the figure (with the hash of the sources it belongs to) into profiles/traffic_latest.json for bench.py."""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def total(d):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True)[0]
        out[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c)
    return 2 * 1024 * out["FETCH_SIZE"], 1024 * out["WRITE_SIZE"]
def gamg_source_hash():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_fingerprint
    return source_fingerprint.gamg_source_hash()
if __name__ == "__main__":
    d1, c1, d2, c2 = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    (r1, w1), (r2, w2) = total(d1), total(d2)
    rd, wr = (r2 - r1) / (c2 - c1), (w2 - w1) / (c2 - c1)
    rec = {"workload": "216x216x216", "read_bytes_per_cycle": rd, "write_bytes_per_cycle": wr, "traffic_bytes_per_cycle": rd + wr,
           "method": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over all dispatches of tools/bench_gamg.py with {c2} and with {c1} V-cycles in the "
                     "timed solve, difference / cycles; read = 2 x FETCH_SIZE KiB (MI355X_MICROARCH.md), MI_GAMG_GRAPH=0",
           "gamg_source_sha256_16": gamg_source_hash()}
    print(json.dumps(rec, indent=1))
    if "--update" in sys.argv:
        p = os.path.join(ROOT, "profiles", "traffic_latest.json")
        cur = json.load(open(p))
        cur["gamg"] = rec
        json.dump(cur, open(p, "w"), indent=1)
