"""Small-matrix PCG: the five-launch device-resident pipeline (hipGraph replay below 4 M cells) against the persistent
cooperative kernel that keeps the vectors in registers (csrc/persist.inc), us per iteration by wall clock around mi_pcg_iterate."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
out = {}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
DIMS = [tuple(int(v) for v in d.split("x")) for d in os.environ.get("DIMS", "32x32x32,54x54x54,80x80x80,108x108x108").split(",")]   # DIMS=108x108x108: one size
for dims in DIMS:
    case = syn.box_case(*dims)
    row = {}
    for mode in ("0", "1"):
        os.environ["MI_PCG_PERSIST"] = mode
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
        src = t(case.source); psi0 = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        K = 400
        mat.pcg_begin(psi0, src, "diagonal", tolerance=0.0, relTol=0.0, maxIter=6 * K, history_len=6 * K + 2)
        mat.pcg_iterate(16); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K // 16): mat.pcg_iterate(16)
            torch.cuda.synchronize(); ts.append(1e6 * (time.perf_counter() - t0) / K)
        perf = mat.pcg_end(None, history_len=6 * K + 2)
        assert perf["nIterations"] == 16 + 4 * K, perf["nIterations"]
        row["persistent kernel" if mode == "1" else "five launches"] = {"us_per_iteration": round(float(np.median(ts)), 2), "tiles": addr.n_tiles,
                                                                         "residual_after": float(perf["history"][-1])}
        del mat, addr
    out[f"{dims[0]}^3"] = row
    print(dims, json.dumps(row), flush=True)
print(json.dumps(out))
