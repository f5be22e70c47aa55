"""mi_pcg_solve in the launch-bound regime: hipGraph replay of 16-iteration batches (MI_PCG_GRAPH=1) vs plain launches (=0)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
out = {}
for dims in ((32, 32, 32), (54, 54, 54), (108, 108, 108), (216, 216, 216)):
    case = syn.box_case(*dims)
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    mat.set_coeffs(t(case.diag), t(case.upper), None)
    b = t(case.source)
    res = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["MI_PCG_GRAPH"] = mode
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        mat.pcg(psi, b, "diagonal", tolerance=0.0, maxIter=31)
        psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        p = mat.pcg(psi, b, "diagonal", tolerance=0.0, maxIter=319)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.setdefault("graph" if mode == "1" else "launches", []).append(round(1e6 * dt / p["nIterations"], 2))
        res.setdefault("hist_" + mode, p["history"][-1])
    assert res["hist_0"] == res["hist_1"], res
    out[f"{dims[0]}^3"] = {k: v for k, v in res.items() if not k.startswith("hist")}
print(json.dumps(out))
