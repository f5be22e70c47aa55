"""PBiCG + DILU / diagonal on the asymmetric (momentum-like) 216^3 matrix, us per iteration:
   single right-hand side, A p and A^T pT in two passes (MI_PBICG_PAIR=0, the round-2 form) / in one pass (default);
   three right-hand sides in one solve (mi_pbicg_solve_multi), per component-iteration."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
out = {}
dims = tuple(int(v) for v in os.environ.get("DIMS", "216,216,216").split(","))
case = syn.box_case(*dims, symmetric=False)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
b = [t(case.source), t(3.0 * (syn.splitmix_uniform(41, case.n_cells) - 0.5)), t(syn.splitmix_uniform(42, case.n_cells) - 0.5)]
IT = 40
for pre in ("DILU", "diagonal"):
    for pair in ("0", "1"):
        os.environ["MI_PBICG_PAIR"] = pair
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), t(case.lower))
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        mat.pbicg(psi, b[0], pre, tolerance=0.0, maxIter=3)
        ts = []
        for _ in range(3):
            psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            p = mat.pbicg(psi, b[0], pre, tolerance=0.0, maxIter=IT)
            torch.cuda.synchronize(); ts.append(1e6 * (time.perf_counter() - t0) / p["nIterations"])
        out[f"PBiCG+{pre} single rhs, " + ("A/T paired in one pass" if pair == "1" else "two passes (round 2)")] = round(float(np.median(ts)), 1)
    psis = [torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    mat.pbicg_multi(psis, b, pre, tolerance=0.0, maxIter=3)
    ts = []
    for _ in range(3):
        for q in psis: q.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = mat.pbicg_multi(psis, b, pre, tolerance=0.0, maxIter=IT)
        torch.cuda.synchronize(); ts.append(1e6 * (time.perf_counter() - t0) / r[0]["nIterations"])
    out[f"PBiCG+{pre} three rhs in one solve: per iteration of all three"] = round(float(np.median(ts)), 1)
    out[f"PBiCG+{pre} three rhs in one solve: per component-iteration"] = round(float(np.median(ts)) / 3, 1)
print(json.dumps({"workload": f"{dims[0]}x{dims[1]}x{dims[2]} asymmetric box matrix, {IT} fixed iterations, us", "us": out}, indent=1))
