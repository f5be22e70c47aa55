"""Level-1 vs level-2 integration on the 216^3 box: the REFERENCE's own PCG::solve / GAMGSolver::solve
(tests/ref_dropin, compiled in place for gfx950) running on the engine's primitives through the C ABI, against the engine's
fused device-resident solvers.  Same matrix, same iteration counts (tolerance 0)."""
import ctypes as C, json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
lib = C.CDLL(os.path.join(ROOT, "tests", "ref_dropin", "_ref", "libref_dropin.so"))
lib.ref_dropin_solve.restype = None; lib.ref_dropin_gamg_solve.restype = None; lib.ref_dropin_solve_order.restype = None
dims = [int(v) for v in os.environ.get("DIMS", "216,216,216").split(",")]
case = syn.box_case(*dims)
n = case.n_cells
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
stream = torch.cuda.current_stream().cuda_stream
ctx = eng.Context(0, stream)
addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
src = t(case.source)
out = {}
for precond in ("diagonal", "AINV"):
    iters = 200
    for name in ("reference PCG::solve on engine primitives (level 1)", "reference PCG::solve on ENGINE-ORDER primitives (level 1)", "mi_pcg_solve (level 2)"):
        best = 1e9
        for rep in range(3):
            psi = torch.zeros(n, dtype=torch.float64, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if name.startswith("reference"):
                o5 = (C.c_double * 5)()
                lib.ref_dropin_solve_order(C.c_int(0), ctx.h, mat.h, C.c_void_p(stream), C.c_int(n), C.c_void_p(psi.data_ptr()), C.c_void_p(src.data_ptr()),
                                           C.c_int(eng.PRECOND[precond]), C.c_double(0.0), C.c_double(0.0), C.c_int(iters), C.c_int(0), C.c_int(1), C.c_double(0.9),
                                           C.c_int(1 if "ENGINE-ORDER" in name else 0), o5)
                nit = int(o5[2])
            else:
                nit = mat.pcg(psi, src, precond, tolerance=0.0, maxIter=iters)["nIterations"]
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        out[f"PCG {precond}: {name}"] = {"iterations": nit, "us_per_iteration": 1e6 * best / nit}
# ---- the same with ORDERED addressing (mesh renumbered once with the engine's cell order; mi_addr_create_ordered): the
# caller-order primitives the reference's solvers bind to pay no permutation passes any more
# round 3: renumber-at-bind -- ONE call (mi_addr_create_adopted) renumbers the mesh into the engine order and returns the maps the
# shim permutes its fields with, once (new[i] = old[cell_map[i]]; faces through face_map)
t0 = time.perf_counter()
addr_o = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr, adopt=True)
out["renumber-at-bind (mi_addr_create_adopted): seconds"] = time.perf_counter() - t0
assert addr_o.is_ordered
mat_o = eng.Matrix(addr_o); mat_o.set_coeffs(t(case.diag[addr_o.cell_map]), t(case.upper[addr_o.face_map]), None)
src_o = t(case.source[addr_o.cell_map])
for precond in ("diagonal", "AINV"):
    best = 1e9
    for rep in range(3):
        psi = torch.zeros(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o5 = (C.c_double * 5)()
        lib.ref_dropin_solve_order(C.c_int(0), ctx.h, mat_o.h, C.c_void_p(stream), C.c_int(n), C.c_void_p(psi.data_ptr()), C.c_void_p(src_o.data_ptr()),
                                   C.c_int(eng.PRECOND[precond]), C.c_double(0.0), C.c_double(0.0), C.c_int(200), C.c_int(0), C.c_int(1), C.c_double(0.9), C.c_int(0), o5)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    out[f"PCG {precond}: reference PCG::solve on CALLER-ORDER primitives, mesh renumbered at bind (level 1)"] = {"iterations": int(o5[2]), "us_per_iteration": 1e6 * best / int(o5[2])}
alg = 24 * n + 16 * case.n_faces
x = t(syn.splitmix_uniform(1, n)); y = torch.empty_like(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def timed(fn, reps=40):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for label, M, S in (("clustered addressing (permutation passes)", mat, src), ("ordered addressing", mat_o, src_o)):
    row = {"mi_amul": timed(lambda: M.amul(x, y)), "mi_residual": timed(lambda: M.residual(x, S, y)),
           "mi_precondition AINV": timed(lambda: M.precondition("AINV", x, y)), "mi_jacobi_smooth x2": timed(lambda: M.jacobi_smooth(x.clone(), S, 2), 20)}
    out[f"caller-order operators, {label}"] = {k: {"us": v, "frac_of_8TBps_on_Amul_bytes": alg / (v * 1e-6) / 8e12} for k, v in row.items()}
d = np.asarray(case.upper_addr, np.int64) - np.asarray(case.lower_addr, np.int64)
w = (1.0 / dims[0]) * np.array([1.0, 1.01, 1.02])[np.where(d == 1, 0, np.where(d == dims[0], 1, 2))]
G = eng.Gamg(addr, w, 100)
ctl = eng.gamg_controls(tolerance=0.0, maxIter=20)
for name in ("reference GAMGSolver::solve on engine level operators (level 1)", "mi_gamg_solve (level 2)"):
    best = 1e9
    for rep in range(3):
        psi = torch.zeros(n, dtype=torch.float64, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if name.startswith("reference"):
            o5 = (C.c_double * 5)()
            lib.ref_dropin_gamg_solve(ctx.h, G.h, mat.h, C.c_void_p(stream), C.c_int(n), C.c_int(0), C.c_void_p(psi.data_ptr()), C.c_void_p(src.data_ptr()), C.byref(ctl), o5)
            nit = int(o5[2])
        else:
            nit = G.solve(mat, psi, src, tolerance=0.0, maxIter=20)["nIterations"]
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    out[f"GAMG: {name}"] = {"cycles": nit, "ms_per_cycle": 1e3 * best / nit}
print(json.dumps(out, indent=1))
