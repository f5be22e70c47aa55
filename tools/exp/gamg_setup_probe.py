import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build(); pkg = graft.load_package(); syn, eng = pkg.synthetic, pkg.engine
case = syn.box_case(216, 216, 216); N = case.n_cells
dev = torch.device("cuda:0"); t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
mat = eng.Matrix(addr); d, u = t(case.diag), t(case.upper); mat.set_coeffs(d, u, None)
lo, up = case.lower_addr.astype(np.int64), case.upper_addr.astype(np.int64)
w = (1.0 / 216) * np.array([1.0, 1.01, 1.02])[np.where(up - lo == 1, 0, np.where(up - lo == 216, 1, 2))]
G = eng.Gamg(addr, w, 100); src = t(case.source)
for label, kw in (("0 cycles (tolerance 1e30)", dict(tolerance=1e30)), ("1 cycle", dict(tolerance=0.0, maxIter=1)), ("3 cycles", dict(tolerance=0.0, maxIter=3))):
    for rep in range(4):
        psi = torch.zeros(N, dtype=torch.float64, device=dev); torch.cuda.synchronize(); t0 = time.perf_counter()
        p = G.solve(mat, psi, src, **kw); torch.cuda.synchronize()
        print(label, "rep", rep, f"{1e3 * (time.perf_counter() - t0):.2f} ms", p["nIterations"], flush=True)
    mat.set_coeffs(d, u, None); torch.cuda.synchronize(); t0 = time.perf_counter(); mat.set_coeffs(d, u, None); torch.cuda.synchronize()
    print("   set_coeffs", f"{1e3 * (time.perf_counter() - t0):.2f} ms")
