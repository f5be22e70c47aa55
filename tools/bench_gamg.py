"""GAMG timing on the 216^3 box (BASELINE config 3): hierarchy build, cycles/s, convergence."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
dims = [int(v) for v in os.environ.get("GAMG_DIMS", "216,216,216").split(",")]
case = syn.box_case(*dims)
nx = dims[0]
lo, up = case.lower_addr.astype(np.int64), case.upper_addr.astype(np.int64)
d = up - lo
direction = np.where(d == 1, 0, np.where(d == nx, 1, 2))
w = (1.0 / nx) * np.array([1.0, 1.01, 1.02])[direction]
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
t0 = time.perf_counter(); addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr); t1 = time.perf_counter()
mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
G = eng.Gamg(addr, w, int(os.environ.get("GAMG_NCOARSEST", "100"))); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"addr {t1-t0:.2f}s  hierarchy+level layouts {t2-t1:.2f}s  levels {G.n_levels}", [G.level_sizes(l)["n_coarse"] for l in range(G.n_levels)], flush=True)
src = t(case.source)
psi = torch.zeros(case.n_cells, dtype=torch.float64, device=dev)
perf = G.solve(mat, psi, src, tolerance=0.0, maxIter=3)  # warm-up
psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
ncyc = int(os.environ.get("GAMG_CYCLES", "20"))
perf = G.solve(mat, psi, src, tolerance=0.0, maxIter=ncyc)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps(dict(cells=case.n_cells, cycles=perf["nIterations"], seconds=dt, cycles_per_s=perf["nIterations"] / dt,
                      ms_per_cycle=1e3 * dt / perf["nIterations"], residual=perf["finalResidual"], history=perf["history"][:8].tolist())), flush=True)
psi.zero_(); t0 = time.perf_counter()
perf = G.solve(mat, psi, src, tolerance=1e-6, maxIter=200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("solve to 1e-6:", perf["nIterations"], "cycles", f"{dt:.3f}s", perf["finalResidual"], flush=True)
psi.zero_(); t0 = time.perf_counter()
p2 = mat.pcg(psi, src, "diagonal", tolerance=1e-6, maxIter=5000); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
print("PCG-diagonal to 1e-6:", p2["nIterations"], "iterations", f"{dt2:.3f}s", p2["finalResidual"], flush=True)
