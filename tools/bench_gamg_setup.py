"""per-solve GAMG set-up cost (level matrices re-agglomerated from the fine coefficients, GAMGSolver.C:88-97) vs cycles"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
from oracle import oracle as orc
case = syn.box_case(216, 216, 216)
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
mat = eng.Matrix(addr)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
mat.set_coeffs(t(case.diag), t(case.upper), None)
G = eng.Gamg(addr, orc.box_face_weights(case), 100)
b = t(case.source); psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
G.solve(mat, psi, b, tolerance=1e30, maxIter=5)
for _ in range(3):
    psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    p = G.solve(mat, psi, b, tolerance=1e30, maxIter=5)   # converged at once: set-up + prologue only
    torch.cuda.synchronize(); t_setup = time.perf_counter() - t0
    psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    p2 = G.solve(mat, psi, b, tolerance=0.0, maxIter=10)
    torch.cuda.synchronize(); t_10 = time.perf_counter() - t0
    print(f"setup+prologue {1e3*t_setup:.2f} ms ({p['nIterations']} cycles); 10 cycles total {1e3*t_10:.2f} ms -> {(t_10-t_setup)*100:.2f} ms/cycle")
