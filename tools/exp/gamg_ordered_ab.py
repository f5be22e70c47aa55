"""V-cycle on the caller's numbering (default addressing) against the mesh renumbered into the tile order (ordered addressing): ms per cycle and the level shapes"""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft, workloads
graft.build()
pkg = graft.load_package(); eng, syn = pkg.engine, pkg.synthetic
case = syn.box_case(216, 216, 216)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
a0 = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
rc = syn.renumber(case, a0.cell_perm()); rc.dims = case.dims
# face weights of the renumbered mesh: the same faces, found through their (sorted) cell pairs
key = lambda lo, up: lo.astype(np.int64) * case.n_cells + up
inv = np.empty(case.n_cells, np.int64); inv[a0.cell_perm()] = np.arange(case.n_cells)
w0 = workloads.box_pair_weights(case)
olo, oup = a0.cell_perm()[rc.lower_addr], a0.cell_perm()[rc.upper_addr]
k_old = key(np.minimum(olo, oup), np.maximum(olo, oup))
order = np.argsort(key(case.lower_addr, case.upper_addr)); pos = np.searchsorted(key(case.lower_addr, case.upper_addr)[order], k_old)
w1 = w0[order[pos]]
for name, c, addr, w in (("default", case, a0, w0), ("ordered", rc, eng.Addressing(ctx, rc.n_cells, rc.lower_addr, rc.upper_addr, ordered=True, tile_cell_start=a0.tile_starts()), w1)):
    mat = eng.Matrix(addr); mat.set_coeffs(t(c.diag), t(c.upper), None)
    G = eng.Gamg(addr, w, 100)
    psi = torch.zeros(c.n_cells, dtype=torch.float64, device="cuda:0")
    G.solve(mat, psi, t(c.source), tolerance=0.0, maxIter=3)
    ts = []
    for _ in range(3):
        psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
        perf = G.solve(mat, psi, t(c.source), tolerance=0.0, maxIter=20); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    sizes = [G.level_sizes(l) for l in range(G.n_levels)]
    print(name, "ms per cycle", round(1e3 * sorted(ts)[1] / 21, 4), "levels", G.n_levels, "coarse cells/faces:", [(s["n_coarse"], s["n_coarse_faces"]) for s in sizes[:8]], "hist", [round(float(h), 6) for h in perf["history"][:4]])
