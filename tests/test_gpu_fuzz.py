"""GPU (-m gpu): randomised sweep of the tile engine against the oracle.  Ragged graphs (hub-free and hubby), symmetric and
asymmetric coefficients, a random cyclic patch pair (partners in the same tile, in other tiles, repeated cells), tiny to
normal tiles (many halos / cut faces), both row-entry formats: every operator of the SpMV family bit for bit, AINV and
Jacobi bit for bit, patchNeighbourField, and a few Krylov iterations within the history bar."""
import copy

import numpy as np
import pytest
import torch

from conftest import random_graph_case

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


CASES = [(seed, n, extra, tile, sym, entry16)
         for seed, n, extra, tile in [(1, 40, 1.0, 8), (2, 257, 2.0, 64), (3, 900, 3.0, 128), (4, 1500, 1.5, 1024), (5, 3000, 2.5, 256),
                                      (6, 64, 6.0, 16), (7, 2200, 0.6, 96)]
         for sym in (True, False) for entry16 in (0, 1)]


@pytest.mark.parametrize("seed,n,extra,tile,sym,entry16", CASES)
def test_random_coupled_matrices_every_operator(pkg, orc, seed, n, extra, tile, sym, entry16, monkeypatch):
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = copy.copy(random_graph_case(pkg, n, extra=extra, seed=seed, symmetric=sym))
    npair = max(1, n // 15)
    u = syn.splitmix_uniform(seed + 11, 2 * npair)
    a = (u[:npair] * n).astype(np.int32); b = (u[npair:] * n).astype(np.int32)
    kb = -(0.05 + 0.3 * syn.splitmix_uniform(seed + 12, npair))
    ki = kb if sym else -(0.05 + 0.3 * syn.splitmix_uniform(seed + 13, npair))
    case.interfaces = [syn.Interface(0, 1, a, kb, ki), syn.Interface(0, 0, b, kb, ki)]
    case.diag = case.diag + np.bincount(a, -kb, n) + np.bincount(b, -kb, n)      # keep the rows dominant
    monkeypatch.setenv("MI_TILE_CELLS", str(tile))
    monkeypatch.setenv("MI_ENTRY16", str(entry16))
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr, [a, b], [b, a])
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if sym else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if sym else dev(itf.int_coeffs))
    S = orc.System([case])
    x = syn.splitmix_uniform(seed + 7, n) - 0.5
    xd, bd = dev(x), dev(case.source)
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(xd, out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(xd, out); assert np.array_equal(host(out), S.tmul(x))
    mat.sumA(out); assert np.array_equal(host(out), S.sumA())
    mat.residual(xd, bd, out); assert np.array_equal(host(out), S.residual(x, case.source))
    mat.H(xd, out); assert np.array_equal(host(out), S.H(x))
    mat.H1(out); assert np.array_equal(host(out), S.H1())
    fh = torch.empty(case.n_faces, dtype=torch.float64, device="cuda:0")
    mat.faceH(xd, fh); assert np.array_equal(host(fh), S.faceH(x))
    mat.precondition("AINV", xd, out); assert np.array_equal(host(out), S.precondition("AINV", x))
    mat.precondition("AINV", xd, out, transpose=True); assert np.array_equal(host(out), S.precondition("AINV", x, transpose=True))
    mat.precondition("diagonal", xd, out); assert np.array_equal(host(out), S.precondition("diagonal", x))
    for sweeps in (1, 3):
        psi = dev(x.copy()); mat.jacobi_smooth(psi, bd, sweeps)
        assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, sweeps))
    nbr = torch.empty(2 * npair, dtype=torch.float64, device="cuda:0")
    mat.patch_neighbour_field(xd, nbr); assert np.array_equal(host(nbr), np.concatenate([x[b], x[a]]))
    nf = mat.norm_factor(xd, bd, dev(S.amul(x)))
    _, pref = S.pcg(x.copy(), case.source, "diagonal", tolerance=0.0, maxIter=1) if sym else S.pbicg(x.copy(), case.source, "diagonal", tolerance=0.0, maxIter=1)
    assert abs(nf - pref["normFactor"]) <= 1e-12 * pref["normFactor"]
    # a few Krylov iterations: same counts, histories within the bar
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    if sym:
        perf = mat.pcg(psi, bd, "AINV", tolerance=1e-10, maxIter=40)
        ref_psi, ref = S.pcg(np.zeros(n), case.source, "AINV", tolerance=1e-10, maxIter=40)
    else:
        perf = mat.pbicgstab(psi, bd, "AINV", tolerance=1e-10, maxIter=25)
        ref_psi, ref = S.pbicgstab(np.zeros(n), case.source, "AINV", tolerance=1e-10, maxIter=25)
    assert perf["nIterations"] == ref["nIterations"]
    h, hr = perf["history"], ref["history"]
    assert np.max(np.abs(h - hr)) <= 1e-9 * hr[0]
    assert np.max(np.abs(host(psi) - ref_psi)) <= 1e-8 * np.max(np.abs(ref_psi))


@pytest.mark.parametrize("seed,n,extra,sym,merge,kw", [(21, 600, 2.0, True, 1, {}), (22, 1500, 1.2, True, 2, dict(nPreSweeps=1)),
                                                        (23, 900, 3.0, False, 1, {}), (24, 2500, 2.0, False, 2, {}),
                                                        (25, 400, 1.0, True, 3, dict(scaleCorrection=0)), (26, 1200, 2.5, True, 1, dict(nFinestSweeps=3, nPostSweeps=1))])
def test_random_coupled_matrices_gamg(pkg, orc, seed, n, extra, sym, merge, kw, monkeypatch):
    """GAMG on ragged graphs with a cyclic patch pair: engine hierarchy (agglomeration of the interfaces included) and
    V-cycle against the oracle's multi-domain restatement; small tiles so that every level has halos."""
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = copy.copy(random_graph_case(pkg, n, extra=extra, seed=seed, symmetric=sym))
    npair = max(1, n // 20)
    u = syn.splitmix_uniform(seed + 11, 2 * npair)
    a = (u[:npair] * n).astype(np.int32); b = (u[npair:] * n).astype(np.int32)
    kb = -(0.05 + 0.3 * syn.splitmix_uniform(seed + 12, npair))
    ki = kb if sym else -(0.05 + 0.3 * syn.splitmix_uniform(seed + 13, npair))
    case.interfaces = [syn.Interface(0, 1, a, kb, ki), syn.Interface(0, 0, b, kb, ki)]
    case.diag = case.diag + np.bincount(a, -kb, n) + np.bincount(b, -kb, n)
    w = 0.5 + syn.splitmix_uniform(seed + 5, case.n_faces)
    monkeypatch.setenv("MI_TILE_CELLS", "128")
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr, [a, b], [b, a])
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if sym else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if sym else dev(itf.int_coeffs))
    args = dict(tolerance=1e-9, maxIter=40); args.update(kw)
    S = orc.System([case])
    H = orc.GamgSysHierarchy(S, [w], 8, merge_levels=merge)
    ref_psi, ref = H.solve(np.zeros(n), case.source, **args)
    G = eng.Gamg(addr, w, 8, merge_levels=merge)
    assert G.n_levels == H.n_levels
    for l in range(G.n_levels):
        o, e = H.level(0, l), G.level_sizes(l)
        assert (o["n_coarse"], o["n_coarse_faces"]) == (e["n_coarse"], e["n_coarse_faces"])
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, dev(case.source), **args)
    assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == ref["converged"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape and np.max(np.abs(h - hr)) <= 1e-9 * hr[0]
    assert np.max(np.abs(host(psi) - ref_psi)) <= 1e-8 * np.max(np.abs(ref_psi))


@pytest.mark.parametrize("seed,n,extra,sym", [(31, 500, 2.0, True), (32, 1800, 1.5, False), (33, 120, 4.0, True), (34, 2600, 2.2, False)])
def test_random_matrices_over_rccl_self_exchange(pkg, orc, seed, n, extra, sym, monkeypatch):
    """The same kind of ragged coupled matrix posed as a DECOMPOSED case: the two patches are processor patches whose
    neighbour rank is this rank (1-rank RCCL communicator), so every operator packs, sends, receives and reduces for real."""
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = copy.copy(random_graph_case(pkg, n, extra=extra, seed=seed, symmetric=sym))
    npair = max(1, n // 12)
    u = syn.splitmix_uniform(seed + 11, 2 * npair)
    a = (u[:npair] * n).astype(np.int32); b = (u[npair:] * n).astype(np.int32)
    kb = -(0.05 + 0.3 * syn.splitmix_uniform(seed + 12, npair))
    ki = kb if sym else -(0.05 + 0.3 * syn.splitmix_uniform(seed + 13, npair))
    case.interfaces = [syn.Interface(0, 1, a, kb, ki), syn.Interface(0, 0, b, kb, ki)]
    case.diag = case.diag + np.bincount(a, -kb, n) + np.bincount(b, -kb, n)
    monkeypatch.setenv("MI_TILE_CELLS", "96")
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr, [a, b])            # no partner cells: processor patches
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if sym else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if sym else dev(itf.int_coeffs))
    comm = eng.Comm(ctx, 1, 0, eng.Comm.unique_id())
    mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=n)
    S = orc.System([case])
    x = syn.splitmix_uniform(seed + 7, n) - 0.5
    xd, bd = dev(x), dev(case.source)
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(xd, out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(xd, out); assert np.array_equal(host(out), S.tmul(x))
    mat.residual(xd, bd, out); assert np.array_equal(host(out), S.residual(x, case.source))
    psi = dev(x.copy()); mat.jacobi_smooth(psi, bd, 2)
    assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 2))
    nbr = torch.empty(2 * npair, dtype=torch.float64, device="cuda:0")
    mat.patch_neighbour_field(xd, nbr); assert np.array_equal(host(nbr), np.concatenate([x[b], x[a]]))
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    if sym:
        perf = mat.pcg(psi, bd, "AINV", tolerance=1e-10, maxIter=40); ref_psi, ref = S.pcg(np.zeros(n), case.source, "AINV", tolerance=1e-10, maxIter=40)
    else:
        perf = mat.pbicg(psi, bd, "AINV", tolerance=1e-10, maxIter=30); ref_psi, ref = S.pbicg(np.zeros(n), case.source, "AINV", tolerance=1e-10, maxIter=30)
    assert perf["nIterations"] == ref["nIterations"]
    assert np.max(np.abs(perf["history"] - ref["history"])) <= 1e-9 * ref["history"][0]
    assert np.max(np.abs(host(psi) - ref_psi)) <= 1e-8 * np.max(np.abs(ref_psi))
    w = 0.5 + syn.splitmix_uniform(seed + 5, case.n_faces)
    H = orc.GamgSysHierarchy(S, [w], 8)
    ref_psi, ref = H.solve(np.zeros(n), case.source, tolerance=1e-9, maxIter=30)
    G = eng.Gamg(addr, w, 8, comms=(comm, comm), patch_rank=[0, 0], patch_nbr_patch=[1, 0])
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, bd, tolerance=1e-9, maxIter=30)
    assert perf["nIterations"] == ref["nIterations"] and G.n_levels == H.n_levels
    assert np.max(np.abs(perf["history"] - ref["history"])) <= 1e-9 * ref["history"][0]
    assert np.max(np.abs(host(psi) - ref_psi)) <= 1e-8 * np.max(np.abs(ref_psi))
    del G
    mat.detach_comm(); comm.close()
