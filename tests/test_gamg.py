"""GAMG: oracle properties and engine-vs-oracle hierarchy (CPU); engine V-cycle parity (gpu)."""
import numpy as np
import pytest

from conftest import random_graph_case


def graph_weights(pkg, case):
    return 0.5 + pkg.synthetic.splitmix_uniform(77, case.n_faces)


@pytest.mark.parametrize("kind,ncoarsest,forward", [("box", 10, True), ("box", 50, False), ("graph", 10, True), ("box_odd", 4, True)])
def test_engine_hierarchy_equals_oracle(pkg, orc, kind, ncoarsest, forward):
    # two independent restatements (C in oracle/, C++ in the engine) of pairGAMGAgglomerate.C +
    # GAMGAgglomerateLduAddressing.C must agree exactly
    if kind == "box":
        case = pkg.synthetic.box_case(14, 11, 9); w = orc.box_face_weights(case)
    elif kind == "box_odd":
        case = pkg.synthetic.box_case(7, 5, 3); w = orc.box_face_weights(case)
    else:
        case = random_graph_case(pkg, 800); w = graph_weights(pkg, case)
    H = orc.GamgHierarchy(case, w, ncoarsest, forward)
    E = pkg.engine.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, w, ncoarsest, forward)
    assert H.n_levels == len(E) >= 2
    for l in range(H.n_levels):
        o, e = H.level(l), E[l]
        assert np.array_equal(o["restrict"], e["restrictMap"])
        assert np.array_equal(o["lower"], e["cLower"]) and np.array_equal(o["upper"], e["cUpper"])
        assert np.array_equal(o["face_restrict"], e["faceRestrict"])
        assert np.array_equal(o["face_flip"], e["faceFlip"].astype(bool))
        assert np.all(o["lower"] < o["upper"])
        # every coarse cell has 1..n children and pairs dominate
        cnt = np.bincount(o["restrict"], minlength=o["n_coarse"])
        assert cnt.min() >= 1
        assert np.array_equal(np.diff(e["cellChildStart"]), cnt)
    assert H.level(H.n_levels - 1)["n_coarse"] >= ncoarsest


@pytest.mark.parametrize("kind", ["box", "graph"])
@pytest.mark.parametrize("merge", [2, 3])
def test_merge_levels_folds_consecutive_pair_steps(pkg, orc, kind, merge):
    """mergeLevels m (pairGAMGAgglomerate.C:110-117 -> GAMGAgglomeration::combineLevels): the pair steps are the same as
    with mergeLevels 1; created level k is steps m*k .. m*k+m-1 composed; the face map follows the composition and its
    flip is the LAST step's flip (the reference drops the earlier ones, GAMGAgglomerateLduAddressing.C:624-629).
    Oracle and engine builders (independent restatements) must agree exactly."""
    if kind == "box":
        case = pkg.synthetic.box_case(14, 11, 9); w = orc.box_face_weights(case)
    else:
        case = random_graph_case(pkg, 800); w = graph_weights(pkg, case)
    H1 = orc.GamgHierarchy(case, w, 6, True)
    Hm = orc.GamgHierarchy(case, w, 6, True, merge_levels=merge)
    E = pkg.engine.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, w, 6, True, merge_levels=merge)
    assert Hm.n_levels == len(E) == -(-H1.n_levels // merge) and Hm.forward_out == H1.forward_out
    for k in range(Hm.n_levels):
        steps = [H1.level(l) for l in range(merge * k, min(merge * k + merge, H1.n_levels))]
        rm = steps[0]["restrict"].copy(); fr = steps[0]["face_restrict"].copy(); flip = steps[0]["face_flip"].copy()
        for st in steps[1:]:
            pos = fr >= 0
            cell_of_interior = -fr[~pos] - 1
            new_fr = np.empty_like(fr); new_flip = np.zeros_like(flip)
            new_fr[pos] = st["face_restrict"][fr[pos]]; new_flip[pos] = st["face_flip"][fr[pos]]
            new_fr[~pos] = -st["restrict"][cell_of_interior] - 1
            fr, flip, rm = new_fr, new_flip, st["restrict"][rm]
        o, e = Hm.level(k), E[k]
        assert np.array_equal(o["restrict"], rm) and np.array_equal(o["face_restrict"], fr)
        used = fr >= 0
        assert np.array_equal(o["face_flip"][used], flip[used])
        assert np.array_equal(o["lower"], steps[-1]["lower"]) and np.array_equal(o["upper"], steps[-1]["upper"])
        assert np.array_equal(e["restrictMap"], o["restrict"]) and np.array_equal(e["faceRestrict"], o["face_restrict"])
        assert np.array_equal(e["faceFlip"].astype(bool)[used], o["face_flip"][used])
        assert np.array_equal(e["cLower"], o["lower"]) and np.array_equal(e["cUpper"], o["upper"])
        cnt = np.bincount(o["restrict"], minlength=o["n_coarse"])
        assert np.array_equal(np.diff(e["cellChildStart"]), cnt) and cnt.min() >= 1


def test_iterative_coarsest_solver_matches_the_direct_one(pkg, orc):
    """directSolveCoarsest false (GAMGSolverSolve.C:572-613): ICCG / BICCG to GAMG's tolerance on the coarsest level instead of
    the LU.  With a tight tolerance the coarsest correction is the same to rounding, so the cycle counts agree and the
    histories stay close; symmetric and asymmetric."""
    for sym in (True, False):
        case = pkg.synthetic.box_case(14, 12, 10, symmetric=sym)
        w = orc.box_face_weights(case)
        H = orc.GamgHierarchy(case, w, 10)
        _, pd = H.solve(np.zeros(case.n_cells), case.source, tolerance=1e-10, maxIter=100)
        _, pi = H.solve(np.zeros(case.n_cells), case.source, tolerance=1e-10, maxIter=100, directSolveCoarsest=False)
        assert pd["converged"] and pi["converged"] and abs(pd["nIterations"] - pi["nIterations"]) <= 1
        k = min(len(pd["history"]), len(pi["history"]), 6)
        assert np.max(np.abs(pd["history"][:k] - pi["history"][:k])) < 1e-6 * pd["history"][0]


def test_merge_levels_oracle_solves(pkg, orc):
    case = pkg.synthetic.box_case(16, 12, 10)
    w = orc.box_face_weights(case)
    _, p1 = orc.GamgHierarchy(case, w, 10).solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=100)
    _, p2 = orc.GamgHierarchy(case, w, 10, merge_levels=2).solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=100)
    assert p1["converged"] and p2["converged"] and p2["nIterations"] <= 3 * p1["nIterations"]


def test_coarse_matrix_is_galerkin_by_summation(pkg, orc):
    # piecewise-constant restriction R: coarse A = R A R^T (GAMGSolverAgglomerateMatrix.C)
    case = pkg.synthetic.box_case(8, 7, 6, symmetric=False)
    H = orc.GamgHierarchy(case, orc.box_face_weights(case), 8)
    n = case.n_cells
    A = np.zeros((n, n)); A[np.arange(n), np.arange(n)] = case.diag
    A[case.lower_addr, case.upper_addr] = case.upper; A[case.upper_addr, case.lower_addr] = case.lower
    for l in range(min(3, H.n_levels)):
        lv = H.level(l)
        R = np.zeros((lv["n_coarse"], lv["n_fine"])); R[lv["restrict"], np.arange(lv["n_fine"])] = 1.0
        A = R @ A @ R.T
        d, u, lo = H.coarse_matrix(l)
        Ac = np.zeros_like(A); Ac[np.arange(len(d)), np.arange(len(d))] = d
        Ac[lv["lower"], lv["upper"]] = u; Ac[lv["upper"], lv["lower"]] = lo
        assert np.max(np.abs(Ac - A)) < 1e-13 * np.max(np.abs(A))


@pytest.mark.parametrize("symmetric", [True, False])
def test_oracle_gamg_converges_mesh_independently(pkg, orc, symmetric):
    its = []
    for dims in [(12, 12, 12), (24, 24, 24)]:
        case = pkg.synthetic.box_case(*dims, symmetric=symmetric)
        H = orc.GamgHierarchy(case, orc.box_face_weights(case), 10)
        psi, perf = H.solve(np.zeros(case.n_cells), case.source, tolerance=1e-7, maxIter=200)
        assert perf["converged"]
        r = orc.System([case]).residual(psi, case.source)
        assert abs(np.abs(r).sum() / perf["normFactor"] - perf["finalResidual"]) < 1e-12
        its.append(perf["nIterations"])
    if symmetric:
        assert abs(its[0] - its[1]) <= 4  # multigrid: iteration count (almost) independent of the mesh size


def test_oracle_gamg_controls(pkg, orc):
    case = pkg.synthetic.box_case(12, 10, 8)
    H = orc.GamgHierarchy(case, orc.box_face_weights(case), 10)
    z = np.zeros(case.n_cells)
    _, p = H.solve(z, case.source, tolerance=0.0, maxIter=5)
    assert p["nIterations"] == 5  # ++nIterations < maxIter (GAMGSolverSolve.C:166-174): exactly maxIter cycles
    _, p0 = H.solve(z, case.source, tolerance=1e-6, nPreSweeps=2)
    _, p1 = H.solve(z, case.source, tolerance=1e-6)
    assert p0["converged"] and p1["converged"] and abs(p0["nIterations"] - p1["nIterations"]) <= 5
    _, p2 = H.solve(z, case.source, tolerance=1e-6, scaleCorrection=0)
    assert p2["converged"]


# ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["box_sym", "box_asym", "graph_sym"])
def test_engine_gamg_operators_bit_exact(pkg, orc, name):
    import torch
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    if name == "graph_sym":
        case = random_graph_case(pkg, 2500); w = graph_weights(pkg, case)
    else:
        case = pkg.synthetic.box_case(20, 16, 12, symmetric=(name == "box_sym")); w = orc.box_face_weights(case)
    H = orc.GamgHierarchy(case, w, 10)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    G = eng.Gamg(addr, w, 10)
    assert G.n_levels == H.n_levels and G.forward_out == H.forward_out
    for l in range(H.n_levels):
        lv = H.level(l)
        assert G.level_sizes(l) == {k: lv[k] for k in ("n_fine", "n_fine_faces", "n_coarse", "n_coarse_faces")}
        ff = pkg.synthetic.splitmix_uniform(l + 1, lv["n_fine"]) - 0.5
        cf = torch.empty(lv["n_coarse"], dtype=torch.float64, device="cuda:0")
        G.restrict(l, dev(ff), cf)
        ref = np.zeros(lv["n_coarse"]); np.add.at(ref, lv["restrict"], ff)   # ascending fine index
        assert np.array_equal(cf.cpu().numpy(), ref)
        back = torch.empty(lv["n_fine"], dtype=torch.float64, device="cuda:0")
        G.prolong(l, cf, back)
        assert np.array_equal(back.cpu().numpy(), ref[lv["restrict"]])
        d = torch.empty(lv["n_coarse"], dtype=torch.float64, device="cuda:0")
        u = torch.empty(max(lv["n_coarse_faces"], 1), dtype=torch.float64, device="cuda:0")
        lo = torch.empty(max(lv["n_coarse_faces"], 1), dtype=torch.float64, device="cuda:0") if case.lower is not None else None
        G.level_coeffs(mat, l, d, u, lo)
        rd, ru, rl = H.coarse_matrix(l)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), rd)
        assert np.array_equal(u.cpu().numpy()[: lv["n_coarse_faces"]], ru)
        if lo is not None:
            assert np.array_equal(lo.cpu().numpy()[: lv["n_coarse_faces"]], rl)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("box_sym", {}), ("box_sym", dict(nPreSweeps=1)), ("box_sym", dict(scaleCorrection=0)),
                                     ("box_asym", {}), ("graph_sym", {}), ("box_sym", dict(tolerance=0.0, maxIter=4)),
                                     ("box_sym", dict(tolerance=1e30, minIter=2)), ("box_sym", dict(merge_levels=2)),
                                     ("box_asym", dict(merge_levels=2)), ("graph_sym", dict(merge_levels=3, nPreSweeps=1)),
                                     ("box_sym", dict(directSolveCoarsest=False)), ("box_asym", dict(directSolveCoarsest=False))])
def test_engine_gamg_history(pkg, orc, name, kw, monkeypatch):
    import torch
    if kw.get("nPreSweeps") or name == "box_asym":        # some cases invert the coarsest matrix on the device, the others on the host
        monkeypatch.setenv("MI_GAMG_DEVICE_INVERT", "1")
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    if name == "graph_sym":
        case = random_graph_case(pkg, 2500); w = graph_weights(pkg, case)
    else:
        case = pkg.synthetic.box_case(24, 20, 16, symmetric=(name == "box_sym")); w = orc.box_face_weights(case)
    args = dict(tolerance=1e-9, maxIter=100); args.update(kw)
    merge = args.pop("merge_levels", 1)
    H = orc.GamgHierarchy(case, w, 10, merge_levels=merge)
    ref_psi, ref = H.solve(np.zeros(case.n_cells), case.source, **args)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    G = eng.Gamg(addr, w, 10, merge_levels=merge)
    assert G.n_levels == H.n_levels
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, dev(case.source), **args)
    assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == ref["converged"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape
    assert np.max(np.abs(h - hr)) < 1e-10 * hr[0]
    torch.cuda.synchronize()
    assert np.max(np.abs(psi.cpu().numpy() - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))
    # a second solve on the cached hierarchy gives the same bits (level matrices are rebuilt, hierarchy is not)
    psi2 = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf2 = G.solve(mat, psi2, dev(case.source), **args)
    assert np.array_equal(perf2["history"], perf["history"])
    # ... and re-binding other coefficients invalidates the kept level matrices: the next solve is the new matrix's
    import copy
    case2 = copy.copy(case); case2.diag = case.diag * 1.07
    mat.set_coeffs(dev(case2.diag), dev(case2.upper), None if case2.lower is None else dev(case2.lower))
    psi3 = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf3 = G.solve(mat, psi3, dev(case.source), **args)
    _, ref3 = orc.GamgHierarchy(case2, w, 10, merge_levels=merge).solve(np.zeros(case.n_cells), case.source, **args)
    assert perf3["nIterations"] == ref3["nIterations"] and np.max(np.abs(perf3["history"] - ref3["history"])) < 1e-10 * ref3["history"][0]


# ---- coupled patches / decomposed cases -------------------------------------------------------------------
def test_oracle_sys_gamg_equals_single_domain_path(pkg, orc):
    # D = 1 without interfaces: the system restatement and the original single-domain one are the same arithmetic
    case = pkg.synthetic.box_case(14, 11, 9)
    w = orc.box_face_weights(case)
    x1, p1 = orc.GamgHierarchy(case, w, 10).solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=60)
    S = orc.System([case])
    x2, p2 = orc.GamgSysHierarchy(S, [w], 10).solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=60)
    assert np.array_equal(p1["history"], p2["history"]) and np.array_equal(x1, x2)


@pytest.mark.parametrize("parts", [(2, 1, 1), (2, 2, 1), (2, 2, 2)])
def test_oracle_gamg_on_a_decomposed_case(pkg, orc, parts):
    # every domain agglomerates on its own, interfaces are agglomerated from both sides' coarse ids, the coarsest
    # level is the global system: the solve converges like the serial one and to the same solution
    syn = pkg.synthetic
    case = syn.box_case(16, 12, 12)
    subs = syn.decompose_box(case, parts)
    S = orc.System(subs)
    H = orc.GamgSysHierarchy(S, [orc.box_face_weights(s) for s in subs], 10)
    assert H.n_levels >= 3
    # both sides of every processor patch built the same coarse interface faces
    for d, sub in enumerate(subs):
        for p, itf in enumerate(sub.interfaces):
            a = H.patch(d, 0, p, len(itf.face_cells))
            b = H.patch(itf.nbr_domain, 0, itf.nbr_patch, len(itf.face_cells))
            assert np.array_equal(a["face_restrict"], b["face_restrict"]) and len(a["face_cells"]) == len(b["face_cells"])
    src = np.concatenate([s.source for s in subs])
    x, perf = H.solve(np.zeros(S.n), src, tolerance=1e-9, maxIter=100)
    assert perf["converged"]
    xs, ps = orc.GamgHierarchy(case, orc.box_face_weights(case), 10).solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=100)
    assert abs(perf["nIterations"] - ps["nIterations"]) <= 4
    glob = np.concatenate([s.global_cells for s in subs])
    assert np.max(np.abs(x - xs[glob])) < 1e-6 * np.max(np.abs(xs))
    assert abs(perf["normFactor"] - ps["normFactor"]) < 1e-10 * ps["normFactor"]


def test_oracle_gamg_cyclic(pkg, orc):
    syn = pkg.synthetic
    case = syn.add_cyclic_y(syn.box_case(12, 12, 10))
    S = orc.System([case])
    H = orc.GamgSysHierarchy(S, [orc.box_face_weights(case)], 10)
    x, perf = H.solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=100)
    assert perf["converged"] and perf["nIterations"] < 40
    xp, _ = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e-12, maxIter=2000)
    assert np.max(np.abs(x - xp)) < 1e-6 * np.max(np.abs(xp))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cyclic", "processor_to_self"])
@pytest.mark.parametrize("symmetric,kw", [(True, {}), (True, dict(nPreSweeps=1)), (False, {}), (True, dict(merge_levels=2)), (False, dict(merge_levels=2)),
                                          (True, dict(directSolveCoarsest=False)), (False, dict(directSolveCoarsest=False))])
def test_engine_gamg_coupled_patches(pkg, orc, mode, symmetric, kw, monkeypatch):
    """GAMG on a matrix with coupled patches: 'cyclic' = local patches (cyclicGAMGInterface), 'processor_to_self' = the
    same periodic box posed with processor patches whose neighbour rank is this rank, on a 1-rank RCCL communicator --
    restrict-addressing exchange, per-level halo exchange, all-reduced scale factors and the global coarsest system all
    run for real.  The oracle solves the same system with its own restatement of the interface agglomeration."""
    import torch
    if not symmetric or kw.get("merge_levels"):           # device-side assembly + inversion of the (global) coarsest system
        monkeypatch.setenv("MI_GAMG_DEVICE_INVERT", "1")
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    case = syn.add_cyclic_y(syn.box_case(20, 16, 12, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.2)
    w = orc.box_face_weights(case)
    S = orc.System([case])
    args = dict(tolerance=1e-9, maxIter=100); args.update(kw)
    merge = args.pop("merge_levels", 1)
    H = orc.GamgSysHierarchy(S, [w], 10, merge_levels=merge)
    ref_psi, ref = H.solve(np.zeros(case.n_cells), case.source, **args)
    fcs = [i.face_cells for i in case.interfaces]
    comm = None
    if mode == "cyclic":
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs, [case.interfaces[i.nbr_patch].face_cells for i in case.interfaces])
    else:
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if symmetric else dev(itf.int_coeffs))
    if mode == "cyclic":
        G = eng.Gamg(addr, w, 10, merge_levels=merge)
    else:
        comm = eng.Comm(ctx, 1, 0, eng.Comm.unique_id())
        mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=case.n_cells)
        G = eng.Gamg(addr, w, 10, comms=(comm, comm), patch_rank=[0, 0], patch_nbr_patch=[1, 0], merge_levels=merge)
    assert G.n_levels == H.n_levels
    for l in range(G.n_levels):
        o, e = H.level(0, l), G.level_sizes(l)
        assert (o["n_coarse"], o["n_coarse_faces"]) == (e["n_coarse"], e["n_coarse_faces"])
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, dev(case.source), **args)
    assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == ref["converged"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape
    assert np.max(np.abs(h - hr)) < 1e-10 * hr[0]
    torch.cuda.synchronize()
    assert np.max(np.abs(psi.cpu().numpy() - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


@pytest.mark.gpu
def test_full_size_gamg_properties(pkg, orc):
    """BASELINE config 3 size (216^3, GAMG pressure solve): size-independent properties -- the reported residual is the true
    residual of the returned psi, the cycle count is mesh independent (same as at 24^3 +- 4), PCG agrees on the solution."""
    import torch
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    small = syn.box_case(24, 24, 24)
    _, ps = orc.GamgHierarchy(small, orc.box_face_weights(small), 10).solve(np.zeros(small.n_cells), small.source, tolerance=1e-6, maxIter=100)
    case = syn.box_case(216, 216, 216)
    n = case.n_cells
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None)
    G = eng.Gamg(addr, orc.box_face_weights(case), 100)
    assert G.n_levels >= 12 and G.level_sizes(0)["n_coarse"] == n // 2
    b = dev(case.source)
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, b, tolerance=1e-6, maxIter=100)
    assert perf["converged"] and abs(perf["nIterations"] - ps["nIterations"]) <= 4
    h = perf["history"]
    assert np.all(np.diff(h) < 0)                                  # every V-cycle reduces the residual
    r = torch.empty_like(psi); mat.residual(psi, b, r)
    assert abs(ctx.sum_mag(r) / perf["normFactor"] - perf["finalResidual"]) < 1e-8 * perf["finalResidual"]
    psi2 = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    p2 = mat.pcg(psi2, b, "diagonal", tolerance=1e-8, maxIter=5000)
    assert p2["converged"]
    assert float(torch.max(torch.abs(psi - psi2))) < 1e-3 * float(torch.max(torch.abs(psi2)))


# ---- pinned against the REFERENCE's own code -----------------------------------------------------------------------
def _ref_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_pair.npz"))


def test_pair_agglomeration_equals_the_reference_code(pkg, orc):
    """tests/golden/golden_ref_pair.npz holds the coarse-cell maps computed by pairGAMGAgglomeration::agglomerate COMPILED
    FROM /root/reference (oracle/ref_shim + oracle/Makefile `ref`; generator tests/golden/make_golden_ref.py).  Both
    restatements -- the oracle's C and the engine's C++ -- must reproduce them on every level, both sweep directions,
    boxes, ragged graphs and an all-ties case."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = _ref_golden()
    n_checked = 0
    for name, (case, w) in make_golden_ref.cases(pkg, orc).items():
        for forward in (True, False):
            H = orc.GamgHierarchy(case, w, 4, forward)
            E = pkg.engine.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, w, 4, forward)
            keys = sorted(k for k in G.files if k.startswith(f"{name}/fwd{int(forward)}/"))
            assert len(keys) == H.n_levels == len(E) >= 1, name
            for l in range(H.n_levels):
                ref = G[f"{name}/fwd{int(forward)}/level{l}"]
                assert np.array_equal(H.level(l)["restrict"], ref), (name, forward, l, "oracle")
                assert np.array_equal(E[l]["restrictMap"], ref), (name, forward, l, "engine")
                n_checked += 1
    assert n_checked >= 40


def test_reference_code_live_when_its_build_is_present(pkg, orc):
    # where oracle/_ref/libref_pair.so exists (this container; it also travels to the GPU box) the goldens are regenerated
    # from the reference's code on the spot
    if not orc.ref_pair_available():
        pytest.skip("oracle/_ref/libref_pair.so not built (needs /root/reference)")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    now = make_golden_ref.build(pkg, orc)
    G = _ref_golden()
    assert sorted(now) == sorted(G.files)
    for k in G.files:
        assert np.array_equal(now[k], G[k]), k


def test_vcycle_equals_the_reference_source(pkg, orc):
    """tests/golden/golden_ref_gamg.npz was produced by the reference's GAMGSolver::solve, ::Vcycle, ::initVcycle and
    ::solveCoarsestLevel COMPILED FROM /root/reference (solvers/GAMG/GAMGSolverSolve.C against oracle/ref_shim/
    foam_gamg_shim.H) running on this oracle's hierarchy, level matrices, smoother, restrict/prolong, scale and coarsest LU.
    The oracle's own V-cycle (orc_gamg_solve_sys) must give the same bits: pre/post/finest sweep schedules, level
    multipliers, scaling on and off, symmetric and asymmetric, decomposed and cyclic systems, the minIter rule."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_gamg.npz"))
    n = 0
    for key, S, H, src, kw in make_golden_ref.gamg_runs(pkg, orc):
        x, p = H.solve(np.zeros(S.n), src, **kw)
        ref = G[key + "/perf"]
        assert np.array_equal(x, G[key + "/psi"]), key
        assert p["initialResidual"] == ref[0] and p["finalResidual"] == ref[1] and p["nIterations"] == int(ref[2]) and bool(p["converged"]) == bool(ref[3]), key
        n += 1
    assert n == 11
    if orc.ref_gamg_available():      # and live, where the reference build is present
        now = make_golden_ref.build_gamg(pkg, orc)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k


def test_inter_level_functors_equal_the_reference_headers(pkg, orc):
    """GAMGSolverAgglomerateMatrixF.H (sym/asym/diag agglomerate) and GAMGAgglomerationF.H (restrict, prolong) of the
    reference, compiled as host code and driven over the stably sorted restrict addressing as GAMGSolverAgglomerateMatrix.C
    does (oracle/_ref/libref_gamg_functors.so), produced tests/golden/golden_ref_gamg_functors.npz.  Sums of doubles in a
    fixed order: the oracle's Galerkin-by-summation level matrices and its restrict/prolong must give the SAME BITS,
    symmetric and asymmetric (flips), plain and merged levels."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_gamg_functors.npz"))
    n = 0
    for name, (case, w) in make_golden_ref.gamg_functor_cases(pkg, orc).items():
        H = orc.GamgHierarchy(case, orc.box_face_weights(case) if w is None else w, 6, merge_levels=2 if name.endswith("merge2") else 1)
        for l in range(min(3, H.n_levels)):
            lv = H.level(l)
            x = pkg.synthetic.splitmix_uniform(40 + l, lv["n_fine"]) - 0.5
            r = H.restrict(l, x)
            assert np.array_equal(r, G[f"{name}/{l}/restrict"]) and np.array_equal(H.prolong(l, r), G[f"{name}/{l}/prolong"])
            d, u, lo = H.coarse_matrix(l)
            assert np.array_equal(d, G[f"{name}/{l}/diag"]) and np.array_equal(u, G[f"{name}/{l}/upper"])
            if lo is not None:
                assert np.array_equal(lo, G[f"{name}/{l}/lower"])
            n += 1
    assert n >= 10
    if orc.ref_gamg_functors_available():
        now = make_golden_ref.build_gamg_functors(pkg, orc)
        assert set(now) == set(G.files)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k


def test_scale_update_equals_the_reference_functor(pkg, orc):
    """GAMGSolver::scale (GAMGSolverScale.C:59-171): field = sf*field + (source - sf*A field)/D with
    sf = sum(source*field)/stabilise(sum(A field*field)).  The reference's GAMGSolverScaleFunctor, compiled from
    GAMGSolverScale.C where it lies (oracle/_ref/libref_gamg_scale.so), produced tests/golden/golden_ref_gamg_scale.npz; the
    oracle's scale must give the SAME BITS on one, two and four domains, symmetric and asymmetric (the contraction of
    `sf*field + ...` and `source - sf*Acf` to fused multiply-adds included)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_gamg_scale.npz"))
    for name, subs in make_golden_ref.gamg_scale_cases(pkg, orc).items():
        S = orc.System(subs)
        field = pkg.synthetic.splitmix_uniform(61, S.n) - 0.5
        source = pkg.synthetic.splitmix_uniform(62, S.n) - 0.5
        scaled, acf = orc.gamg_sys_scale(S, field, source)
        assert np.array_equal(acf, S.amul(field))
        assert make_golden_ref.scale_factor(source, field, acf) == G[f"{name}/sf"][0]
        assert np.array_equal(scaled, G[f"{name}/field"]), name
        assert np.array_equal(source * field, G[f"{name}/terms"])
        assert not np.array_equal(scaled, field)
    if orc.ref_gamg_scale_available():
        now = make_golden_ref.build_gamg_scale(pkg, orc)
        assert set(now) == set(G.files)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k


@pytest.mark.parametrize("parts,merge", [((2, 1, 1), 1), ((2, 2, 1), 1), ((2, 2, 2), 1), ((1, 3, 2), 2)])
def test_per_rank_hierarchies_of_a_decomposed_case_equal_the_oracle(pkg, orc, parts, merge):
    """The per-rank GAMG builder a real rank runs (pair agglomeration per processor, `continueAgglomerating` agreed over all
    processors, restrict addressing of the patch cells exchanged with the neighbour, coarse processor interfaces from the
    distinct (mine, theirs) pairs) for EVERY domain of a decomposed box -- all domains in one process, a barrier and a shared
    table as the communicator -- against the oracle's multi-domain restatement: cell maps, coarse addressing and the
    agglomerated processor patches of every level and domain."""
    syn = pkg.synthetic
    case = syn.box_case(12, 10, 8)
    subs = syn.decompose_box(case, parts)
    ws = [orc.box_face_weights(s) for s in subs]
    S = orc.System(subs)
    H = orc.GamgSysHierarchy(S, ws, 6, merge_levels=merge)
    E = pkg.engine.gamg_host_hierarchy_domains(subs, ws, 6, True, merge_levels=merge)
    assert len(E) == len(subs)
    for d, sub in enumerate(subs):
        assert len(E[d]) == H.n_levels >= 2
        n_patch = [len(i.face_cells) for i in sub.interfaces]
        for l in range(H.n_levels):
            o, e = H.level(d, l), E[d][l]
            assert np.array_equal(o["restrict"], e["restrictMap"])
            assert np.array_equal(o["lower"], e["cLower"]) and np.array_equal(o["upper"], e["cUpper"])
            for p in range(len(sub.interfaces)):
                op = H.patch(d, l, p, n_patch[p])
                ep = e["patches"][p]
                assert np.array_equal(op["face_restrict"], ep["faceRestrict"]) and np.array_equal(op["face_cells"], ep["faceCells"])
                # both sides of a processor patch agree: my nbrCells are the neighbour's faceCells, face for face
                nd, npch = sub.interfaces[p].nbr_domain, sub.interfaces[p].nbr_patch
                assert np.array_equal(ep["nbrCells"], E[nd][l]["patches"][npch]["faceCells"])
                n_patch[p] = len(ep["faceCells"])


def test_per_rank_hierarchies_on_ragged_graphs_with_arbitrary_partitions(pkg, orc):
    """The same comparison on what decomposePar can produce for an unstructured mesh: ragged graphs cut by ARBITRARY
    cell-to-processor maps (contiguous chunks, interleaved stripes, random labels -> many small patches, several neighbours
    per domain, domains without a common boundary), symmetric and asymmetric, plain and merged levels.  The multi-domain
    system must also still be the global matrix (Amul of the pieces = Amul of the whole)."""
    syn = pkg.synthetic
    for seed, n, extra, nd, kind, merge in [(1, 300, 2.0, 3, "chunks", 1), (2, 500, 1.5, 4, "stripes", 1), (3, 400, 3.0, 5, "random", 1),
                                            (4, 700, 2.5, 2, "random", 2), (5, 260, 1.0, 6, "chunks", 3)]:
        case = random_graph_case(pkg, n, extra=extra, seed=seed, symmetric=(seed % 2 == 1))
        c = np.arange(n)
        dom = {"chunks": c * nd // n, "stripes": (c // 7) % nd, "random": (syn.splitmix_uniform(90 + seed, n) * nd).astype(np.int64)}[kind]
        subs = syn.decompose(case, dom, nd)
        S = orc.System(subs)
        x = syn.splitmix_uniform(seed, n) - 0.5
        xs = np.concatenate([x[s.global_cells] for s in subs])
        got = S.amul(xs)
        ref = orc.System([case]).amul(x)
        off = 0
        for s in subs:
            assert np.max(np.abs(got[off:off + s.n_cells] - ref[s.global_cells])) <= 1e-13 * np.max(np.abs(ref))
            off += s.n_cells
        w = 0.5 + syn.splitmix_uniform(77 + seed, case.n_faces)
        ws = [w[s.global_faces] for s in subs]
        H = orc.GamgSysHierarchy(S, ws, 4, merge_levels=merge)
        E = pkg.engine.gamg_host_hierarchy_domains(subs, ws, 4, True, merge_levels=merge)
        assert H.n_levels >= 1
        for d, sub in enumerate(subs):
            assert len(E[d]) == H.n_levels
            n_patch = [len(i.face_cells) for i in sub.interfaces]
            for l in range(H.n_levels):
                o, e = H.level(d, l), E[d][l]
                assert np.array_equal(o["restrict"], e["restrictMap"])
                assert np.array_equal(o["lower"], e["cLower"]) and np.array_equal(o["upper"], e["cUpper"])
                for p in range(len(sub.interfaces)):
                    op, ep = H.patch(d, l, p, n_patch[p]), e["patches"][p]
                    assert np.array_equal(op["face_restrict"], ep["faceRestrict"]) and np.array_equal(op["face_cells"], ep["faceCells"])
                    nbd, nbp = sub.interfaces[p].nbr_domain, sub.interfaces[p].nbr_patch
                    assert np.array_equal(ep["nbrCells"], E[nbd][l]["patches"][nbp]["faceCells"])
                    n_patch[p] = len(ep["faceCells"])
        # and the multi-domain GAMG solves the same problem as the single-domain one (different hierarchy, same answer)
        if case.lower is None:
            psi, p = H.solve(np.zeros(n), np.concatenate([case.source[s.global_cells] for s in subs]), tolerance=1e-10, maxIter=200)
            ref_psi = np.linalg.solve(_dense(case), case.source)
            assert p["converged"]
            off = 0
            for s in subs:
                assert np.max(np.abs(psi[off:off + s.n_cells] - ref_psi[s.global_cells])) < 1e-7 * np.max(np.abs(ref_psi))
                off += s.n_cells


def _dense(case):
    n = case.n_cells
    A = np.zeros((n, n)); A[np.arange(n), np.arange(n)] = case.diag
    lo = case.upper if case.lower is None else case.lower
    A[case.lower_addr, case.upper_addr] = case.upper; A[case.upper_addr, case.lower_addr] = lo
    return A


def test_merged_levels_galerkin_and_the_flip_rule(pkg, orc):
    """What combineLevels' flip rule (only the last pair step's flip survives) means for the level matrices: with a symmetric
    fine matrix the merged coarse matrix is exactly R A R^T; with an asymmetric one the diagonal and the SYMMETRIC part
    (upper + lower per coarse face) are still Galerkin, while upper and lower of a coarse face may be interchanged for the
    fine faces whose earlier flip was dropped -- the reference's behaviour, reproduced on purpose."""
    for sym in (True, False):
        case = pkg.synthetic.box_case(8, 7, 6, symmetric=sym)
        n = case.n_cells
        A = _dense(case)
        H = orc.GamgHierarchy(case, orc.box_face_weights(case), 4, merge_levels=2)
        Af = A
        for l in range(min(2, H.n_levels)):
            lv = H.level(l)
            R = np.zeros((lv["n_coarse"], lv["n_fine"])); R[lv["restrict"], np.arange(lv["n_fine"])] = 1.0
            G = R @ Af @ R.T                                     # Galerkin with piecewise-constant restriction
            d, u, lo = H.coarse_matrix(l)
            Ac = np.zeros_like(G); Ac[np.arange(len(d)), np.arange(len(d))] = d
            Ac[lv["lower"], lv["upper"]] = u; Ac[lv["upper"], lv["lower"]] = u if lo is None else lo
            tol = 1e-12 * np.max(np.abs(G))
            assert np.max(np.abs(np.diag(Ac) - np.diag(G))) < tol
            assert np.max(np.abs((Ac + Ac.T) - (G + G.T))) < tol              # symmetric part: always Galerkin
            if sym:
                assert np.max(np.abs(Ac - G)) < tol
            Af = Ac if sym else G                                               # next level from the true Galerkin operator would differ: stop after comparing
            if not sym:
                break


@pytest.mark.gpu
@pytest.mark.parametrize("symmetric", [True, False])
def test_dummy_agglomeration_levels_are_the_fine_mesh(pkg, orc, symmetric):
    """agglomerator dummy (dummyAgglomeration.C:45-90): nLevels identity levels.  Level sizes equal the fine mesh, the level
    matrices are the fine matrix (Galerkin sum over one child), and the V-cycle history follows the oracle's."""
    import torch
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    case = syn.box_case(9, 8, 7, symmetric=symmetric)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    G = eng.Gamg(addr, None, dummy_levels=3)
    H = orc.GamgHierarchy(case, None, dummy_levels=3)
    assert G.n_levels == H.n_levels == 3
    for l in range(3):
        s = G.level_sizes(l)
        assert s["n_fine"] == s["n_coarse"] == case.n_cells and s["n_coarse_faces"] == case.n_faces
    n = case.n_cells
    kw = dict(tolerance=1e-9, maxIter=30)
    ref_psi, ref = H.solve(np.zeros(n), case.source, **kw)
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, dev(case.source), **kw)
    assert perf["nIterations"] == ref["nIterations"] and ref["converged"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape and np.max(np.abs(h - hr)) < 1e-10 * hr[0]
    torch.cuda.synchronize()
    assert np.max(np.abs(psi.cpu().numpy() - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(nFinestSweeps=3), dict(nPreSweeps=1, nPostSweeps=1), dict(scaleCorrection=1)])
def test_cycle_graph_replay_equals_eager_cycles(pkg, orc, kw, monkeypatch):
    """The captured V-cycle (hipGraph, MI_GAMG_GRAPH) replays with the reference's DEFAULT sweep schedule as well -- 2, 3, 4, 4 ...
    post sweeps: odd counts used to switch the replay off -- and with odd finest / pre sweep counts; every cycle's
    residual equals the eagerly enqueued cycle's bit for bit, and both follow the oracle."""
    import torch
    eng = pkg.engine
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    case = pkg.synthetic.box_case(24, 20, 16, symmetric=True); w = orc.box_face_weights(case)
    args = dict(tolerance=1e-10, maxIter=40); args.update(kw)
    hist = {}
    for graph in ("1", "0"):
        monkeypatch.setenv("MI_GAMG_GRAPH", graph)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr); mat.set_coeffs(dev(case.diag), dev(case.upper), None)
        G = eng.Gamg(addr, w, 10)
        assert G.n_levels >= 4                      # levels 1.. have 3 and 4 post sweeps with the defaults
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        perf = G.solve(mat, psi, dev(case.source), **args)
        hist[graph] = (perf["history"], psi.cpu().numpy())
    assert np.array_equal(hist["1"][0], hist["0"][0]) and np.array_equal(hist["1"][1], hist["0"][1])
    _, ref = orc.GamgHierarchy(case, w, 10).solve(np.zeros(case.n_cells), case.source, **args)
    assert hist["1"][0].shape == ref["history"].shape and np.max(np.abs(hist["1"][0] - ref["history"])) < 1e-10 * ref["history"][0]


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("box_sym", dict()), ("box_sym", dict(nFinestSweeps=1)), ("box_sym", dict(nFinestSweeps=3, nPostSweeps=1)), ("box_asym", dict(scaleCorrection=1)),
                                     ("box_asym", dict()), ("graph_sym", dict()), ("box_big", dict())])
def test_fused_transfers_equal_the_separate_kernels_bit_for_bit(pkg, orc, name, kw, monkeypatch):
    """Round 6: the prolongation of a level is formed in the staging of the tile pass that consumes it (the Amul of the correction
    scaling, or the first smoothing sweep where a level is not scaled), the scaling pass in the staging of the first smoothing sweep,
    psi += finestCorrection in the staging of the first finest sweep; the two sums of the scaling factor are folded by the last
    workgroup of the Amul in the order k_fold_partials2 + sum_partials add them (MI_GAMG_FUSE, default on).  Same arithmetic, same
    summation order: residual history AND solution are bit-identical to the cycle of separate kernels, for symmetric and asymmetric
    matrices, ragged graphs, odd sweep counts, and a matrix with more tiles than reduction slots (box_big: 2 304 tiles > 1 024)."""
    import torch
    from conftest import random_graph_case
    eng = pkg.engine
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    if name == "graph_sym":
        case = random_graph_case(pkg, 6000, extra=2.0, seed=11); w = 0.5 + pkg.synthetic.splitmix_uniform(3, case.n_faces)
    else:
        dims = (160, 120, 120) if name == "box_big" else (40, 32, 24)
        case = pkg.synthetic.box_case(*dims, symmetric=name != "box_asym"); w = orc.box_face_weights(case)
    args = dict(tolerance=1e-9, maxIter=12); args.update(kw)
    got = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("MI_GAMG_FUSE", fuse)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        if name == "box_big":
            assert addr.n_tiles > 1024
        mat = eng.Matrix(addr); mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
        G = eng.Gamg(addr, w, 10)
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        perf = G.solve(mat, psi, dev(case.source), **args)
        got[fuse] = (perf["history"], psi.cpu().numpy(), perf["nIterations"])
    assert got["1"][2] == got["0"][2] and np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])


@pytest.mark.gpu
def test_cycle_graph_is_not_replayed_across_a_symmetric_to_asymmetric_rebind(pkg, orc):
    """the cached V-cycle graph captured tile kernels of the SYMMETRIC matrix; re-binding the same matrix
    handle with asymmetric coefficients (same hierarchy, same vectors, explicit scaleCorrection so that nothing else in the key
    changes) must not replay them -- the second solve has to be the asymmetric system's, cycle by cycle."""
    import torch
    eng = pkg.engine
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    sym = pkg.synthetic.box_case(24, 20, 16, symmetric=True)
    asym = pkg.synthetic.box_case(24, 20, 16, symmetric=False)
    w = orc.box_face_weights(sym)
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    addr = eng.Addressing(ctx, sym.n_cells, sym.lower_addr, sym.upper_addr)
    mat = eng.Matrix(addr)
    G = eng.Gamg(addr, w, 10)
    args = dict(tolerance=1e-10, maxIter=30, scaleCorrection=0, nPostSweeps=2, postSweepsLevelMultiplier=0)   # even sweeps everywhere
    for case in (sym, asym, sym):
        mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        perf = G.solve(mat, psi, dev(case.source), **args)
        ref_psi, ref = orc.GamgHierarchy(case, w, 10).solve(np.zeros(case.n_cells), case.source, **args)
        assert perf["nIterations"] == ref["nIterations"] and perf["nIterations"] >= 3
        assert np.max(np.abs(perf["history"] - ref["history"])) < 1e-10 * ref["history"][0]
        assert np.max(np.abs(psi.cpu().numpy() - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


@pytest.mark.gpu
@pytest.mark.parametrize("name,ncoarsest", [("box_sym", 10), ("box_sym", 60), ("box_asym", 120), ("graph_sym", 60)])
def test_register_resident_coarsest_inverse_equals_the_host_elimination(pkg, orc, name, ncoarsest, monkeypatch):
    """Round 3: the dense inverse of the coarsest level is formed in the REGISTERS of one workgroup (k_dense_invert_reg, in-place
    Gauss-Jordan with partial pivoting, up to 192 coarsest cells: the default there).  Element by element it performs the host
    elimination's operations (gamg.cpp invert_dense), so a solve gives the SAME BITS whichever path built the inverse -- and the
    global-memory kernel of larger systems agrees to rounding."""
    import torch
    eng = pkg.engine
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    if name == "graph_sym":
        case = random_graph_case(pkg, 2500); w = graph_weights(pkg, case)
    else:
        case = pkg.synthetic.box_case(24, 20, 16, symmetric=(name == "box_sym")); w = orc.box_face_weights(case)
    args = dict(tolerance=1e-10, maxIter=60)
    out = {}
    for mode, env in (("registers", {}), ("host", {"MI_GAMG_DEVICE_INVERT": "0"}), ("global", {"MI_GAMG_DEVICE_INVERT": "1", "MI_GAMG_REG_INVERT": "0"})):
        for k in ("MI_GAMG_DEVICE_INVERT", "MI_GAMG_REG_INVERT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr); mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
        G = eng.Gamg(addr, w, ncoarsest)
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        perf = G.solve(mat, psi, dev(case.source), **args)
        out[mode] = (perf["history"], psi.cpu().numpy(), G.level_sizes(G.n_levels - 1)["n_coarse"])
    assert out["registers"][2] <= 192
    assert np.array_equal(out["registers"][0], out["host"][0]) and np.array_equal(out["registers"][1], out["host"][1])
    assert out["global"][0].shape == out["host"][0].shape and np.max(np.abs(out["global"][0] - out["host"][0])) < 1e-10 * out["host"][0][0]
    _, ref = orc.GamgHierarchy(case, w, ncoarsest).solve(np.zeros(case.n_cells), case.source, **args)
    assert out["registers"][0].shape == ref["history"].shape and np.max(np.abs(out["registers"][0] - ref["history"])) < 1e-10 * ref["history"][0]


@pytest.mark.gpu
def test_level_matrices_straight_into_the_tile_slots(pkg, orc, monkeypatch):
    """Round 4: mi_gamg_update forms every level matrix in the coarse level's ENGINE arrays (tile slots,
    engine-order diagonal) from the fine level's engine arrays through slot-to-slot children lists (k_gamg_agg_slots /
    k_gamg_agg_diag_e), instead of caller-order arrays + mi_matrix_set_coeffs per level.  Same children, same order: the solve
    histories are BIT-identical to the round-3 path (MI_GAMG_DIRECT_SLOTS=0), symmetric and asymmetric, with coupled patches,
    and equal to the oracle's (1e-10)."""
    import torch
    eng, syn = pkg.engine, pkg.synthetic
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    def host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()
    out = {}
    for direct in ("1", "0"):
        monkeypatch.setenv("MI_GAMG_DIRECT_SLOTS", direct)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        for name, case in (("sym", syn.box_case(40, 32, 24)), ("asym", syn.box_case(24, 20, 16, symmetric=False)),
                           ("cyclic", syn.add_cyclic_y(syn.box_case(24, 20, 16)))):
            itfs = case.interfaces
            addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, [i.face_cells for i in itfs],
                                  [itfs[i.nbr_patch].face_cells for i in itfs] if itfs else ())
            mat = eng.Matrix(addr)
            mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
            for p, itf in enumerate(itfs):
                mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if case.lower is None else dev(itf.int_coeffs))
            w = orc.box_face_weights(case)
            G = eng.Gamg(addr, w, 10)
            psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
            perf = G.solve(mat, psi, dev(case.source), tolerance=1e-9, maxIter=60)
            # a re-bind with other coefficients, then back: the level matrices follow
            mat.set_coeffs(dev(1.5 * case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
            psi2 = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
            perf2 = G.solve(mat, psi2, dev(case.source), tolerance=1e-9, maxIter=60)
            out[direct, name] = (perf["history"], host(psi), perf2["history"])
            if direct == "1":
                H = orc.GamgSysHierarchy(orc.System([case]), [w], 10) if itfs else orc.GamgHierarchy(case, w, 10)
                ref_psi, ref = H.solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=60)
                assert perf["nIterations"] == ref["nIterations"] and np.max(np.abs(perf["history"] - ref["history"])) < 1e-10 * ref["history"][0], name
    for name in ("sym", "asym", "cyclic"):
        for k in range(3):
            assert np.array_equal(out["1", name][k], out["0", name][k]), (name, k)


@pytest.mark.gpu
@pytest.mark.parametrize("symmetric", [True, False])
def test_rebound_coefficients_step_after_step_with_both_inversion_kernels(pkg, orc, symmetric, monkeypatch):
    """Round 6: a sequence of 'time steps' -- new coefficients, solve: the graph-replayed agglomeration, the coarsest inversion on the
    side stream beside the (one-pass) prologue -- incl. a solve that converges in its prologue and is followed at once by a re-bind.
    k_dense_invert_reg2 (default) and k_dense_invert_reg (MI_GAMG_INVERT_V2=0) perform the same operations: SAME BITS step for step."""
    import torch
    eng, syn = pkg.engine, pkg.synthetic
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    case = syn.box_case(40, 32, 24, symmetric=symmetric)
    n = case.n_cells
    w = orc.box_face_weights(case)
    out = {}
    for mode, env in (("v2", {}), ("v1", {"MI_GAMG_INVERT_V2": "0"}), ("host", {"MI_GAMG_DEVICE_INVERT": "0"})):
        for k in ("MI_GAMG_DEVICE_INVERT", "MI_GAMG_INVERT_V2"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr)
        G = eng.Gamg(addr, w, 60)
        res = []
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        for step in range(5):
            f = 1.0 + 0.07 * step
            mat.set_coeffs(dev(case.diag * f - 0.01 * step), dev(case.upper * (2.0 - f)), None if symmetric else dev(case.lower * f))
            perf = G.solve(mat, psi, dev(case.source), tolerance=1e-9, maxIter=40)
            res.append((perf["history"], perf["nIterations"], psi.cpu().numpy().copy()))
            if step == 2:   # converged already: the prologue ends the solve; then straight into the next re-bind
                p2 = G.solve(mat, psi, dev(case.source), tolerance=1e-3, maxIter=40)
                assert p2["nIterations"] == 0
                mat.set_coeffs(dev(case.diag * 3.0), dev(case.upper * 0.5), None if symmetric else dev(case.lower * 0.5))
                p3 = G.solve(mat, psi, dev(case.source), tolerance=1e-3, maxIter=40)
                res.append((p3["history"], p3["nIterations"], psi.cpu().numpy().copy()))
        out[mode] = res
        del G, mat, addr, ctx
    for mode in ("v1", "host"):
        for (ha, na, xa), (hb, nb, xb) in zip(out["v2"], out[mode]):
            assert na == nb and np.array_equal(ha, hb) and np.array_equal(xa, xb), mode
    assert out["v2"][0][1] > 2
