"""cyclicAMI interfaces and transformed coupled patches: neighbour values interpolated with the AMI
weights of the patch's side, transformCoupleField factor applied first, low-weight faces on their own cell's value.

CPU: the oracle's interpolation against the REFERENCE's AMIInterpolationF.H functors compiled in place (oracle/_ref), and
against a plain numpy reading.  GPU: every operator bit for bit, whole solvers, against the oracle on a non-conformal
box interface whose two sides have different face counts (cyclicAMIFvPatchField.C:195-224, cyclicAMIGAMGInterfaceField.C:97-130)."""
import numpy as np
import pytest


def naive_amul(pkg, orc, case, x):
    """numpy reading of the interface update on top of the interface-free product"""
    import copy
    c0 = copy.copy(case); c0.interfaces = []
    y = orc.System([c0]).amul(x)
    for itf in case.interfaces:
        ot = case.interfaces[itf.nbr_patch]
        f = itf.transform * x[ot.face_cells]
        for i in range(itf.face_cells.shape[0]):
            if itf.ami_low is not None and itf.ami_low[i]:
                pn = x[itf.face_cells[i]]
            else:
                sl = slice(itf.ami_start[i], itf.ami_start[i + 1])
                pn = float(np.dot(itf.ami_w[sl], f[itf.ami_addr[sl]]))
            y[itf.face_cells[i]] -= itf.bou_coeffs[i] * pn
    return y


def test_oracle_ami_interpolation_is_the_references_functor(pkg, orc):
    if not orc.ref_ami_available():
        pytest.skip("oracle/_ref/libref_ami.so not built (needs /root/reference)")
    syn = pkg.synthetic
    case = syn.add_cyclic_ami_y(syn.box_case(7, 5, 4), shift=0.41)
    x = syn.splitmix_uniform(4, case.n_cells) - 0.5
    # the oracle's pnf through a unit-coefficient interface on a zero matrix: Amul = -pnf scattered onto faceCells
    for p, itf in enumerate(case.interfaces):
        ot = case.interfaces[itf.nbr_patch]
        ref = orc.ref_ami_interpolate(itf.ami_start, itf.ami_addr, itf.ami_w, x[ot.face_cells])
        mine = np.array([np.nan] * itf.face_cells.shape[0])
        for i in range(mine.shape[0]):                       # the oracle's chain, restated: out = fma(w, f, out) in address order
            acc = 0.0
            for k in range(itf.ami_start[i], itf.ami_start[i + 1]):
                acc = float(np.float64(np.longdouble(itf.ami_w[k]) * np.longdouble(x[ot.face_cells[itf.ami_addr[k]]]) + np.longdouble(acc)))
            mine[i] = acc
        # long-double fma emulation is exact to 64 bits of mantissa: may differ from a true fma in rare double roundings
        assert np.max(np.abs(ref - mine)) <= 2e-16 * np.max(np.abs(ref))
        # low-weight correction: faces under the threshold return the default value, the others the sum
        ws = np.add.reduceat(itf.ami_w, itf.ami_start[:-1])
        thr = np.sort(ws)[ws.shape[0] // 3]
        dflt = x[itf.face_cells]
        ref_low = orc.ref_ami_interpolate(itf.ami_start, itf.ami_addr, itf.ami_w, x[ot.face_cells], thr, ws, dflt)
        low = ws < thr
        assert low.any() and not low.all()
        assert np.array_equal(ref_low[low], dflt[low]) and np.array_equal(ref_low[~low], ref[~low])
    # and the oracle's operator IS that chain: compare bitwise through a system whose only entries are the interface's
    import copy
    z = copy.copy(case)
    z.diag = np.zeros(case.n_cells); z.upper = np.zeros(case.n_faces); z.lower = None
    z.interfaces = [copy.copy(i) for i in case.interfaces]
    for itf in z.interfaces:
        itf.bou_coeffs = np.ones_like(itf.bou_coeffs); itf.int_coeffs = np.ones_like(itf.int_coeffs)
    # one face per cell on side 0: Amul[faceCells[i]] = -pnf[i] exactly
    y = orc.System([z]).amul(x)
    itf, ot = z.interfaces[0], z.interfaces[1]
    ref = orc.ref_ami_interpolate(itf.ami_start, itf.ami_addr, itf.ami_w, x[ot.face_cells])
    assert np.array_equal(y[itf.face_cells], -ref)


def test_oracle_ami_operator_against_numpy_and_unit_weights_reduce_to_cyclic(pkg, orc):
    syn = pkg.synthetic
    for sym in (True, False):
        case = syn.add_cyclic_ami_y(syn.box_case(6, 5, 4, symmetric=sym), low_weight_every=5, transform=-0.75)
        x = syn.splitmix_uniform(1, case.n_cells) - 0.5
        y = orc.System([case]).amul(x)
        assert np.max(np.abs(y - naive_amul(pkg, orc, case, x))) < 1e-15 * np.max(np.abs(y)) * 10
    # one-to-one addressing with unit weights and factor 1 is the plain cyclic interface, bit for bit
    import copy
    cyc = syn.add_cyclic_y(syn.box_case(6, 5, 4))
    ami = copy.copy(cyc); ami.interfaces = [copy.copy(i) for i in cyc.interfaces]
    for itf in ami.interfaces:
        n = itf.face_cells.shape[0]
        itf.ami_start, itf.ami_addr, itf.ami_w = np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), np.ones(n)
    x = syn.splitmix_uniform(2, cyc.n_cells) - 0.5
    assert np.array_equal(orc.System([ami]).amul(x), orc.System([cyc]).amul(x))
    assert np.array_equal(orc.System([ami]).jacobi_smooth(x, cyc.source, 2), orc.System([cyc]).jacobi_smooth(x, cyc.source, 2))


# ---- GPU -----------------------------------------------------------------------------------------------------------------

def _engine(pkg, ctx, case, ordered=False):
    import torch
    eng = pkg.engine
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    fcs = [i.face_cells for i in case.interfaces]
    kw = {}
    if ordered:
        a0 = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs)
        case = pkg.synthetic.renumber(case, a0.cell_perm())
        fcs = [i.face_cells for i in case.interfaces]
        kw = dict(ordered=True, tile_cell_start=a0.tile_starts())
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs, **kw)      # no neighbour cells: ext region
    for p, itf in enumerate(case.interfaces):
        if itf.ami_start is not None:
            addr.set_ami_patch(p, itf.nbr_patch, itf.ami_start, itf.ami_addr, itf.ami_w, itf.ami_low)
        else:
            addr.set_ami_patch(p, itf.nbr_patch)
        if itf.ami_magsf is not None:
            addr.set_ami_face_areas(p, itf.ami_magsf)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), dev(itf.int_coeffs))
        if itf.transform != 1.0:
            mat.set_patch_transform(p, itf.transform)
    return case, addr, mat


def _hist(perf, ref):
    assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == ref["converged"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape and np.max(np.abs(h - hr)) < 1e-10 * hr[0]


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "low_weight", "transformed", "ordered"])
@pytest.mark.parametrize("symmetric", [True, False])
def test_engine_cyclic_ami_bit_exact_and_solvers(pkg, orc, symmetric, variant):
    import torch
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    base = syn.box_case(18, 12, 10, symmetric=symmetric)
    case = syn.add_cyclic_ami_y(base, shift=0.37, low_weight_every=7 if variant == "low_weight" else 0,
                                transform=0.6 if variant == "transformed" else 1.0)
    case, addr, mat = _engine(pkg, ctx, case, ordered=(variant == "ordered"))
    assert addr.n_ext == sum(i.face_cells.shape[0] for i in case.interfaces)
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(3, n) - 0.5
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
    mat.sumA(out); assert np.array_equal(host(out), S.sumA())
    mat.residual(dev(x), dev(case.source), out); assert np.array_equal(host(out), S.residual(x, case.source))
    mat.H(dev(x), out); assert np.array_equal(host(out), S.H(x))
    for sweeps in (1, 2, 3):
        psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), sweeps)
        assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, sweeps))
    # patchNeighbourField: the interpolated (and transformed) partner values, what enters result -= coeffs*pnf
    nbr = torch.empty(addr.n_ext, dtype=torch.float64, device="cuda:0")
    mat.patch_neighbour_field(dev(x), nbr)
    got, off = host(nbr), 0
    for itf in case.interfaces:
        ot = case.interfaces[itf.nbr_patch]
        m = itf.face_cells.shape[0]
        for i in (0, m // 2, m - 1):
            if itf.ami_low is not None and itf.ami_low[i]:
                assert got[off + i] == x[itf.face_cells[i]]
            else:
                sl = slice(itf.ami_start[i], itf.ami_start[i + 1])
                want = float(np.dot(itf.ami_w[sl], itf.transform * x[ot.face_cells[itf.ami_addr[sl]]]))
                assert abs(got[off + i] - want) < 1e-14
        off += m
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    src = dev(case.source)
    if symmetric and variant != "transformed":   # (the transformed case goes through the bi-conjugate solvers below)
        for pre in ("diagonal", "AINV"):
            psi.zero_()
            perf = mat.pcg(psi, src, pre, tolerance=1e-9, maxIter=500)
            ref_psi, ref = S.pcg(np.zeros(n), case.source, pre, tolerance=1e-9, maxIter=500)
            _hist(perf, ref)
            assert np.max(np.abs(host(psi) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))
    else:
        # (tolerance 1e-8: bi-conjugate residuals wander, and at 1e-10 the two histories -- equal to 1e-10 of the initial
        #  residual -- can sit on either side of the threshold in the last iteration)
        if not symmetric:  # (the bi-conjugate solvers on the symmetric-but-transformed system -- no longer symmetric, hardly
            #                   preconditioned -- have near-breakdowns that amplify rounding differences to 1e-3 of the residual
            #                   mid-way; that variant keeps the operators, the smoother and GAMG (test below) as its checks)
            perf = mat.pbicg(psi, src, "DILU", tolerance=1e-8, maxIter=400)
            ref_psi, ref = S.pbicg(np.zeros(n), case.source, "AINV", tolerance=1e-8, maxIter=400)
            _hist(perf, ref)
            assert np.max(np.abs(host(psi) - ref_psi)) < 1e-7 * np.max(np.abs(ref_psi))
            psi.zero_()
            perf = mat.pbicgstab(psi, src, "diagonal", tolerance=1e-8, maxIter=400)
            ref_psi, ref = S.pbicgstab(np.zeros(n), case.source, "diagonal", tolerance=1e-8, maxIter=400)
            _hist(perf, ref)
    psi.zero_()
    perf = mat.smooth_solve(psi, src, n_sweeps=2, tolerance=1e-3, maxIter=60)
    ref_psi, ref = S.smooth_solve(np.zeros(n), case.source, n_sweeps=2, tolerance=1e-3, maxIter=60)
    _hist(perf, ref)


@pytest.mark.gpu
def test_engine_one_to_one_patch_with_unit_weights_is_the_cyclic_patch(pkg, orc):
    """mi_addr_set_ami_patch without tables = a cyclic patch that can carry a transformation factor: with factor 1 the same
    bits as the cyclic patch created with neighbour cells; with a factor, the oracle's transformed interface"""
    import copy, torch
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    cyc = syn.add_cyclic_y(syn.box_case(18, 12, 10))
    x = syn.splitmix_uniform(9, cyc.n_cells) - 0.5
    out = torch.empty(cyc.n_cells, dtype=torch.float64, device="cuda:0")
    for factor in (1.0, 0.5, -1.0):
        case = copy.copy(cyc); case.interfaces = [copy.copy(i) for i in cyc.interfaces]
        for itf in case.interfaces:
            itf.transform = factor
        _, addr, mat = _engine(pkg, ctx, case)
        S = orc.System([case])
        mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
        psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 2)
        assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 2))
        if factor == 1.0:
            assert np.array_equal(host(out), orc.System([cyc]).amul(x))
    # a plain cyclic patch refuses a factor (it has no place to apply it): the error says how to declare it
    fcs = [i.face_cells for i in cyc.interfaces]
    nbrs = [cyc.interfaces[i.nbr_patch].face_cells for i in cyc.interfaces]
    mat = eng.Matrix(eng.Addressing(ctx, cyc.n_cells, cyc.lower_addr, cyc.upper_addr, fcs, nbrs))
    mat.set_patch_transform(0, 1.0)
    with pytest.raises(eng.MiError):
        mat.set_patch_transform(0, -1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "transformed"])
@pytest.mark.parametrize("symmetric", [True, False])
def test_engine_gamg_with_agglomerated_ami(pkg, orc, symmetric, variant):
    """GAMG on a mesh with a cyclicAMI pair: every level carries the agglomerated AMI (cyclicAMIGAMGInterface.C:47-165,
    AMIInterpolation::agglomerate), the coarsest level is solved by ICCG / BICCG (the reference's direct coarsest solver
    only knows cyclic interfaces).  Cycle-by-cycle residuals against the oracle's hierarchy."""
    import torch
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    base = syn.box_case(20, 16, 12, symmetric=symmetric)
    case = syn.add_cyclic_ami_y(base, shift=0.37, transform=0.6 if variant == "transformed" else 1.0)
    case, addr, mat = _engine(pkg, ctx, case)
    w = orc.box_face_weights(base)
    S = orc.System([case])
    H = orc.GamgSysHierarchy(S, [w], 10)
    G = eng.Gamg(addr, w, 10)
    assert G.n_levels == H.n_levels and H.n_levels >= 4
    n = case.n_cells
    kw = dict(tolerance=1e-9, maxIter=60, directSolveCoarsest=False)
    ref_psi, ref = H.solve(np.zeros(n), case.source, **kw)
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = G.solve(mat, psi, dev(case.source), **kw)
    assert ref["converged"]
    _hist(perf, ref)
    assert np.max(np.abs(host(psi) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))
    with pytest.raises(eng.MiError):                       # as in the reference: no direct coarsest solver with cyclicAMI interfaces
        G.solve(mat, psi, dev(case.source), tolerance=1e-9, maxIter=5, directSolveCoarsest=True)


def test_oracle_ami_agglomeration_follows_the_reference_loop(pkg, orc):
    """AMIInterpolation::agglomerate + normaliseWeights (AMIInterpolation.C:199-247,279-540) and the face agglomeration of
    cyclicAMIGAMGInterface.C:66-110, restated here in plain Python straight from the reference's loops, against the oracle's C
    (oracle/gamg_oracle.c: ami_agglomerate) on every level of a hierarchy: same coarse faces, same addresses in the same order,
    the same weights bit for bit (host arithmetic: a product, an addition, one division per weight)."""
    syn = pkg.synthetic
    base = syn.box_case(12, 10, 8)
    case = syn.add_cyclic_ami_y(base, shift=0.37)
    S = orc.System([case])
    H = orc.GamgSysHierarchy(S, [orc.box_face_weights(base)], 10)
    assert H.n_levels >= 3
    fine = [dict(start=i.ami_start, addr=i.ami_addr, w=i.ami_w, magsf=i.ami_magsf, cells=i.face_cells) for i in case.interfaces]
    for l in range(H.n_levels):
        rmap = H.level(0, l)["restrict"]
        # cyclicAMIGAMGInterface.C:66-110: coarse face per distinct local coarse cell, order of first appearance
        face_restrict, coarse_cells = [], []
        for f in fine:
            seen, fr, cc = {}, [], []
            for c in f["cells"]:
                m = int(rmap[c])
                if m not in seen:
                    seen[m] = len(cc); cc.append(m)
                fr.append(seen[m])
            face_restrict.append(np.array(fr)); coarse_cells.append(np.array(cc, np.int32))
        coarse = []
        for p, f in enumerate(fine):
            src_r, tgt_r = face_restrict[p], face_restrict[1 - p]
            nc = coarse_cells[p].shape[0]
            mag = np.zeros(nc)
            for i in range(src_r.shape[0]):                       # "Agglomerate face areas"
                mag[src_r[i]] += f["magsf"][i]
            elems, weights = [[] for _ in range(nc)], [[] for _ in range(nc)]
            for i in range(src_r.shape[0]):                       # "Agglomerate weights and indices" (no distribution map)
                fine_area = f["magsf"][i]
                I = src_r[i]
                for k in range(f["start"][i], f["start"][i + 1]):
                    K = int(tgt_r[f["addr"][k]])
                    if K in elems[I]:
                        weights[I][elems[I].index(K)] += fine_area * f["w"][k]
                    else:
                        elems[I].append(K); weights[I].append(fine_area * f["w"][k])
            start, addr, w = [0], [], []
            for I in range(nc):                                   # normaliseWeights, conformal: denom = sum(w)
                s = 0.0
                for v in weights[I]:
                    s += v
                addr += elems[I]; w += [v / s for v in weights[I]]
                start.append(len(addr))
            coarse.append(dict(start=np.array(start, np.int32), addr=np.array(addr, np.int32), w=np.array(w), magsf=mag, cells=coarse_cells[p]))
        for p in range(2):
            got_patch = H.patch(0, l, p, fine[p]["cells"].shape[0])
            assert np.array_equal(got_patch["face_restrict"], face_restrict[p]) and np.array_equal(got_patch["face_cells"], coarse_cells[p])
            got = H.patch_ami(0, l, p, coarse_cells[p].shape[0])
            assert np.array_equal(got["start"], coarse[p]["start"]) and np.array_equal(got["addr"], coarse[p]["addr"])
            assert np.array_equal(got["w"], coarse[p]["w"]) and np.array_equal(got["magsf"], coarse[p]["magsf"])
            ws = np.add.reduceat(got["w"], got["start"][:-1])
            assert np.max(np.abs(ws - 1.0)) < 1e-14               # every coarse face's weights sum to one
        fine = coarse


def test_engine_host_builder_agglomerates_the_ami_like_the_oracle(pkg, orc):
    """the ENGINE's host-side hierarchy builder (csrc/gamg.cpp, no device needed) against the oracle's, level by level: the same
    coarse patch faces, the same addresses in the same order, the same weights and face areas bit for bit"""
    import copy
    syn, eng = pkg.synthetic, pkg.engine
    for sym, shift in ((True, 0.37), (False, 0.61)):
        base = syn.box_case(14, 10, 8, symmetric=sym)
        case = syn.add_cyclic_ami_y(base, shift=shift)
        w = orc.box_face_weights(base)
        H = orc.GamgSysHierarchy(orc.System([case]), [w], 10)
        L = eng.gamg_host_hierarchy_ami(case, w, 10)
        assert len(L) == H.n_levels >= 3
        n_fine = [i.face_cells.shape[0] for i in case.interfaces]
        for l in range(H.n_levels):
            assert np.array_equal(L[l]["restrictMap"], H.level(0, l)["restrict"])
            for p in range(2):
                ref = H.patch(0, l, p, n_fine[p])
                got = L[l]["patches"][p]
                assert np.array_equal(got["faceRestrict"], ref["face_restrict"]) and np.array_equal(got["faceCells"], ref["face_cells"])
                A = H.patch_ami(0, l, p, ref["face_cells"].shape[0])
                assert np.array_equal(got["amiStart"], A["start"]) and np.array_equal(got["amiAddr"], A["addr"])
                assert np.array_equal(got["amiW"], A["w"]) and np.array_equal(got["amiMagSf"], A["magsf"])
            n_fine = [L[l]["patches"][p]["faceCells"].shape[0] for p in range(2)]


@pytest.mark.parametrize("py", [2, 4])
@pytest.mark.parametrize("symmetric", [True, False])
def test_oracle_with_the_ami_sides_in_different_domains_agrees_with_the_single_domain_oracle(pkg, orc, py, symmetric):
    """The multi-domain oracle is what the cyclicAMI-across-ranks tests of tests/test_distributed.py compare the engine with
    (AMIInterpolation.C:940-1091: the partner patch lives on another processor).  Pin it here, on the CPU: the y-slab
    decomposition of synthetic.decompose_cyclic_ami_y, its two AMI sides in the first and the last domain, against the SAME
    case as one domain (the form tests/golden and test_ami's GPU tests pin) -- Amul to round-off (the row sums of the cut
    cells are ordered differently), Krylov histories to 1e-10 of the initial residual, same iteration counts."""
    syn = pkg.synthetic
    base = syn.box_case(14, 12, 10, symmetric=symmetric)
    kw = dict(shift=0.37, low_weight_every=(0 if symmetric else 7), transform=(1.0 if symmetric else 0.6))
    one = syn.add_cyclic_ami_y(base, **kw)
    subs = syn.decompose_cyclic_ami_y(base, py, **kw)
    assert len(subs) == py and sum(s.n_cells for s in subs) == one.n_cells
    assert subs[0].interfaces[-1].nbr_domain == py - 1 and subs[-1].interfaces[-1].nbr_domain == 0
    S1, S = orc.System([one]), orc.System(subs)
    cells = np.concatenate([s.global_cells for s in subs])
    x = syn.splitmix_uniform(3, one.n_cells) - 0.5
    a1, a = S1.amul(x), S.amul(x[cells])
    assert np.max(np.abs(a - a1[cells])) < 1e-13 * np.max(np.abs(a1))
    src = one.source
    solvers = [("pcg", dict(precond="diagonal")), ("pcg", dict(precond="AINV"))] if symmetric else [("pbicg", dict(precond="AINV")), ("pbicgstab", dict(precond="diagonal"))]
    for name, skw in solvers:
        p1, r1 = getattr(S1, name)(np.zeros(one.n_cells), src, tolerance=1e-9, maxIter=300, **skw)
        p, r = getattr(S, name)(np.zeros(one.n_cells), src[cells], tolerance=1e-9, maxIter=300, **skw)
        if skw.get("precond") == "AINV":   # DIC / DILU of the decomposed case is block-local (as in the reference): another preconditioner
            assert r["converged"] and r1["converged"]
            assert np.max(np.abs(p - p1[cells])) < 1e-6 * np.max(np.abs(p1))
            continue
        assert r["nIterations"] == r1["nIterations"], (name, r["nIterations"], r1["nIterations"])
        assert np.max(np.abs(r["history"] - r1["history"])) < 1e-10 * r1["history"][0], name
        assert np.max(np.abs(p - p1[cells])) < 1e-8 * np.max(np.abs(p1))


@pytest.mark.parametrize("symmetric,kw,px", [(True, dict(shift=0.37), 2), (False, dict(shift=1.61, low_weight_every=7, transform=0.6), 2), (True, dict(shift=2.3), 3)])
def test_oracle_split_sides_equal_the_single_domain_interface(pkg, orc, symmetric, kw, px):
    """CPU: the multi-domain oracle with BOTH cyclicAMI sides split over px domains each (orc_sys_set_iface_ami_parts: addresses
    numbering the partner pieces' faces concatenated) against the single-domain interface, which the reference's own
    AMIInterpolationF.H pins (tests/test_oracle.py): every piece addresses several partner pieces; Amul / Tmul agree BIT FOR BIT on
    every row that no processor patch touches (those rows add a processor-interface term where the single domain adds a face term:
    another summation order, rounding-level difference), PCG / PBiCG histories and the GAMG solve agree to rounding."""
    import copy
    syn = pkg.synthetic
    base = syn.box_case(12, 8, 5, symmetric=symmetric)
    full = syn.add_cyclic_ami_y(base, **kw)
    subs = syn.decompose_cyclic_ami_split(base, px, **kw)
    assert len(subs) == 2 * px and all(len(s.interfaces[-1].ami_parts) >= 2 for s in subs)      # every piece talks to several partner pieces
    for s in subs:
        itf = s.interfaces[-1]
        assert itf.ami_addr.max() < sum(itf.ami_part_sizes) and len(set(q for q, _ in itf.ami_parts)) == len(itf.ami_parts)
    S1, SN = orc.System([full]), orc.System(subs)
    glob = np.concatenate([s.global_cells for s in subs])
    x = syn.splitmix_uniform(3, base.n_cells) - 0.5
    touched, off = set(), 0
    for s in subs:
        for itf in s.interfaces[:-1]:
            touched |= set((off + itf.face_cells).tolist())
        off += s.n_cells
    clean = np.array(sorted(set(range(base.n_cells)) - touched))
    for op in ("amul", "tmul"):
        y1, yn = getattr(S1, op)(x)[glob], getattr(SN, op)(x[glob])
        assert np.array_equal(y1[clean], yn[clean]) and np.max(np.abs(y1 - yn)) < 4e-16 * np.max(np.abs(y1))
    ami_rows, off = set(), 0
    for s in subs:
        ami_rows |= set((off + s.interfaces[-1].face_cells).tolist()); off += s.n_cells
    assert len(ami_rows - touched) > 20                       # ... and most cyclicAMI rows are among the bit-exact ones
    src = np.concatenate([s.source for s in subs])
    z = np.zeros(base.n_cells)
    solve = (lambda S, b: S.pcg(z, b, "diagonal", tolerance=1e-10, maxIter=400)) if symmetric else (lambda S, b: S.pbicg(z, b, "diagonal", tolerance=1e-10, maxIter=300))   # (DILU is block-local on a decomposed case: another preconditioner)
    (p1, f1), (pn, fn) = solve(S1, full.source), solve(SN, src)
    assert abs(f1["nIterations"] - fn["nIterations"]) <= 1 and np.max(np.abs(p1[glob] - pn)) < 1e-8 * np.max(np.abs(p1))
    b0 = copy.copy(full); b0.interfaces = []
    g1 = orc.GamgSysHierarchy(S1, [orc.box_face_weights(b0)], 6).solve(z, full.source, tolerance=1e-9, maxIter=80, directSolveCoarsest=False)
    gn = orc.GamgSysHierarchy(SN, [orc.box_face_weights(s) for s in subs], 6).solve(z, src, tolerance=1e-9, maxIter=80, directSolveCoarsest=False)
    assert gn[1]["converged"] and g1[1]["converged"] and np.max(np.abs(g1[0][glob] - gn[0])) < 1e-7 * np.max(np.abs(g1[0]))
