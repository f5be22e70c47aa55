"""GPU (-m gpu): the solvers' prologue as ONE pass over the coefficients (round 6).

Every solver starts with  wA = A psi, rA = source - wA, normFactor (sumA, gAverage(psi)), the initial residual and the first
convergence test (PCG.C:91-121, PBiCG.C:91-131, PBiCGStab.C:91-128, GAMGSolverSolve.C:59-110, lduMatrixSolver.C:182-236).  As
separate launches that is a tile pass for A psi, a tile pass for sumA (the coefficients are re-bound before every solve of a time
step), a subtraction, two reductions, the normFactor pass and -- single right-hand side -- a host read for gAverage(psi).
tile_kernel<OP_PROLOGUE> / tile_kernel_multi<OP_PROLOGUE> run the Amul's fma chain and the sumA pass's chain of additions side by
side over one staging of the tile and store rA with them; k_normfactor_mag forms normFactor's and gSumMag(rA)'s partials in one
vector pass in the per-thread order of their own kernels; the average stays on the device.  Nothing may change: normFactor,
initial residual, every entry of the history, the iteration count and psi must equal the separate passes' BIT FOR BIT -- every
solver, symmetric and asymmetric, tiles with coupled patches, a non-zero initial psi, solves with and without re-bound
coefficients, per-component diagonals, a communicator attached (one-rank RCCL and peer windows: the prologue's sums are all-reduced)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    assert torch.cuda.is_available(), "GPU tests need a device"
    assert pkg.engine.device_available(), "HIP engine sees no gfx950 device"
    c = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
    yield c
    torch.cuda.synchronize()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def _same(pa, pb, what):
    for k in ("nIterations", "converged", "singular", "initialResidual", "finalResidual", "normFactor"):
        assert pa[k] == pb[k] or (np.isnan(pa[k]) and np.isnan(pb[k])), (what, k, pa[k], pb[k])
    assert np.array_equal(pa["history"], pb["history"], equal_nan=True), what


def _bind(mat, case):
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(getattr(case, "interfaces", []) or []):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if case.lower is None else dev(itf.int_coeffs))


def _cases(pkg):
    syn = pkg.synthetic
    return {
        "box_sym": syn.box_case(21, 17, 13),
        "box_asym": syn.box_case(21, 17, 13, symmetric=False),
        "box_30tiles_sym": syn.box_case(40, 32, 24),
        "box_30tiles_asym": syn.box_case(40, 32, 24, symmetric=False),
        "cyclic_sym": syn.add_cyclic_y(syn.box_case(18, 12, 10)),
        "cyclic_asym": syn.add_cyclic_y(syn.box_case(18, 12, 10, symmetric=False), asym_shift=0.25),
        "one_cell": syn.box_case(1, 1, 1),
        "line": syn.box_case(70, 1, 1),
    }


def _solvers(mat, symmetric):
    s = []
    if symmetric:
        s += [("pcg diagonal", lambda p, b, **kw: mat.pcg(p, b, "diagonal", **kw)), ("pcg AINV", lambda p, b, **kw: mat.pcg(p, b, "AINV", **kw))]
    s += [("pbicg AINV", lambda p, b, **kw: mat.pbicg(p, b, "AINV", **kw)), ("pbicg diagonal", lambda p, b, **kw: mat.pbicg(p, b, "diagonal", **kw)),
          ("pbicgstab AINV", lambda p, b, **kw: mat.pbicgstab(p, b, "AINV", **kw)), ("smooth", lambda p, b, **kw: mat.smooth_solve(p, b, n_sweeps=2, **kw))]
    return s


@pytest.mark.parametrize("name", ["box_sym", "box_asym", "box_30tiles_sym", "box_30tiles_asym", "cyclic_sym", "cyclic_asym", "one_cell", "line"])
def test_fused_prologue_equals_the_separate_passes_bit_for_bit(pkg, ctx, name):
    eng, syn = pkg.engine, pkg.synthetic
    case = _cases(pkg)[name]
    n = case.n_cells
    sym = case.lower is None
    itf = getattr(case, "interfaces", []) or []
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr, [i.face_cells for i in itf], patch_nbr_cells=[itf[i.nbr_patch].face_cells for i in itf]) if itf \
        else eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    psi0 = 0.3 * (syn.splitmix_uniform(5, n) - 0.5)          # a non-zero start: A psi, gAverage(psi) and sumA all matter
    kw = dict(tolerance=1e-9, maxIter=60)
    try:
        for what, fn in _solvers(mat, sym):
            out = {}
            for fuse in (0, 1):
                ctx.set_option("fuse_prologue", fuse)
                res = []
                for start, rebind in ((psi0, True), (np.zeros(n), True), (psi0, False)):   # (False: sumA is valid from the solve before)
                    if rebind:
                        _bind(mat, case)
                    psi = dev(start.copy())
                    res.append((fn(psi, dev(case.source), **kw), host(psi)))
                out[fuse] = res
            for (pa, xa), (pb, xb) in zip(out[0], out[1]):
                _same(pa, pb, (name, what))
                assert np.array_equal(xa, xb, equal_nan=True), (name, what)
            assert out[1][0][0]["normFactor"] > 0 and np.isfinite(out[1][0][0]["initialResidual"])
    finally:
        ctx.set_option("fuse_prologue", 1)


@pytest.mark.parametrize("name", ["box_asym", "box_30tiles_asym", "box_sym"])
@pytest.mark.parametrize("precond", ["AINV", "diagonal"])
def test_fused_prologue_of_the_multi_vector_solver(pkg, ctx, name, precond):
    """mi_pbicg_solve_multi: three components, shared and per-component diagonals (sumA per component out of the same pass), two and one"""
    eng, syn = pkg.engine, pkg.synthetic
    case = _cases(pkg)[name]
    n = case.n_cells
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    srcs = [case.source, 3.0 * (syn.splitmix_uniform(41, n) - 0.5), 0.5 * (syn.splitmix_uniform(42, n) - 0.5)]
    starts = [0.3 * (syn.splitmix_uniform(60 + c, n) - 0.5) for c in range(3)]
    diags = [case.diag * (1.0 + 0.05 * c) + 0.01 * c * syn.splitmix_uniform(50 + c, n) for c in range(3)]
    kw = dict(tolerance=1e-9, maxIter=40)
    try:
        out = {}
        for fuse in (0, 1):
            ctx.set_option("fuse_prologue", fuse)
            res = []
            for nrhs, dg, rebind in ((3, None, True), (3, diags, True), (2, diags, True), (1, None, True), (3, None, False), (1, diags, False)):
                if rebind:
                    _bind(mat, case)
                psis = [dev(starts[c].copy()) for c in range(nrhs)]
                got = mat.pbicg_multi(psis, [dev(b) for b in srcs[:nrhs]], precond, diags=None if dg is None else [dev(d) for d in dg[:nrhs]], **kw)
                res.append((got, [host(p) for p in psis]))
            out[fuse] = res
        for (ga, xa), (gb, xb) in zip(out[0], out[1]):
            for c in range(len(ga)):
                _same(ga[c], gb[c], (name, precond, c))
                assert np.array_equal(xa[c], xb[c]), (name, precond, c)
        # and the per-component result is the single solve's on the re-bound matrix (which takes tile_kernel<OP_PROLOGUE>)
        ctx.set_option("fuse_prologue", 1)
        got, xs = out[1][1]
        for c in range(3):
            cc = copy.copy(case); cc.diag = diags[c]
            m2 = eng.Matrix(addr); _bind(m2, cc)
            psi = dev(starts[c].copy())
            ref = m2.pbicg(psi, dev(srcs[c]), precond, **kw)
            _same(got[c], ref, (name, precond, c, "single"))
            assert np.array_equal(xs[c], host(psi))
    finally:
        ctx.set_option("fuse_prologue", 1)


@pytest.mark.parametrize("symmetric", [True, False])
def test_fused_prologue_of_gamg(pkg, orc, ctx, symmetric):
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.box_case(40, 32, 24, symmetric=symmetric)
    n = case.n_cells
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    G = eng.Gamg(addr, orc.box_face_weights(case), 10)
    psi0 = 0.3 * (syn.splitmix_uniform(5, n) - 0.5)
    out = {}
    try:
        for fuse in (0, 1):
            ctx.set_option("fuse_prologue", fuse)
            res = []
            for start in (psi0, np.zeros(n)):
                _bind(mat, case)
                psi = dev(start.copy())
                res.append((G.solve(mat, psi, dev(case.source), tolerance=1e-8, maxIter=30), host(psi)))
            out[fuse] = res
        for (pa, xa), (pb, xb) in zip(out[0], out[1]):
            _same(pa, pb, ("gamg", symmetric))
            assert np.array_equal(xa, xb)
        assert out[1][0][0]["nIterations"] > 2
    finally:
        ctx.set_option("fuse_prologue", 1)


@pytest.mark.parametrize("mode", ["rccl", "peer"])
@pytest.mark.parametrize("symmetric", [True, False])
def test_fused_prologue_with_a_communicator_attached(pkg, ctx, symmetric, mode):
    """a one-rank communicator whose processor patches face this rank: the halo of psi is exchanged inside the prologue's tile pass
    (send / recv, or the one-launch window form) and its three sums are all-reduced -- the code an N-rank run executes"""
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.add_cyclic_y(syn.box_case(40, 32, 24, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)
    n = case.n_cells
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr, [i.face_cells for i in case.interfaces])
    mat = eng.Matrix(addr)
    _bind(mat, case)
    comm = eng.Comm(ctx, 1, 0, eng.Comm.unique_id())
    if mode == "peer":
        assert comm.peer_auto()
    mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=n)
    assert mat.peer_halo_status()[0] == (mode == "peer")
    psi0 = 0.3 * (syn.splitmix_uniform(5, n) - 0.5)
    kw = dict(tolerance=1e-9, maxIter=40)
    try:
        for what, fn in _solvers(mat, symmetric):
            out = {}
            for fuse in (0, 1):
                ctx.set_option("fuse_prologue", fuse)
                _bind(mat, case)
                psi = dev(psi0.copy())
                out[fuse] = (fn(psi, dev(case.source), **kw), host(psi))
            _same(out[0][0], out[1][0], (mode, what))
            assert np.array_equal(out[0][1], out[1][1]), (mode, what)
        if not symmetric:
            srcs = [case.source, 3.0 * (syn.splitmix_uniform(41, n) - 0.5)]
            out = {}
            for fuse in (0, 1):
                ctx.set_option("fuse_prologue", fuse)
                _bind(mat, case)
                psis = [dev(psi0.copy()), dev(np.zeros(n))]
                out[fuse] = (mat.pbicg_multi(psis, [dev(b) for b in srcs], "AINV", **kw), [host(p) for p in psis])
            for c in range(2):
                _same(out[0][0][c], out[1][0][c], (mode, "multi", c))
                assert np.array_equal(out[0][1][c], out[1][1][c])
    finally:
        ctx.set_option("fuse_prologue", 1)
        mat.detach_comm()
        comm.close()
