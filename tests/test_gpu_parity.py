"""GPU (-m gpu): the HIP engine, called through the C ABI, against the CPU oracle.

Bars (north_star): SpMV-family outputs are BIT-EXACT against the oracle (same fma chain, same
order); reductions agree to 1e-14 relative (different but deterministic tree); solver residual
histories agree to 1e-10 relative with identical iteration counts.
"""
import os

import numpy as np
import pytest
import torch

from conftest import random_graph_case

pytestmark = pytest.mark.gpu

HIST_RTOL = 1e-10


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
    return t.cpu().numpy()


def make(pkg, ctx, case):
    eng = pkg.engine
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    return addr, mat


def cases(pkg):
    syn = pkg.synthetic
    return {
        "box_sym": syn.box_case(21, 17, 13),
        "box_asym": syn.box_case(21, 17, 13, symmetric=False),
        "box_2tiles": syn.box_case(16, 16, 8),
        "graph_sym": random_graph_case(pkg, 3000, symmetric=True),
        "graph_asym": random_graph_case(pkg, 3000, symmetric=False),
        "one_cell": syn.box_case(1, 1, 1),
        "two_cells": syn.box_case(2, 1, 1),
        "line": syn.box_case(70, 1, 1),
    }


@pytest.mark.parametrize("name", ["box_sym", "box_asym", "box_2tiles", "graph_sym", "graph_asym", "one_cell", "two_cells", "line"])
def test_spmv_family_bit_exact(pkg, orc, ctx, name):
    case = cases(pkg)[name]
    addr, mat = make(pkg, ctx, case)
    S = orc.System([case])
    n = case.n_cells
    x = pkg.synthetic.splitmix_uniform(99, n) - 0.5
    xd, bd = dev(x), dev(case.source)
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(xd, out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(xd, out); assert np.array_equal(host(out), S.tmul(x))
    mat.sumA(out); assert np.array_equal(host(out), S.sumA())
    mat.residual(xd, bd, out); assert np.array_equal(host(out), S.residual(x, case.source))
    mat.H(xd, out); assert np.array_equal(host(out), S.H(x))
    mat.H1(out); assert np.array_equal(host(out), S.H1())
    if case.n_faces:
        fh = torch.empty(case.n_faces, dtype=torch.float64, device="cuda:0")
        mat.faceH(xd, fh)
        assert np.array_equal(host(fh), S.faceH(x))
    # engine-order round trip
    xe = torch.empty(n, dtype=torch.float64, device="cuda:0")
    back = torch.empty(n, dtype=torch.float64, device="cuda:0")
    addr.to_engine(xd, xe); addr.from_engine(xe, back)
    assert np.array_equal(host(back), x)
    assert np.array_equal(host(xe), x[addr.cell_perm()])


@pytest.mark.parametrize("name", ["box_sym", "box_asym", "graph_asym"])
def test_preconditioners_and_smoother_bit_exact(pkg, orc, ctx, name):
    case = cases(pkg)[name]
    addr, mat = make(pkg, ctx, case)
    S = orc.System([case])
    n = case.n_cells
    r = pkg.synthetic.splitmix_uniform(7, n) - 0.5
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    for kind in ("none", "diagonal", "AINV"):
        for tr in (False, True):
            mat.precondition(kind, dev(r), out, transpose=tr)
            assert np.array_equal(host(out), S.precondition(kind, r, transpose=tr)), (kind, tr)
    psi = dev(r.copy())
    mat.jacobi_smooth(psi, dev(case.source), 3, omega=0.9)
    assert np.array_equal(host(psi), S.jacobi_smooth(r, case.source, 3, omega=0.9))


def test_coefficient_rebind_and_sym_to_asym(pkg, orc, ctx):
    syn = pkg.synthetic
    a, b = syn.box_case(15, 14, 9), syn.box_case(15, 14, 9, symmetric=False)
    addr, mat = make(pkg, ctx, a)
    x = syn.splitmix_uniform(1, a.n_cells)
    out = torch.empty(a.n_cells, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), orc.System([a]).amul(x))
    mat.set_coeffs(dev(b.diag), dev(b.upper), dev(b.lower))  # "coefficients changed" epoch
    mat.amul(dev(x), out); assert np.array_equal(host(out), orc.System([b]).amul(x))
    mat.tmul(dev(x), out); assert np.array_equal(host(out), orc.System([b]).tmul(x))
    mat.set_coeffs(dev(a.diag), dev(a.upper), None)
    mat.amul(dev(x), out); assert np.array_equal(host(out), orc.System([a]).amul(x))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 100003])
def test_reductions(pkg, orc, ctx, n):
    a = pkg.synthetic.splitmix_uniform(5, n) - 0.5
    b = pkg.synthetic.splitmix_uniform(6, n) - 0.5
    ad, bd = dev(a), dev(b)
    ref_sum = float(np.sum(a.astype(np.longdouble)))
    ref_mag = float(np.sum(np.abs(a).astype(np.longdouble)))
    ref_dot = float(np.sum(a.astype(np.longdouble) * b.astype(np.longdouble)))
    tol = 1e-14 * max(ref_mag, 1e-300)
    assert abs(ctx.sum(ad) - ref_sum) <= tol
    assert abs(ctx.sum_mag(ad) - ref_mag) <= tol
    assert abs(ctx.sum_prod(ad, bd) - ref_dot) <= tol
    # deterministic: bitwise identical run to run
    assert ctx.sum_prod(ad, bd) == ctx.sum_prod(ad, bd)


# Solves that may run through the persistent kernel (csrc/persist.inc): its sums are grouped per workgroup, not per 1024-cell
# chunk like the five-launch loop's, and CG amplifies that rounding difference as the residual falls.  The bar follows what the
# kernel delivers (VERDICT r04 "next" 2; gpurun_out/parity_small_observed.json keeps the observed figures, profiles/ the round's
# copy): every history entry within 1e-6 RELATIVE of the oracle's, except where the residual has fallen so far that 1e-6 of it is
# below 2e-13 of the INITIAL residual -- there the absolute deviation must stay under that floor (seen: 6.8e-14 at a residual of
# 1e-9, i.e. 5e-5 relative, box_sym / none; 1.1e-12 relative at 108^3 down to 1e-8).
PERSIST_REL = 1e-6
PERSIST_FLOOR = 2e-13


def _observe(name, h, hr, rel, floor):
    """append the observed deviations of one history comparison to gpurun_out/parity_small_observed.json"""
    import json
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    f = os.path.join(out, "parity_small_observed.json")
    try:
        d = json.load(open(f))
    except Exception:
        d = {}
    dev_abs = np.abs(h - hr) / hr[0]
    dev_rel = np.abs(h - hr) / np.maximum(np.abs(hr), 1e-300)
    above = np.abs(hr) * rel >= floor * hr[0] if floor > 0 else np.ones(hr.shape, bool)
    d[name] = dict(iterations=int(h.shape[0] - 1), final_residual_over_initial=float(hr[-1] / hr[0]), max_dev_over_initial=float(dev_abs.max()),
                   max_rel_dev_any_iteration=float(dev_rel.max()), max_rel_dev_where_the_relative_bar_applies=float(dev_rel[above].max()) if above.any() else 0.0,
                   max_dev_over_initial_below_it=float(dev_abs[~above].max()) if (~above).any() else 0.0, rel_bar=rel, floor_over_initial=floor)
    json.dump(d, open(f, "w"), indent=1, sort_keys=True)


def _check_hist(perf, ref, rel=1e-5, floor=0.0):
    """rel: per-iteration relative bar over the WHOLE history.  1e-5 for every pipeline whose sums are grouped like the
    five-launch loop's (they were bit-identical to each other before the persistent kernel existed); solves that may take the
    persistent kernel pass rel=PERSIST_REL, floor=PERSIST_FLOOR: |h - hr| < max(rel * hr, floor * hr[0]) entry by entry."""
    assert perf["nIterations"] == ref["nIterations"]
    assert perf["converged"] == ref["converged"] and perf["singular"] == ref["singular"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape
    if floor > 0:
        _observe(os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].split(" ")[0], h, hr, rel, floor)
    # north_star bar: residual histories within 1e-10 relative (to the normalised initial residual;
    # CG amplifies last-bit differences of the reduction tree as the residual falls, so the
    # per-iteration relative check is looser -- the oracle's own serial-vs-decomposed drift is 1e-8)
    assert np.max(np.abs(h - hr)) < HIST_RTOL * hr[0]
    assert np.max(np.abs(h[:10] - hr[:10]) / np.maximum(np.abs(hr[:10]), 1e-300)) < HIST_RTOL
    assert np.all(np.abs(h - hr) < np.maximum(rel * np.abs(hr), floor * hr[0]))
    assert abs(perf["normFactor"] - ref["normFactor"]) < 1e-13 * ref["normFactor"]


PERSIST = dict(rel=PERSIST_REL, floor=PERSIST_FLOOR)


@pytest.mark.parametrize("precond", ["none", "diagonal", "AINV", "DIC"])
@pytest.mark.parametrize("name", ["box_sym", "graph_sym"])
def test_pcg_residual_history(pkg, orc, ctx, name, precond):
    case = cases(pkg)[name]
    _, mat = make(pkg, ctx, case)
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = mat.pcg(psi, dev(case.source), precond, tolerance=1e-9, maxIter=400)
    ref_psi, ref = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, precond, tolerance=1e-9, maxIter=400)
    _check_hist(perf, ref, **(PERSIST if precond in ("none", "diagonal") else {}))   # (small matrix, diagonal / none: the persistent kernel)
    assert np.max(np.abs(host(psi) - ref_psi)) < 1e-10 * np.max(np.abs(ref_psi))


def test_pcg_controls(pkg, orc, ctx):
    case = cases(pkg)["box_sym"]
    _, mat = make(pkg, ctx, case)
    S = orc.System([case])
    z = np.zeros(case.n_cells)
    for kw in (dict(tolerance=0.0, maxIter=5), dict(tolerance=1e30, maxIter=50, minIter=3), dict(tolerance=1e30, maxIter=50),
               dict(tolerance=0.0, relTol=0.01, maxIter=300), dict(tolerance=0.0, maxIter=37)):
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        perf = mat.pcg(psi, dev(case.source), "diagonal", **kw)
        ref_psi, ref = S.pcg(z, case.source, "diagonal", **kw)
        _check_hist(perf, ref, **PERSIST)
        assert np.max(np.abs(host(psi) - ref_psi)) <= 1e-10 * max(np.max(np.abs(ref_psi)), 1e-300)
    # non-zero initial guess
    x0 = pkg.synthetic.splitmix_uniform(8, case.n_cells) * 1e-3
    psi = dev(x0.copy())
    perf = mat.pcg(psi, dev(case.source), "diagonal", tolerance=1e-8)
    _, ref = S.pcg(x0, case.source, "diagonal", tolerance=1e-8)
    _check_hist(perf, ref, **PERSIST)
    # singular: zero residual => wApA == 0 => break without counting the iteration
    zero = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = mat.pcg(zero.clone(), zero, "diagonal", tolerance=0.0, maxIter=5)
    assert perf["singular"] == 1 and perf["nIterations"] == 0


@pytest.mark.parametrize("precond", ["diagonal", "AINV"])
@pytest.mark.parametrize("solver", ["pbicg", "pbicgstab", "pbicgstab_textbook"])
def test_asymmetric_solver_histories(pkg, orc, ctx, solver, precond):
    case = cases(pkg)["box_asym"]
    _, mat = make(pkg, ctx, case)
    S = orc.System([case])
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    kw = dict(tolerance=1e-10, maxIter=200)
    if solver == "pbicg":
        perf = mat.pbicg(psi, dev(case.source), precond, **kw)
        ref_psi, ref = S.pbicg(np.zeros(case.n_cells), case.source, precond, **kw)
    else:
        quirk = solver == "pbicgstab"
        perf = mat.pbicgstab(psi, dev(case.source), precond, replicate_quirk=quirk, **kw)
        ref_psi, ref = S.pbicgstab(np.zeros(case.n_cells), case.source, precond, replicate_quirk=quirk, **kw)
    _check_hist(perf, ref)
    assert np.max(np.abs(host(psi) - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


def test_smooth_solver(pkg, orc, ctx):
    case = pkg.synthetic.box_case(12, 11, 10, dirichlet_all=True)
    _, mat = make(pkg, ctx, case)
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = mat.smooth_solve(psi, dev(case.source), n_sweeps=2, tolerance=1e-4, maxIter=200)
    ref_psi, ref = orc.System([case]).smooth_solve(np.zeros(case.n_cells), case.source, n_sweeps=2, tolerance=1e-4, maxIter=200)
    _check_hist(perf, ref)
    assert np.max(np.abs(host(psi) - ref_psi)) < 1e-11 * np.max(np.abs(ref_psi))


def test_full_size_properties(pkg, ctx):
    """BASELINE config 2 size (216^3): size-independent properties instead of the (slow) oracle."""
    syn = pkg.synthetic
    case = syn.box_case(216, 216, 216)
    n = case.n_cells
    addr, mat = make(pkg, ctx, case)
    x = dev(syn.splitmix_uniform(31, n) - 0.5)
    y = dev(syn.splitmix_uniform(32, n) - 0.5)
    Ax = torch.empty_like(x); Ay = torch.empty_like(x); Axy = torch.empty_like(x)
    mat.amul(x, Ax); mat.amul(y, Ay); mat.amul(x + y, Axy)
    scale = float(torch.max(torch.abs(Ax)))
    # linearity and symmetry (x'Ay == y'Ax)
    assert float(torch.max(torch.abs(Axy - Ax - Ay))) < 1e-13 * scale
    assert abs(ctx.sum_prod(y, Ax) - ctx.sum_prod(x, Ay)) < 1e-11 * abs(ctx.sum_prod(x, Ax))
    # A*1 == sumA == diag + row sums of off-diagonals: exact row-sum identity
    ones = torch.ones(n, dtype=torch.float64, device="cuda:0")
    A1 = torch.empty_like(x); sA = torch.empty_like(x)
    mat.amul(ones, A1); mat.sumA(sA)
    assert float(torch.max(torch.abs(A1 - sA))) < 1e-14 * scale
    # sampled rows against a direct numpy evaluation of the LDU definition
    rows = np.unique((syn.splitmix_uniform(33, 4000) * n).astype(np.int64))
    lo, up = case.lower_addr, case.upper_addr
    xh = host(x)
    ref = case.diag[rows] * xh[rows]
    sel = np.isin(lo, rows); idx = np.searchsorted(rows, lo[sel]); np.add.at(ref, idx, case.upper[sel] * xh[up[sel]])
    sel = np.isin(up, rows); idx = np.searchsorted(rows, up[sel]); np.add.at(ref, idx, case.upper[sel] * xh[lo[sel]])
    assert np.max(np.abs(host(Ax)[rows] - ref)) < 1e-14 * scale
    # PCG: residual reported by the solver equals the true residual of the returned psi
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    b = dev(case.source)
    perf = mat.pcg(psi, b, "diagonal", tolerance=0.0, maxIter=60)
    assert perf["nIterations"] == 61 and np.all(np.isfinite(perf["history"]))
    r = torch.empty_like(psi)
    mat.residual(psi, b, r)
    true = ctx.sum_mag(r) / perf["normFactor"]
    assert abs(true - perf["finalResidual"]) < 1e-8 * perf["finalResidual"]
    assert perf["finalResidual"] < perf["initialResidual"]
    # deterministic: a second run gives the same bits
    psi2 = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf2 = mat.pcg(psi2, b, "diagonal", tolerance=0.0, maxIter=60)
    assert np.array_equal(perf2["history"], perf["history"]) and torch.equal(psi, psi2)


@pytest.mark.parametrize("parts", [(2, 1, 1), (2, 2, 2)])
def test_decomposed_pcg_on_one_gpu(pkg, orc, ctx, parts):
    """N sub-domains emulated on one GPU: the distributed phases (mi_dpcg_phase), interface slots,
    interior/boundary tile split and halo pack, with the exchange done by device copies and the
    all-reduce by summing the ranks' scalar blocks -- against the serial oracle."""
    case = pkg.synthetic.box_case(20, 18, 14)
    kw = dict(tolerance=1e-9, relTol=0.0, maxIter=300, minIter=0)
    ref_psi, ref = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e-9, maxIter=300)
    run_decomposed_pcg(pkg, case, parts, kw, ref_psi, ref)


def run_decomposed_pcg(pkg, case, parts, kw, ref_psi, ref):
    from importlib import import_module
    import __graft_entry__ as graft
    par = import_module(graft.PKG_NAME + ".parallel")
    syn = pkg.synthetic
    subs = syn.decompose_box(case, parts)
    # one engine context per emulated rank (each owns its device-side solver state), same stream
    ctxs = [pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream) for _ in subs]
    ops = [par.HipOps(c, s, torch.device("cuda:0")) for c, s in zip(ctxs, subs)]

    def exchange(field):
        for d, s in enumerate(subs):
            for k, itf in enumerate(s.interfaces):
                o = ops[itf.nbr_domain]
                a, b = int(o.offsets[itf.nbr_patch]), int(o.offsets[itf.nbr_patch + 1])
                dst = getattr(ops[d], field)
                a2, b2 = int(ops[d].offsets[k]), int(ops[d].offsets[k + 1])
                dst[ops[d].n + a2: ops[d].n + b2].copy_(o.send[a:b])

    def allreduce(sl):
        tot = sum(o.scal[sl] for o in ops)
        for o in ops:
            o.scal[sl] = tot

    for o in ops:
        o.set_initial(None); o.begin(history_len=kw["maxIter"] + 2, **kw); o.phase(0)
    exchange("psi")
    for o in ops:
        o.phase(1)
    allreduce(slice(3, 4))
    avg = float(ops[0].scal[3].item()) / case.n_cells
    for o in ops:
        o.phase(2, 0, avg)
    allreduce(slice(0, 2)); allreduce(slice(4, 5))
    for o in ops:
        o.phase(3)
    it = 0
    while it <= kw["maxIter"]:
        for _ in range(8):
            for o in ops:
                o.phase(10, it)
            exchange("pA")
            for o in ops:
                o.phase(11, it); o.phase(12, it)
            allreduce(slice(2, 3))
            for o in ops:
                o.phase(13, it)
            allreduce(slice(0, 2))
            it += 1
        for o in ops:
            o.phase(14, it - 1)
        if all(o.status()["done"] for o in ops):
            break
    psi = np.zeros(case.n_cells)
    for o, s in zip(ops, subs):
        st = o.status(kw["maxIter"] + 2)
        assert st["nIterations"] == ref["nIterations"] and st["converged"] == ref["converged"]
        assert st["history"].shape == ref["history"].shape
        assert np.max(np.abs(st["history"] - ref["history"])) < HIST_RTOL * ref["history"][0]
        psi[s.global_cells] = o.solution()
    from full_size_ref import check_solution
    check_solution(psi, ref_psi, 1e-9)       # ref_psi: the oracle's vector, or its committed record (tests/full_size_ref.py)


def test_error_behaviour_of_the_abi(pkg, ctx):
    """status codes instead of the reference's abort(): bad arguments, unbound matrix, bad addressing."""
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.box_case(6, 5, 4)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    x = dev(syn.splitmix_uniform(1, case.n_cells)); y = torch.empty_like(x)
    with pytest.raises(eng.MiError, match="not bound"):
        mat.amul(x, y)                                   # MI_ERR_STATE: coefficients never bound
    with pytest.raises(eng.MiError, match="not bound"):
        mat.pcg(y, x)
    with pytest.raises(eng.MiError, match="lowerAddr"):
        eng.Addressing(ctx, 4, np.array([3], np.int32), np.array([1], np.int32))
    mat.set_coeffs(dev(case.diag), dev(case.upper), None)
    with pytest.raises(eng.MiError, match="unknown preconditioner|bad argument"):
        pkg.engine._chk(pkg.engine.lib().mi_precondition(mat.h, 99, 0, pkg.engine._ptr(x), pkg.engine._ptr(y)))
    with pytest.raises(eng.MiError, match="bad argument"):
        pkg.engine._chk(pkg.engine.lib().mi_amul(mat.h, None, pkg.engine._ptr(y)))
    with pytest.raises(eng.MiError):
        eng.Gamg(addr, np.ones(case.n_faces), n_cells_in_coarsest_level=10 ** 6)   # "No coarse levels created" (GAMGSolver.C:175-190)


def test_star_mesh_long_rows(pkg, orc, ctx):
    # rows with 300 faces: exercises the beyond-register-prefetch path of the tile kernel
    n = 301
    lo = np.zeros(n - 1, np.int32); up = np.arange(1, n, dtype=np.int32)
    syn = pkg.synthetic
    upper = -(0.1 + syn.splitmix_uniform(1, n - 1)); lower = -(0.1 + syn.splitmix_uniform(4, n - 1))
    diag = np.zeros(n); np.subtract.at(diag, lo, lower); np.subtract.at(diag, up, upper); diag += 0.5
    case = syn.LduCase(n, lo, up, diag, upper, lower, syn.splitmix_uniform(2, n))
    _, mat = make(pkg, ctx, case)
    S = orc.System([case])
    x = syn.splitmix_uniform(3, n)
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = mat.pbicgstab(psi, dev(case.source), "DILU", tolerance=1e-12, maxIter=100)
    _, ref = S.pbicgstab(np.zeros(n), case.source, "AINV", tolerance=1e-12, maxIter=100)
    assert perf["nIterations"] == ref["nIterations"]


@pytest.mark.parametrize("symmetric", [True, False])
def test_cyclic_interfaces(pkg, orc, ctx, symmetric):
    """cyclic (local coupled) patches: every operator and whole solvers, single process, against the oracle."""
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.add_cyclic_y(syn.box_case(18, 12, 10, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)
    fcs = [i.face_cells for i in case.interfaces]
    nbrs = [case.interfaces[i.nbr_patch].face_cells for i in case.interfaces]
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs, nbrs)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if symmetric else dev(itf.int_coeffs))
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(3, n) - 0.5
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
    mat.sumA(out); assert np.array_equal(host(out), S.sumA())
    mat.residual(dev(x), dev(case.source), out); assert np.array_equal(host(out), S.residual(x, case.source))
    nbr = torch.empty(sum(len(f) for f in fcs), dtype=torch.float64, device="cuda:0")
    mat.patch_neighbour_field(dev(x), nbr)                                   # cyclic: the partner patch's internal values
    assert np.array_equal(host(nbr), np.concatenate([x[q] for q in nbrs]))
    mat.H(dev(x), out); assert np.array_equal(host(out), S.H(x))           # H and H1 are face sums: no interface terms
    mat.H1(out); assert np.array_equal(host(out), S.H1())
    psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 2)
    assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 2))   # interface terms enter bPrime, like the reference
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    if symmetric:
        perf = mat.pcg(psi, dev(case.source), "diagonal", tolerance=1e-9, maxIter=500)
        ref_psi, ref = S.pcg(np.zeros(n), case.source, "diagonal", tolerance=1e-9, maxIter=500)
    else:
        mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
        perf = mat.pbicg(psi, dev(case.source), "DILU", tolerance=1e-10, maxIter=300)
        ref_psi, ref = S.pbicg(np.zeros(n), case.source, "AINV", tolerance=1e-10, maxIter=300)
    _check_hist(perf, ref, **PERSIST)
    assert np.max(np.abs(host(psi) - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


@pytest.mark.parametrize("symmetric", [True, False])
def test_compact_row_entries_opt_in(pkg, orc, ctx, symmetric, monkeypatch):
    """MI_ENTRY16=1: 16-bit row entries with implied slots (half the addressing bytes) -- the same bits as the
    explicit form for every operator, with face and interface (cyclic) terms, AINV and Jacobi included."""
    syn, eng = pkg.synthetic, pkg.engine
    monkeypatch.setenv("MI_ENTRY16", "1")
    case = syn.add_cyclic_y(syn.box_case(18, 12, 10, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)
    fcs = [i.face_cells for i in case.interfaces]
    nbrs = [case.interfaces[i.nbr_patch].face_cells for i in case.interfaces]
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs, nbrs)
    monkeypatch.delenv("MI_ENTRY16")
    ref_addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, fcs, nbrs)
    assert addr.stats()["entries"] < 0.6 * ref_addr.stats()["entries"]      # the compact form is in use
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if symmetric else dev(itf.int_coeffs))
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(3, n) - 0.5
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
    mat.sumA(out); assert np.array_equal(host(out), S.sumA())
    mat.residual(dev(x), dev(case.source), out); assert np.array_equal(host(out), S.residual(x, case.source))
    mat.H(dev(x), out); assert np.array_equal(host(out), S.H(x))
    mat.H1(out); assert np.array_equal(host(out), S.H1())
    mat.precondition("AINV", dev(x), out); assert np.array_equal(host(out), S.precondition("AINV", x))
    psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 2)
    assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 2))
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    if symmetric:
        perf = mat.pcg(psi, dev(case.source), "AINV", tolerance=1e-9, maxIter=500)
        ref_psi, ref = S.pcg(np.zeros(n), case.source, "AINV", tolerance=1e-9, maxIter=500)
    else:
        perf = mat.pbicgstab(psi, dev(case.source), "AINV", tolerance=1e-10, maxIter=300)
        ref_psi, ref = S.pbicgstab(np.zeros(n), case.source, "AINV", tolerance=1e-10, maxIter=300)
    _check_hist(perf, ref)
    assert np.max(np.abs(host(psi) - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


def test_native_rccl_loop_self_exchange(pkg, orc, ctx):
    """C++ host loop over RCCL (mi_dpcg_comm_*) on a 1-rank communicator: the y-periodic box is posed with two
    PROCESSOR patches whose neighbour rank is this rank, so pack -> ncclSend/ncclRecv -> ext region -> boundary
    tiles and the scalar all-reduces all run for real; the oracle solves the same system as one domain."""
    syn, par = pkg.synthetic, pkg.parallel
    case = syn.add_cyclic_y(syn.box_case(20, 16, 12, symmetric=True))
    S = orc.System([case])
    ref_psi, ref = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e-9, maxIter=400)
    solver = par.DistributedPCG(ctx, case, "cuda:0", precond="diagonal", n_global=case.n_cells)
    assert solver.driver == "native"
    st = solver.solve(tolerance=1e-9, max_iter=400)
    _check_hist(st, ref)
    got = solver.ops.solution()
    assert np.max(np.abs(got - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))
    # raw all-reduce on the communicator
    t = torch.arange(5, dtype=torch.float64, device="cuda:0")
    solver.comms[0].allreduce_sum(t); torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(5, dtype=torch.float64))


@pytest.mark.parametrize("dims", [(96, 80, 72), (21, 17, 13)])
def test_pipelined_multi_vector_pass_equals_the_plain_one_bit_for_bit(pkg, dims, monkeypatch):
    """csrc/multi_pipe.inc (MI_MULTI_PIPE=1): the multi-vector tile passes of the Krylov iterations as persistent workgroups that
    prefetch their next tile into registers while they walk the current one.  Only WHEN a tile's image is loaded differs from
    tile_kernel_multi: the three-component PBiCG + DILU / + diagonal solves, the paired single-component solve and PBiCGStab must
    give the same bits -- histories, iteration counts, solutions -- on a matrix with several tiles per resident workgroup (553 k
    cells = 540 tiles on 256 workgroups) and on one with two tiles (first-tile staging only)."""
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.box_case(*dims, symmetric=False)
    n = case.n_cells
    srcs = [case.source, 3.0 * (syn.splitmix_uniform(41, n) - 0.5), syn.splitmix_uniform(42, n) - 0.5]
    out = {}
    for pipe in ("0", "1"):
        monkeypatch.setenv("MI_MULTI_PIPE", pipe)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
        if dims[0] > 50:
            assert addr.n_tiles > 2 * 256
        mat = eng.Matrix(addr)
        mat.set_coeffs(dev(case.diag), dev(case.upper), dev(case.lower))
        res = []
        for precond in ("DILU", "diagonal"):
            psis = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
            got = mat.pbicg_multi(psis, [dev(b) for b in srcs], precond, tolerance=0.0, maxIter=9)
            res += [(g["history"], g["nIterations"], host(q)) for g, q in zip(got, psis)]
            psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            g = mat.pbicg(psi, dev(srcs[1]), precond, tolerance=1e-7, maxIter=200)
            res.append((g["history"], g["nIterations"], host(psi)))
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        g = mat.pbicgstab(psi, dev(srcs[0]), "DILU", tolerance=0.0, maxIter=8)
        res.append((g["history"], g["nIterations"], host(psi)))
        out[pipe] = res
        mat.close(); addr.close(); ctx.close()
    assert len(out["0"]) == len(out["1"]) == 9
    for k, ((h0, n0, p0), (h1, n1, p1)) in enumerate(zip(out["0"], out["1"])):
        assert n0 == n1 and np.array_equal(h0, h1) and np.array_equal(p0, p1), k
    assert np.all(np.isfinite(out["1"][0][0])) and out["1"][0][0][-1] < out["1"][0][0][0]


@pytest.mark.parametrize("name", ["box_asym", "graph_asym", "box_sym"])
@pytest.mark.parametrize("precond", ["AINV", "diagonal", "none"])
def test_multi_rhs_pbicg_equals_the_single_solves_bit_for_bit(pkg, orc, ctx, name, precond, monkeypatch):
    """mi_pbicg_solve_multi (fvMatrix<vector>::solveSegregated as one solve, fvMatrixSolve.C:103-225): three right-hand sides --
    one of them converging at once, the others after different iteration counts -- through tile_kernel_multi (one staging of
    upper / lower for the six operand vectors of a step).  Per component: the SAME BITS (history, iteration count, psi) as
    mi_pbicg_solve on that component alone, which in turn follows the oracle to 1e-10; also with per-component DIAGONALS (the
    boundary contribution solveSegregated adds per component) against single solves of the re-bound matrices, with two
    components, and on the fall-back to single-vector passes (MI_MULTI_TILE=0)."""
    case = cases(pkg)[name]
    n = case.n_cells
    addr, mat = make(pkg, ctx, case)
    S = orc.System([case])
    srcs = [case.source, 3.0 * (pkg.synthetic.splitmix_uniform(41, n) - 0.5), np.zeros(n)]
    kw = dict(tolerance=1e-9, maxIter=300)

    def single(m, b):
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        perf = m.pbicg(psi, dev(b), precond, **kw)
        return perf, host(psi)

    for tiles in ("1", "0"):
        monkeypatch.setenv("MI_MULTI_TILE", tiles)
        for nrhs in ((3, 2, 1) if tiles == "1" else (3,)):
            psis = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(nrhs)]
            got = mat.pbicg_multi(psis, [dev(b) for b in srcs[:nrhs]], precond, **kw)
            for c in range(nrhs):
                ref, ref_psi = single(mat, srcs[c])
                assert got[c]["nIterations"] == ref["nIterations"] and got[c]["converged"] == ref["converged"]
                assert np.array_equal(got[c]["history"], ref["history"]) and np.array_equal(host(psis[c]), ref_psi)
                if np.any(srcs[c]):     # (bit-equal to the single solve, whose own oracle comparison is test_krylov_histories'; here the north_star bar)
                    o_psi, o = S.pbicg(np.zeros(n), srcs[c], precond, **kw)
                    assert got[c]["nIterations"] == o["nIterations"] and np.max(np.abs(got[c]["history"] - o["history"])) < HIST_RTOL * o["history"][0]
    monkeypatch.setenv("MI_MULTI_TILE", "1")
    assert len({g["nIterations"] for g in got}) > 1 and got[2]["nIterations"] == 0          # the components really ran different loops
    # per-component diagonals
    import copy
    diags = [case.diag * (1.0 + 0.05 * c) + 0.01 * c * pkg.synthetic.splitmix_uniform(50 + c, n) for c in range(3)]
    psis = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
    got = mat.pbicg_multi(psis, [dev(b) for b in srcs], precond, diags=[dev(d) for d in diags], **kw)
    for c in range(3):
        cc = copy.copy(case); cc.diag = diags[c]
        _, mc = make(pkg, ctx, cc)
        ref, ref_psi = single(mc, srcs[c])
        assert got[c]["nIterations"] == ref["nIterations"]
        assert np.array_equal(got[c]["history"], ref["history"]) and np.array_equal(host(psis[c]), ref_psi)
        assert got[c]["normFactor"] == ref["normFactor"]
        if np.any(srcs[c]):
            _, o = orc.System([cc]).pbicg(np.zeros(n), srcs[c], precond, **kw)
            assert got[c]["nIterations"] == o["nIterations"] and np.max(np.abs(got[c]["history"] - o["history"])) < HIST_RTOL * o["history"][0]


@pytest.mark.parametrize("name", ["box_sym", "box_30tiles", "graph_sym", "one_cell", "two_cells"])
@pytest.mark.parametrize("precond", ["diagonal", "none"])
def test_persistent_pcg_kernel_small_matrices(pkg, orc, name, precond, monkeypatch):
    """csrc/persist.inc: the PCG iteration of a small matrix as ONE cooperative kernel per batch -- every CU's workgroup keeps
    its rows' rA, pA, psi and the Amul result in registers (1/diag, diag in LDS) across iterations, the three synchronisation
    points of the reference's loop are grid barriers.  Same iteration counts and histories (1e-10) as the oracle, the maxIter
    quirk and the convergence rule included; sums are grouped per workgroup, so it equals the five-launch pipeline to rounding."""
    monkeypatch.setenv("MI_PCG_PERSIST", "1")
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = pkg.synthetic.box_case(40, 32, 24) if name == "box_30tiles" else cases(pkg)[name]
    addr, mat = make(pkg, ctx, case)
    S = orc.System([case])
    n = case.n_cells
    for kw in (dict(tolerance=1e-9, maxIter=600), dict(tolerance=0.0, maxIter=7), dict(tolerance=1e-30, relTol=1e-3, maxIter=600), dict(tolerance=1e30, minIter=3, maxIter=600)):
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        perf = mat.pcg(psi, dev(case.source), precond, **kw)
        ref_psi, ref = S.pcg(np.zeros(n), case.source, precond, **kw)
        assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == ref["converged"], (kw, perf["nIterations"], ref["nIterations"])
        h, hr = perf["history"], ref["history"]
        assert h.shape == hr.shape and np.max(np.abs(h - hr)) < HIST_RTOL * hr[0]
        assert np.max(np.abs(host(psi) - ref_psi)) < 1e-9 * max(np.max(np.abs(ref_psi)), 1e-300)
    # a solve that starts from the previous solution (residual already small) and the session API mixing both pipelines
    psi2 = psi.clone()
    perf2 = mat.pcg(psi2, dev(case.source), precond, tolerance=1e-9, maxIter=600)
    ref_psi2, ref2 = S.pcg(host(psi), case.source, precond, tolerance=1e-9, maxIter=600)
    assert perf2["nIterations"] == ref2["nIterations"]
    assert ctx.stat(0) > 0                                   # the persistent kernel really ran


@pytest.mark.parametrize("zp", ["0", "1", "5"])
def test_persistent_pcg_kernel_with_and_without_its_first_barrier(pkg, orc, zp, monkeypatch):
    """csrc/persist.inc, ZP: from the second iteration of a batch on, the residual update publishes z = rD o rA and every tile
    forms the pA of its halo cells itself (the owner's fma on the owner's operands), so the barrier between p-update and Amul is
    gone.  MI_PERSIST_ZP = largest number of tiles per workgroup that takes this form (default 1): 0 (never), the default and 5
    (always) must give the SAME BITS on a matrix with one tile per workgroup and on one with two (500 tiles), and the oracle's
    history"""
    monkeypatch.setenv("MI_PCG_PERSIST", "1")
    monkeypatch.setenv("MI_PERSIST_ZP", zp)
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    for dims, precond in (((40, 32, 24), "diagonal"), ((40, 32, 24), "none"), ((80, 80, 78), "diagonal")):
        case = pkg.synthetic.box_case(*dims)
        addr, mat = make(pkg, ctx, case)
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        perf = mat.pcg(psi, dev(case.source), precond, tolerance=1e-8, maxIter=300)
        key = (dims, precond)
        if key not in _ZP_SEEN:
            ref_psi, ref = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, precond, tolerance=1e-8, maxIter=300)
            assert perf["nIterations"] == ref["nIterations"] and np.max(np.abs(perf["history"] - ref["history"])) < HIST_RTOL * ref["history"][0]
            _ZP_SEEN[key] = (perf["history"].copy(), host(psi))
        else:   # every setting: the bits of the first one
            assert np.array_equal(perf["history"], _ZP_SEEN[key][0]) and np.array_equal(host(psi), _ZP_SEEN[key][1])
    assert ctx.stat(0) > 0


_ZP_SEEN = {}


def test_persistent_kernel_on_an_unattached_subdomain_with_processor_patches(pkg, orc, monkeypatch):
    """a sub-domain matrix with processor patches and NO communicator attached takes the single-rank persistent
    kernel; its halo lists index the ext region behind the owned cells (zero in every engine vector).  The ZP form gathers z
    there too: the z buffer used to be n_cells long and uninitialised.  Same history as the five-launch pipeline on the same
    matrix (1e-10), whatever MI_PERSIST_ZP says."""
    eng, syn = pkg.engine, pkg.synthetic
    sub = syn.box_subdomain((40, 32, 24), (1, 2, 2), 1)          # three processor patches, 7 680 cells = 8 tiles
    assert len(sub.interfaces) >= 2
    out = {}
    for persist, zp in (("0", "1"), ("1", "0"), ("1", "1"), ("1", "5")):
        monkeypatch.setenv("MI_PCG_PERSIST", persist); monkeypatch.setenv("MI_PERSIST_ZP", zp)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, sub.n_cells, sub.lower_addr, sub.upper_addr, [i.face_cells for i in sub.interfaces])
        assert addr.n_ext > 0
        mat = eng.Matrix(addr)
        mat.set_coeffs(dev(sub.diag), dev(sub.upper), None)
        for p, itf in enumerate(sub.interfaces):
            mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None)
        psi = torch.zeros(sub.n_cells, dtype=torch.float64, device="cuda:0")
        perf = mat.pcg(psi, dev(sub.source), "diagonal", tolerance=1e-9, maxIter=400)
        assert (ctx.stat(0) > 0) == (persist == "1")
        out[persist, zp] = (perf, host(psi))
    ref, ref_psi = out["0", "1"]
    for key, (perf, psi) in out.items():
        assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == 1, key
        assert np.max(np.abs(perf["history"] - ref["history"])) < HIST_RTOL * ref["history"][0], key
        assert np.max(np.abs(psi - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi)), key
    assert np.array_equal(out["1", "0"][0]["history"], out["1", "5"][0]["history"])     # ZP never changes a bit


def test_persistent_kernel_barrier_litmus_and_recovery_after_a_timeout(pkg, orc, monkeypatch):
    """Hardening of the cooperative path: (i) the grid barrier's litmus (64 generations with payload and
    sum checks on the full cooperative grid) runs once per context before the persistent kernel is used and gates it;
    (ii) a workgroup that never arrives at a barrier (injected: MI_PERSIST_SKIP_ARRIVAL, with a short poll limit) makes the
    solve fail with MI_ERR_DEVICE instead of hanging or returning garbage; (iii) the NEXT solve on the same matrix runs through
    the persistent kernel again and reproduces the oracle -- the barrier words are re-zeroed after a fault (they used to
    stay out of step for every later solve on that matrix)."""
    monkeypatch.setenv("MI_PCG_PERSIST", "1")
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = pkg.synthetic.box_case(40, 32, 24)
    addr, mat = make(pkg, ctx, case)
    S = orc.System([case])
    kw = dict(tolerance=1e-9, maxIter=400)
    ref_psi, ref = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", **kw)
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = mat.pcg(psi, dev(case.source), "diagonal", **kw)
    assert ctx.stat(2) == 1 and ctx.stat(0) > 0                   # the litmus ran (once) and let the kernel through
    assert perf["nIterations"] == ref["nIterations"]
    for skip in ("2", "7"):                                      # a plain barrier of the first iteration / one of the third
        monkeypatch.setenv("MI_PERSIST_SKIP_ARRIVAL", skip); monkeypatch.setenv("MI_PERSIST_BAR_POLLS", "20000")
        psi.zero_()
        with pytest.raises(eng.MiError) as ei:
            mat.pcg(psi, dev(case.source), "diagonal", **kw)
        assert "grid barrier" in str(ei.value)
        monkeypatch.delenv("MI_PERSIST_SKIP_ARRIVAL"); monkeypatch.delenv("MI_PERSIST_BAR_POLLS")
        before = ctx.stat(0)
        psi.zero_()
        perf = mat.pcg(psi, dev(case.source), "diagonal", **kw)
        assert ctx.stat(0) > before and ctx.stat(2) == 1
        assert perf["nIterations"] == ref["nIterations"] and np.max(np.abs(perf["history"] - ref["history"])) < HIST_RTOL * ref["history"][0]
        assert np.max(np.abs(host(psi) - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


def test_fused_distributed_pcg_over_peer_windows_self_exchange(pkg, orc, monkeypatch):
    """Round 3: the THREE-launch distributed PCG iteration (k_dpcg_update_p with the halo pack into the neighbours' windows,
    tile_kernel_dist with boundary tiles polling the flags + the fused wA.pA all-reduce, k_dpcg_update_psi_r with the fused
    two-scalar all-reduce; peer.inc) on a 1-rank communicator whose processor patches point at the rank itself: every store
    goes through the windows, every flag is waited for.  Same iteration counts and history as the serial oracle, and the SAME
    BITS as the phase loop over RCCL (the sums are formed in the same order)."""
    syn, par = pkg.synthetic, pkg.parallel
    monkeypatch.setenv("MI_PCG_PERSIST", "0")                                # (the persistent kernel has its own test below)
    ctx = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
    case = syn.add_cyclic_y(syn.box_case(40, 32, 24, symmetric=True))        # 30 tiles: interior and boundary ones
    S = orc.System([case])
    out = {}
    for mode in ("peer", "rccl"):
        monkeypatch.setenv("MI_ALLREDUCE", mode)
        for precond in ("diagonal", "none"):
            solver = par.DistributedPCG(ctx, case, "cuda:0", precond=precond, n_global=case.n_cells)
            assert solver.driver == "native"
            st = solver.solve(tolerance=1e-9, max_iter=500)
            used, bad = solver.ops.mat.peer_halo_status()
            assert used == (mode == "peer") and bad == 0
            assert solver.comms[0].peer_mode == (mode == "peer")
            ref_psi, ref = S.pcg(np.zeros(case.n_cells), case.source, precond, tolerance=1e-9, maxIter=500)
            _check_hist(st, ref)
            got = solver.ops.solution()
            assert np.max(np.abs(got - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))
            out[mode, precond] = (st["history"], got)
            if mode == "peer":
                assert solver.comms[0].peer_status()[0] == 0
    for precond in ("diagonal", "none"):
        assert np.array_equal(out["peer", precond][0], out["rccl", precond][0])
        assert np.array_equal(out["peer", precond][1], out["rccl", precond][1])


@pytest.mark.parametrize("dims", [(40, 32, 24), (70, 64, 60)])
def test_persistent_distributed_pcg_over_peer_windows_self_exchange(pkg, orc, dims, monkeypatch):
    """csrc/persist.inc, DIST form: the whole distributed PCG iteration -- p-update, halo stores into the neighbours' windows,
    flags, Amul with neighbour-rank values gathered from the window, both all-reduces through the communicator's windows,
    residual update, convergence test -- inside ONE persistent cooperative kernel per batch.  1-rank communicator whose
    processor patches point at the rank itself (every store goes through the windows, every flag is waited for); iteration
    counts and histories (1e-10) of the serial oracle; batches of the five-launch loop in between (Amul timing samples)."""
    syn, par = pkg.synthetic, pkg.parallel
    monkeypatch.setenv("MI_PCG_PERSIST", "1")
    monkeypatch.setenv("MI_ALLREDUCE", "peer")
    ctx = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
    case = syn.add_cyclic_y(syn.box_case(*dims, symmetric=True))
    S = orc.System([case])
    for precond in ("diagonal", "none"):
        solver = par.DistributedPCG(ctx, case, "cuda:0", precond=precond, n_global=case.n_cells)
        assert solver.driver == "native"
        before = ctx.stat(1)
        st = solver.solve(tolerance=1e-9, max_iter=500)
        assert ctx.stat(1) > before                              # the persistent kernel really ran
        assert solver.ops.mat.peer_halo_status() == (True, 0) and solver.comms[0].peer_status()[0] == 0
        ref_psi, ref = S.pcg(np.zeros(case.n_cells), case.source, precond, tolerance=1e-9, maxIter=500)
        _check_hist(st, ref, **PERSIST)
        assert np.max(np.abs(solver.ops.solution() - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))
        # mixed batches: persistent, five-launch (Amul timing), persistent
        solver.begin(tolerance=0.0, max_iter=200)
        solver.iterate(7); solver.iterate(5, time_amul=True, event_stride=2); solver.iterate(9)
        st = solver.end()
        _, ref = S.pcg(np.zeros(case.n_cells), case.source, precond, tolerance=0.0, maxIter=20)
        assert st["nIterations"] == 21 and np.max(np.abs(st["history"][:22] - ref["history"][:22])) < HIST_RTOL * ref["history"][0]
        assert solver.ops.mat.peer_halo_status() == (True, 0)
    # the attached mi_pcg_solve entry point
    eng = pkg.engine
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, [i.face_cells for i in case.interfaces])
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None)
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None)
    comm = eng.Comm(ctx, 1, 0, eng.Comm.unique_id())
    assert comm.peer_auto()
    mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=case.n_cells)
    before = ctx.stat(1)
    psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
    perf = mat.pcg(psi, dev(case.source), "diagonal", tolerance=1e-9, maxIter=500)
    ref_psi, ref = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e-9, maxIter=500)
    _check_hist(perf, ref, **PERSIST)
    assert ctx.stat(1) > before and np.max(np.abs(host(psi) - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))
    mat.detach_comm(); comm.close()


def test_attached_operators_over_halo_windows(pkg, orc, ctx):
    """every attached operator and solver with the halo in peer windows (k_halo_push / k_halo_pull instead of ncclSend/ncclRecv)
    and the scalars in the all-reduce windows, symmetric and asymmetric, incl. GAMG whose level matrices get windows of their
    own; 1-rank communicator, patches to self; bit-exact operators, histories to 1e-10"""
    syn, eng = pkg.synthetic, pkg.engine
    for symmetric in (True, False):
        case = syn.add_cyclic_y(syn.box_case(18, 12, 10, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, [i.face_cells for i in case.interfaces])
        mat = eng.Matrix(addr)
        mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
        for p, itf in enumerate(case.interfaces):
            mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if symmetric else dev(itf.int_coeffs))
        comm = eng.Comm(ctx, 1, 0, eng.Comm.unique_id())
        assert comm.peer_auto()
        mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=case.n_cells)
        assert mat.peer_halo_status() == (True, 0)
        S = orc.System([case])
        n = case.n_cells
        x = syn.splitmix_uniform(3, n) - 0.5
        out = torch.empty(n, dtype=torch.float64, device="cuda:0")
        mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
        mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
        mat.residual(dev(x), dev(case.source), out); assert np.array_equal(host(out), S.residual(x, case.source))
        psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 3)
        assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 3))

        def run(fn_eng, fn_orc, **kw):
            psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            perf = fn_eng(psi, dev(case.source), **kw)
            ref_psi, ref = fn_orc(np.zeros(n), case.source, **kw)
            _check_hist(perf, ref)
            assert np.max(np.abs(host(psi) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))

        if symmetric:
            run(mat.pcg, S.pcg, precond="diagonal", tolerance=1e-9, maxIter=500)      # the fused three-launch iteration
            run(mat.pcg, S.pcg, precond="AINV", tolerance=1e-9, maxIter=500)
        else:
            run(mat.pbicg, S.pbicg, precond="AINV", tolerance=1e-10, maxIter=300)
            run(mat.pbicgstab, S.pbicgstab, precond="diagonal", tolerance=1e-10, maxIter=300)
        run(mat.smooth_solve, S.smooth_solve, n_sweeps=2, tolerance=1e-4, maxIter=400)
        assert mat.peer_halo_status() == (True, 0) and comm.peer_status()[0] == 0
        mat.detach_comm()
        comm.close()


@pytest.mark.parametrize("symmetric", [True, False])
def test_attached_comm_operators_and_solvers(pkg, orc, ctx, symmetric):
    """mi_matrix_attach_comm on a 1-rank RCCL communicator.  The y-periodic box is posed with PROCESSOR patches whose
    neighbour rank is this rank: every operator exchanges its halo through ncclSend/ncclRecv and every solver
    all-reduces its sums, exactly the code an N-rank run executes; the oracle treats the same system as one domain."""
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.add_cyclic_y(syn.box_case(18, 12, 10, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, [i.face_cells for i in case.interfaces])
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if symmetric else dev(itf.int_coeffs))
    uid = eng.Comm.unique_id()
    comm = eng.Comm(ctx, 1, 0, uid)
    mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=case.n_cells)
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(3, n) - 0.5
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
    mat.residual(dev(x), dev(case.source), out); assert np.array_equal(host(out), S.residual(x, case.source))
    psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 3)
    assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 3))
    # patchNeighbourField through the halo exchange: the other side of patch p is patch nbr_patch's faceCells
    nbr = torch.empty(sum(i.face_cells.shape[0] for i in case.interfaces), dtype=torch.float64, device="cuda:0")
    mat.patch_neighbour_field(dev(x), nbr)
    assert np.array_equal(host(nbr), np.concatenate([x[case.interfaces[i.nbr_patch].face_cells] for i in case.interfaces]))

    def run(fn_eng, fn_orc, **kw):
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        perf = fn_eng(psi, dev(case.source), **kw)
        ref_psi, ref = fn_orc(np.zeros(n), case.source, **kw)
        _check_hist(perf, ref)
        assert np.max(np.abs(host(psi) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))

    if symmetric:
        run(mat.pcg, S.pcg, precond="diagonal", tolerance=1e-9, maxIter=500)
        run(mat.pcg, S.pcg, precond="AINV", tolerance=1e-9, maxIter=500)
    else:
        run(mat.pbicg, S.pbicg, precond="AINV", tolerance=1e-10, maxIter=300)
        run(mat.pbicgstab, S.pbicgstab, precond="diagonal", tolerance=1e-10, maxIter=300)
    run(mat.smooth_solve, S.smooth_solve, n_sweeps=2, tolerance=1e-4, maxIter=400)
    mat.detach_comm()
    comm.close()


@pytest.mark.parametrize("mode", ["peer_direct", "peer_pull", "rccl"])
@pytest.mark.parametrize("symmetric", [True, False])
def test_decomposed_solver_paths_self_exchange(pkg, orc, symmetric, mode, monkeypatch):
    """Round 4: the paths a DECOMPOSED case takes through GAMG and the momentum solvers, on a 1-rank
    communicator whose processor patches point at the rank itself (every store, flag and all-reduce issued):
      * every tile operator of an attached matrix in ONE launch, boundary tiles reading the halo window (peer_direct), against
        the round-3 form (k_halo_pull + a second launch: peer_pull) and the send / recv path (rccl): Amul, Tmul, residual and
        Jacobi sweeps bit for bit against the oracle in all three;
      * the attached V-cycle -- processor interfaces on every level, both sums of the scaling factor out of the Amul pass,
        fold + window all-reduce in one launch, fused finest residual -- replayed as a hipGraph over peer windows; cycle by
        cycle against the multi-domain oracle (1e-10), graph on and off;
      * mi_pbicg_solve_multi on an attached matrix: ONE halo exchange for pA and pT of the three components per pass, the
        components' sums in one all-reduce; per component the oracle's history (1e-10) and the single attached solve's."""
    syn, eng = pkg.synthetic, pkg.engine
    monkeypatch.setenv("MI_PCG_PERSIST", "0")
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("win_direct", 0 if mode == "peer_pull" else 1)
    case = syn.add_cyclic_y(syn.box_case(40, 32, 24, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)   # 30 tiles: interior and boundary ones
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, [i.face_cells for i in case.interfaces])
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None if symmetric else dev(itf.int_coeffs))
    comm = eng.Comm(ctx, 1, 0, eng.Comm.unique_id())
    if mode != "rccl":
        assert comm.peer_auto()
    mat.attach_comm(comm, comm, [0, 0], [1, 0], n_global=case.n_cells)
    assert mat.peer_halo_status()[0] == (mode != "rccl")
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(3, n) - 0.5
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    for _ in range(3):          # consecutive exchanges alternate the window parity
        mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
        mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
    mat.residual(dev(x), dev(case.source), out); assert np.array_equal(host(out), S.residual(x, case.source))
    psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 3)
    assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 3))
    # ---- GAMG with processor interfaces on every level
    w = orc.box_face_weights(case)
    G = eng.Gamg(addr, w, 10, comms=(comm, comm), patch_rank=[0, 0], patch_nbr_patch=[1, 0])
    H = orc.GamgSysHierarchy(S, [w], 10)
    for graph in (1, 0):
        ctx.set_option("gamg_graph_attached", graph)
        for kw in (dict(tolerance=1e-9, maxIter=60), dict(tolerance=1e-9, maxIter=60, nPreSweeps=1, nFinestSweeps=3)):
            before = ctx.stat(3)
            psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            perf = G.solve(mat, psi, dev(case.source), **kw)
            ref_psi, ref = H.solve(np.zeros(n), case.source, **kw)
            assert perf["nIterations"] == ref["nIterations"] and perf["nIterations"] > 3
            assert np.max(np.abs(perf["history"] - ref["history"])) < HIST_RTOL * ref["history"][0]
            assert np.max(np.abs(host(psi) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))
            assert (ctx.stat(3) > before) == (graph == 1 and mode != "rccl"), (graph, mode, ctx.stat(3), before)
    # ---- the momentum solvers: single (paired passes, one exchange for pA and pT) and batched
    if not symmetric:
        kw = dict(tolerance=1e-10, maxIter=300)
        srcs = [case.source, 0.5 * case.source + 0.01, np.zeros(n)]
        single = []
        for b in srcs[:2]:
            psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            perf = mat.pbicg(psi, dev(b), "AINV", **kw)
            ref_psi, ref = S.pbicg(np.zeros(n), b, "AINV", **kw)
            _check_hist(perf, ref)
            assert np.max(np.abs(host(psi) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))
            single.append((perf, host(psi)))
        for precond in ("AINV", "diagonal"):
            psis = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
            got = mat.pbicg_multi(psis, [dev(b) for b in srcs], precond, **kw)
            assert got[2]["nIterations"] == 0
            for c in range(2):
                ref_psi, ref = S.pbicg(np.zeros(n), srcs[c], precond, **kw)
                _check_hist(got[c], ref)
                assert np.max(np.abs(host(psis[c]) - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi))
                if precond == "AINV":
                    assert got[c]["nIterations"] == single[c][0]["nIterations"]
                    assert np.allclose(got[c]["history"], single[c][0]["history"], rtol=1e-9, atol=0.0)
        # per-component diagonals on an attached matrix
        diags = [case.diag * (1.0 + 0.05 * c) for c in range(3)]
        psis = [torch.zeros(n, dtype=torch.float64, device="cuda:0") for _ in range(3)]
        got = mat.pbicg_multi(psis, [dev(srcs[0])] * 3, "AINV", diags=[dev(d) for d in diags], **kw)
        import copy
        for c in range(3):
            cc = copy.copy(case); cc.diag = diags[c]
            ref_psi, ref = orc.System([cc]).pbicg(np.zeros(n), srcs[0], "AINV", **kw)
            _check_hist(got[c], ref)
    assert mat.peer_halo_status()[1] == 0
    del G
    mat.detach_comm(); comm.close()


def test_distributed_matrix_single_rank(pkg, orc, ctx):
    """parallel.DistributedMatrix (the per-rank object of a decomposed case) on one rank: comm creation, attach, every solver."""
    syn, par = pkg.synthetic, pkg.parallel
    case = syn.add_cyclic_y(syn.box_case(16, 12, 10, symmetric=True))
    S = orc.System([case])
    dm = par.DistributedMatrix(ctx, case, "cuda:0")
    assert dm.n_global == case.n_cells
    n = case.n_cells
    for solver, kw, ref_fn in [("PCG", dict(precond="diagonal", tolerance=1e-9, maxIter=400), S.pcg),
                               ("smoothSolver", dict(n_sweeps=2, tolerance=1e-4, maxIter=300), S.smooth_solve)]:
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        perf = dm.solve(solver, psi, dev(case.source), **kw)
        ref_psi, ref = ref_fn(np.zeros(n), case.source, **kw)
        _check_hist(perf, ref)
    w = orc.box_face_weights(case)
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = dm.solve("GAMG", psi, dev(case.source), face_weights=w, tolerance=1e-9, maxIter=60)
    ref_psi, ref = orc.GamgSysHierarchy(S, [w], 10).solve(np.zeros(n), case.source, tolerance=1e-9, maxIter=60)
    assert perf["nIterations"] == ref["nIterations"]
    assert np.max(np.abs(perf["history"] - ref["history"])) < 1e-10 * ref["history"][0]


def test_pcg_session_owns_the_context_scratch(pkg, orc, ctx):
    """mi_pcg_begin ... mi_pcg_end owns the context's solver scratch: calls that would overwrite it are refused with
    MI_ERR_STATE instead of corrupting the running solve, the operators stay usable, and the solve is the oracle's."""
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.box_case(15, 12, 9)
    _, mat = make(pkg, ctx, case)
    _, other = make(pkg, ctx, case)
    n = case.n_cells
    b = dev(case.source)
    psi0 = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    mat.pcg_begin(psi0, b, "diagonal", tolerance=1e-9, relTol=0.0, maxIter=300, history_len=302)
    mat.pcg_iterate(5)
    x = dev(syn.splitmix_uniform(3, n)); y = torch.empty_like(x)
    for call in (lambda: ctx.sum(x), lambda: ctx.sum_prod(x, x), lambda: other.pcg(y, b), lambda: other.pbicg(y, b, "diagonal"),
                 lambda: other.pcg_begin(psi0, b, "diagonal", tolerance=1e-9, relTol=0.0, maxIter=10, history_len=12),
                 lambda: mat.smooth_solve(y, b, n_sweeps=1, tolerance=1e-3, maxIter=5)):
        with pytest.raises(eng.MiError, match="PCG session"):
            call()
    other.amul(x, y)                                         # operators do not touch the session's scratch
    assert np.array_equal(host(y), orc.System([case]).amul(host(x)))
    mat.pcg_iterate(400)
    perf = mat.pcg_end(psi0, history_len=302)
    _, ref = orc.System([case]).pcg(np.zeros(n), case.source, "diagonal", tolerance=1e-9, maxIter=300)
    _check_hist(perf, ref, **PERSIST)
    assert abs(ctx.sum(x) - float(np.sum(host(x).astype(np.longdouble)))) < 1e-12 * n   # usable again after mi_pcg_end


@pytest.mark.parametrize("name", ["box_sym", "box_asym", "graph_asym"])
def test_ordered_addressing_runs_on_the_callers_numbering(pkg, orc, ctx, name):
    """mi_addr_create_ordered (the drop-in operators pay no permutation): the mesh is renumbered ONCE with
    the cell order an ordinary layout proposes (what renumberMesh does), the addressing of the renumbered mesh keeps that
    numbering -- identity permutation, same tiles -- and every operator / solver equals the oracle on the renumbered case."""
    eng, syn = pkg.engine, pkg.synthetic
    case = cases(pkg)[name]
    addr0 = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    perm, starts = addr0.cell_perm(), addr0.tile_starts()
    assert not addr0.is_ordered or np.array_equal(perm, np.arange(case.n_cells))
    rc = syn.renumber(case, perm)
    assert np.all(rc.lower_addr < rc.upper_addr) and np.all(np.diff(rc.lower_addr) >= 0)
    addr = eng.Addressing(ctx, rc.n_cells, rc.lower_addr, rc.upper_addr, ordered=True, tile_cell_start=starts)
    assert addr.is_ordered and np.array_equal(addr.cell_perm(), np.arange(rc.n_cells))
    assert addr.n_tiles == addr0.n_tiles and np.array_equal(addr.tile_starts(), starts)
    assert addr.stats()["slots"] == addr0.stats()["slots"]                      # the same tiles, the same cut faces
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(rc.diag), dev(rc.upper), None if rc.lower is None else dev(rc.lower))
    S = orc.System([rc])
    n = rc.n_cells
    x = syn.splitmix_uniform(99, n) - 0.5
    xd, bd = dev(x), dev(rc.source)
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    mat.amul(xd, out); assert np.array_equal(host(out), S.amul(x))
    mat.tmul(xd, out); assert np.array_equal(host(out), S.tmul(x))
    mat.sumA(out); assert np.array_equal(host(out), S.sumA())
    mat.residual(xd, bd, out); assert np.array_equal(host(out), S.residual(x, rc.source))
    mat.H(xd, out); assert np.array_equal(host(out), S.H(x))
    for kind in ("none", "diagonal", "AINV"):
        for tr in (False, True):
            mat.precondition(kind, xd, out, transpose=tr)
            assert np.array_equal(host(out), S.precondition(kind, x, transpose=tr)), (kind, tr)
    for sweeps in (1, 2, 3):
        psi = dev(x.copy()); mat.jacobi_smooth(psi, bd, sweeps, omega=0.9)
        assert np.array_equal(host(psi), S.jacobi_smooth(x, rc.source, sweeps, omega=0.9))
    assert np.array_equal(host(xd), x)                                           # the operators did not write to their input
    # the renumbered product is the original product, renumbered (rows keep their entries; only the order inside a row changes)
    y0 = orc.System([case]).amul(x[np.argsort(perm)])
    assert np.max(np.abs(S.amul(x) - y0[perm])) < 1e-13 * np.max(np.abs(y0))
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    if rc.lower is None:
        perf = mat.pcg(psi, bd, "diagonal", tolerance=1e-9, maxIter=400)
        ref_psi, ref = S.pcg(np.zeros(n), rc.source, "diagonal", tolerance=1e-9, maxIter=400)
    else:
        perf = mat.pbicg(psi, bd, "AINV", tolerance=1e-10, maxIter=300)
        ref_psi, ref = S.pbicg(np.zeros(n), rc.source, "AINV", tolerance=1e-10, maxIter=300)
    _check_hist(perf, ref, **PERSIST)
    # greedy tiles (no tile starts given): still the identity, still exact
    addr_g = eng.Addressing(ctx, rc.n_cells, rc.lower_addr, rc.upper_addr, ordered=True)
    assert addr_g.is_ordered
    mg = eng.Matrix(addr_g); mg.set_coeffs(dev(rc.diag), dev(rc.upper), None if rc.lower is None else dev(rc.lower))
    mg.amul(xd, out); assert np.array_equal(host(out), S.amul(x))
    with pytest.raises(eng.MiError, match="tile_cell_start"):
        eng.Addressing(ctx, rc.n_cells, rc.lower_addr, rc.upper_addr, ordered=True, tile_cell_start=starts[:-1])


def test_ordered_addressing_with_coupled_patches(pkg, orc, ctx):
    # cyclic (local) patches + ordered addressing: operators that read the coupled-patch neighbour values copy the input once
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.add_cyclic_y(syn.box_case(18, 12, 10))
    mk = lambda c, **kw: eng.Addressing(ctx, c.n_cells, c.lower_addr, c.upper_addr, [i.face_cells for i in c.interfaces],
                                        [c.interfaces[i.nbr_patch].face_cells for i in c.interfaces], **kw)
    a0 = mk(case)
    rc = syn.renumber(case, a0.cell_perm())
    addr = mk(rc, ordered=True, tile_cell_start=a0.tile_starts())
    assert addr.is_ordered
    mat = eng.Matrix(addr); mat.set_coeffs(dev(rc.diag), dev(rc.upper), None)
    for p, itf in enumerate(rc.interfaces):
        mat.set_interface_coeffs(p, dev(itf.bou_coeffs), None)
    S = orc.System([rc])
    x = syn.splitmix_uniform(7, rc.n_cells) - 0.5
    out = torch.empty(rc.n_cells, dtype=torch.float64, device="cuda:0")
    mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
    psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(rc.source), 2)
    assert np.array_equal(host(psi), S.jacobi_smooth(x, rc.source, 2))


def test_engine_against_the_reference_source_run_on_the_host(pkg, orc, ctx):
    """The bound between the ENGINE and the reference's own lduMatrixATmul.C (compiled where it lies, run on the host over a
    sequential thrust: tests/golden/golden_ref_atmul.npz, made by tests/golden/make_golden_ref.py), stated where the GPU runs:
      * sumA and H1 -- sums of coefficients in the reference's row order -- are the reference's BITS;
      * Amul / Tmul / residual are within 2 ulp of the row magnitude sum_j |a_ij x_j| of them.  The reference's functor rounds
        its three staged products per side before adding them (matrixMultiplyFunctor<fast,3>), engine and oracle fold every
        term with one fma (what nvcc's contraction makes of the unstaged loop): the same terms in the same order, at most one
        rounding apart per staged term;
      * and the engine IS the oracle's reading, bit for bit (test_spmv_family_bit_exact), so the two statements chain."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_atmul.npz"))
    eps = np.finfo(float).eps
    worst = 0.0
    for name, case in make_golden_ref.atmul_cases(pkg, orc).items():
        addr, mat = make(pkg, ctx, case)
        n = case.n_cells
        x = pkg.synthetic.splitmix_uniform(5, n) - 0.5
        b = pkg.synthetic.splitmix_uniform(6, n) - 0.5
        lower = case.upper if case.lower is None else case.lower
        row_mag = np.abs(case.diag * x)
        np.add.at(row_mag, case.lower_addr, np.abs(case.upper * x[case.upper_addr])); np.add.at(row_mag, case.upper_addr, np.abs(lower * x[case.lower_addr]))
        out = torch.empty(n, dtype=torch.float64, device="cuda:0")
        mat.sumA(out); assert np.array_equal(host(out), G[f"{name}/sumA"]), name
        mat.H1(out); assert np.array_equal(host(out), G[f"{name}/H1/fs0"]), name
        for fs in (0, 1, 2):
            mat.amul(dev(x), out); d = np.max(np.abs(host(out) - G[f"{name}/amul/fs{fs}"]) / row_mag); worst = max(worst, d)
            assert d < 2 * eps, (name, fs, d / eps)
            mat.tmul(dev(x), out); d = np.max(np.abs(host(out) - G[f"{name}/tmul/fs{fs}"]) / row_mag)
            assert d < 2 * eps, (name, fs, d / eps)
        mat.residual(dev(x), dev(b), out)
        assert np.max(np.abs(host(out) - G[f"{name}/residual/fs0"]) / (row_mag + np.abs(b))) < 2 * eps, name
    assert worst > 0.0      # (the two readings do differ somewhere: the bound is not vacuous)


def test_no_device_memory_is_lost_over_create_solve_destroy_cycles(pkg, orc):
    """a solver application creates and destroys matrices, hierarchies and patches for every equation of every time step:
    after a few warm-up cycles (allocator pools, lazily built tables) the free device memory must stop moving"""
    import copy
    eng, syn = pkg.engine, pkg.synthetic
    ctxl = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = syn.add_cyclic_ami_y(syn.box_case(24, 20, 16), shift=0.37)
    base = copy.copy(case); base.interfaces = []
    w = orc.box_face_weights(base)

    def cycle():
        addr = eng.Addressing(ctxl, case.n_cells, case.lower_addr, case.upper_addr, [i.face_cells for i in case.interfaces])
        for p, itf in enumerate(case.interfaces):
            addr.set_ami_patch(p, itf.nbr_patch, itf.ami_start, itf.ami_addr, itf.ami_w, itf.ami_low)
            addr.set_ami_face_areas(p, itf.ami_magsf)
        mat = eng.Matrix(addr)
        mat.set_coeffs(dev(case.diag), dev(case.upper), None)
        for p, itf in enumerate(case.interfaces):
            mat.set_interface_coeffs(p, dev(itf.bou_coeffs), dev(itf.int_coeffs))
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
        mat.pcg(psi, dev(case.source), "AINV", tolerance=1e-6, maxIter=50)
        G = eng.Gamg(addr, w, 10)
        G.solve(mat, psi, dev(case.source), tolerance=1e-6, maxIter=5, directSolveCoarsest=False)
        asm = eng.Assembly(addr)
        d = torch.empty(case.n_cells, dtype=torch.float64, device="cuda:0"); u = torch.empty(case.n_faces, dtype=torch.float64, device="cuda:0")
        asm.fvm_laplacian(dev(np.ones(case.n_faces)), dev(np.ones(case.n_faces)), u, d)
        P = eng.Patch(ctxl, case.n_cells, case.interfaces[0].face_cells)
        P.close(); G.close(); mat.close(); addr.close()
        del psi, d, u
        torch.cuda.synchronize()

    for _ in range(3):
        cycle()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        cycle()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)       # nothing accumulates (a leaked 24x20x16 layout alone is several MB per cycle)


def test_randomly_numbered_mesh_through_every_path(pkg, orc, ctx):
    """a box whose cells were renumbered at random (no locality at all: the clustering takes its Cuthill-McKee path): the
    operators bit for bit, PCG / PBiCG / GAMG histories, and the assembly row passes on the caller's (random) numbering"""
    syn, eng = pkg.synthetic, pkg.engine
    rng = np.random.default_rng(3)
    for symmetric in (True, False):
        base = syn.box_case(24, 20, 16, symmetric=symmetric)
        case = syn.renumber(base, rng.permutation(base.n_cells).astype(np.int32))
        addr, mat = make(pkg, ctx, case)
        ref_addr = eng.Addressing(ctx, base.n_cells, base.lower_addr, base.upper_addr)
        assert addr.n_tiles == ref_addr.n_tiles and addr.stats()["slots"] == ref_addr.stats()["slots"]   # the bricks of the well-numbered box
        S = orc.System([case])
        n = case.n_cells
        x = syn.splitmix_uniform(5, n) - 0.5
        out = torch.empty(n, dtype=torch.float64, device="cuda:0")
        mat.amul(dev(x), out); assert np.array_equal(host(out), S.amul(x))
        mat.tmul(dev(x), out); assert np.array_equal(host(out), S.tmul(x))
        mat.precondition("AINV", dev(x), out); assert np.array_equal(host(out), S.precondition("AINV", x))
        psi = dev(x.copy()); mat.jacobi_smooth(psi, dev(case.source), 2); assert np.array_equal(host(psi), S.jacobi_smooth(x, case.source, 2))
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        if symmetric:
            perf = mat.pcg(psi, dev(case.source), "AINV", tolerance=1e-9, maxIter=300)
            ref_psi, ref = S.pcg(np.zeros(n), case.source, "AINV", tolerance=1e-9, maxIter=300)
        else:
            perf = mat.pbicg(psi, dev(case.source), "DILU", tolerance=1e-8, maxIter=300)
            ref_psi, ref = S.pbicg(np.zeros(n), case.source, "AINV", tolerance=1e-8, maxIter=300)
        _check_hist(perf, ref)
        w = 0.5 + syn.splitmix_uniform(77, case.n_faces)
        H = orc.GamgHierarchy(case, w, 20)
        G = eng.Gamg(addr, w, 20)
        assert G.n_levels == H.n_levels
        ref_psi, ref = H.solve(np.zeros(n), case.source, tolerance=1e-8, maxIter=60)
        psi.zero_()
        perf = G.solve(mat, psi, dev(case.source), tolerance=1e-8, maxIter=60)
        assert perf["nIterations"] == ref["nIterations"]
        assert np.max(np.abs(perf["history"] - ref["history"])) < HIST_RTOL * ref["history"][0]
        asm = eng.Assembly(addr)
        nf = case.n_faces
        delta, gam = 1.0 + syn.splitmix_uniform(7, nf), 0.5 + syn.splitmix_uniform(8, nf)
        uo, do = torch.empty(nf, dtype=torch.float64, device="cuda:0"), torch.empty(n, dtype=torch.float64, device="cuda:0")
        asm.fvm_laplacian(dev(delta), dev(gam), uo, do)
        ru, rd = orc.fvm_laplacian(n, case.lower_addr, case.upper_addr, delta, gam)
        assert np.array_equal(host(uo), ru) and np.array_equal(host(do), rd)
