"""GPU (-m gpu): the REFERENCE's own solver classes running on top of the engine.

tests/ref_dropin/ compiles the reference's PCG.C, PBiCG.C, PBiCGStab.C and smoothSolver.C from where
they lie (in the build container, where /root/reference exists) with hipcc against a shim whose
device primitives are this repo's C ABI: lduMatrix::Amul/Tmul/residual -> mi_amul/mi_tmul/mi_residual,
preconditioner::precondition[T] -> mi_precondition, smoother::smooth -> mi_jacobi_smooth,
gSumProd/gSumMag -> mi_sum_prod/mi_sum_mag, normFactor -> mi_norm_factor; the reference's axpy functors
run unchanged in a HIP kernel. This is the drop-in boundary of INTEGRATION.md exercised by the
reference's callers themselves. Results must agree with the engine's fused device-resident solvers
and with the oracle (different, deterministic reduction trees: tolerance, not bits).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "ref_dropin", "_ref", "libref_dropin.so")


@pytest.fixture(scope="module")
def dropin(pkg):
    if not os.path.exists(LIB):
        pytest.skip("tests/ref_dropin/_ref/libref_dropin.so not built (needs /root/reference at build time)")
    pkg.engine.lib()  # torch's HIP runtime and the engine first
    lib = C.CDLL(LIB)
    lib.ref_dropin_solve.restype = None
    return lib


@pytest.fixture(scope="module")
def ctx(pkg):
    assert torch.cuda.is_available(), "GPU tests need a device"
    c = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
    yield c
    torch.cuda.synchronize()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def run_ref(lib, pkg, ctx, mat, kind, n, psi0, source, precond, maxIter, tolerance=0.0, n_sweeps=1, omega=0.9):
    psi = dev(psi0.copy())
    src = dev(source)
    out5 = (C.c_double * 5)()
    lib.ref_dropin_solve(C.c_int(kind), ctx.h, mat.h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_int(n),
                         C.c_void_p(psi.data_ptr()), C.c_void_p(src.data_ptr()), C.c_int(pkg.engine.PRECOND[precond]),
                         C.c_double(tolerance), C.c_double(0.0), C.c_int(maxIter), C.c_int(0), C.c_int(n_sweeps), C.c_double(omega), out5)
    torch.cuda.synchronize()
    return psi.cpu().numpy(), dict(initialResidual=out5[0], finalResidual=out5[1], nIterations=int(out5[2]),
                                   converged=bool(out5[3]), singular=bool(out5[4]))


def build(pkg, ctx, case):
    eng = pkg.engine
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None if case.lower is None else dev(case.lower))
    return addr, mat


def check(psi_ref, perf_ref, psi_eng, perf_eng, psi_orc, perf_orc, rtol=1e-9):
    scale = np.max(np.abs(psi_orc)) + 1e-300
    assert perf_ref["nIterations"] == perf_eng["nIterations"] == perf_orc["nIterations"]
    for other in (perf_eng, perf_orc):
        assert abs(perf_ref["initialResidual"] - other["initialResidual"]) <= 1e-12 * abs(other["initialResidual"]) + 1e-300
        assert abs(perf_ref["finalResidual"] - other["finalResidual"]) <= 1e-7 * abs(other["finalResidual"]) + 1e-16
    assert np.max(np.abs(psi_ref - psi_orc)) / scale < rtol
    assert np.max(np.abs(psi_ref - psi_eng)) / scale < rtol


@pytest.mark.parametrize("precond", ["diagonal", "AINV", "none"])
def test_reference_PCG_on_engine(pkg, orc, ctx, dropin, precond):
    case = pkg.synthetic.box_case(21, 17, 13)
    addr, mat = build(pkg, ctx, case)
    n = case.n_cells
    psi0 = np.zeros(n)
    iters = 25
    psi_ref, perf_ref = run_ref(dropin, pkg, ctx, mat, 0, n, psi0, case.source, precond, iters)
    pe = dev(psi0.copy())
    perf_eng = mat.pcg(pe, dev(case.source), precond, tolerance=0.0, maxIter=iters)
    psi_orc, perf_orc = orc.System([case]).pcg(psi0.copy(), case.source, precond, tolerance=0.0, maxIter=iters)
    check(psi_ref, perf_ref, pe.cpu().numpy(), perf_eng, psi_orc, perf_orc)


def test_reference_PCG_converges_to_tolerance(pkg, orc, ctx, dropin):
    case = pkg.synthetic.box_case(24, 20, 16)
    addr, mat = build(pkg, ctx, case)
    n = case.n_cells
    psi_ref, perf_ref = run_ref(dropin, pkg, ctx, mat, 0, n, np.zeros(n), case.source, "diagonal", 500, tolerance=1e-8)
    psi_orc, perf_orc = orc.System([case]).pcg(np.zeros(n), case.source, "diagonal", tolerance=1e-8, maxIter=500)
    assert perf_ref["converged"] and perf_ref["nIterations"] == perf_orc["nIterations"]
    assert np.max(np.abs(psi_ref - psi_orc)) / np.max(np.abs(psi_orc)) < 1e-9


@pytest.mark.parametrize("kind,precond", [(1, "diagonal"), (1, "AINV"), (2, "diagonal"), (2, "AINV")])
def test_reference_PBiCG_PBiCGStab_on_engine(pkg, orc, ctx, dropin, kind, precond):
    case = pkg.synthetic.box_case(21, 17, 13, symmetric=False)
    addr, mat = build(pkg, ctx, case)
    n = case.n_cells
    psi0 = np.zeros(n)
    iters = 12
    psi_ref, perf_ref = run_ref(dropin, pkg, ctx, mat, kind, n, psi0, case.source, precond, iters)
    pe = dev(psi0.copy())
    S = orc.System([case])
    if kind == 1:
        perf_eng = mat.pbicg(pe, dev(case.source), precond, tolerance=0.0, maxIter=iters)
        psi_orc, perf_orc = S.pbicg(psi0.copy(), case.source, precond, tolerance=0.0, maxIter=iters)
    else:
        perf_eng = mat.pbicgstab(pe, dev(case.source), precond, tolerance=0.0, maxIter=iters)
        psi_orc, perf_orc = S.pbicgstab(psi0.copy(), case.source, precond, tolerance=0.0, maxIter=iters)
    check(psi_ref, perf_ref, pe.cpu().numpy(), perf_eng, psi_orc, perf_orc, rtol=1e-8)


@pytest.mark.parametrize("sym", [True, False])
def test_reference_smoothSolver_on_engine(pkg, orc, ctx, dropin, sym):
    case = pkg.synthetic.box_case(21, 17, 13, symmetric=sym)
    addr, mat = build(pkg, ctx, case)
    n = case.n_cells
    psi0 = np.zeros(n)
    psi_ref, perf_ref = run_ref(dropin, pkg, ctx, mat, 3, n, psi0, case.source, "none", 8, n_sweeps=2, omega=0.9)
    pe = dev(psi0.copy())
    perf_eng = mat.smooth_solve(pe, dev(case.source), n_sweeps=2, omega=0.9, tolerance=0.0, maxIter=8)
    psi_orc, perf_orc = orc.System([case]).smooth_solve(psi0.copy(), case.source, n_sweeps=2, omega=0.9, tolerance=0.0, maxIter=8)
    check(psi_ref, perf_ref, pe.cpu().numpy(), perf_eng, psi_orc, perf_orc, rtol=1e-12)


def test_norm_factor_matches_solver_prologue(pkg, orc, ctx):
    case = pkg.synthetic.box_case(21, 17, 13)
    addr, mat = build(pkg, ctx, case)
    n = case.n_cells
    x = pkg.synthetic.splitmix_uniform(5, n) - 0.25
    xd, bd = dev(x), dev(case.source)
    Ax = torch.empty_like(xd)
    mat.amul(xd, Ax)
    nf = mat.norm_factor(xd, bd, Ax)
    perf = mat.pcg(dev(x.copy()), bd, "diagonal", tolerance=0.0, maxIter=1)
    assert nf == perf["normFactor"]                                   # same kernels, same order: same bits
    _, perf_orc = orc.System([case]).pcg(x.copy(), case.source, "diagonal", tolerance=0.0, maxIter=1)
    assert abs(nf - perf_orc["normFactor"]) <= 1e-13 * perf_orc["normFactor"]


@pytest.mark.parametrize("sym,kw", [(True, {}), (True, dict(nPreSweeps=1)), (True, dict(scaleCorrection=0)), (False, {}),
                                    (True, dict(merge_levels=2))])
def test_reference_GAMG_Vcycle_on_engine(pkg, orc, ctx, dropin, sym, kw):
    """The reference's GAMGSolverSolve.C (solve, Vcycle, initVcycle, solveCoarsestLevel) compiled in place for gfx950 runs its
    V-cycle on the engine's level operators: mi_gamg_level_matrix + mi_amul / mi_jacobi_smooth, mi_gamg_restrict / prolong,
    mi_gamg_scale, mi_gamg_solve_coarsest.  Same cycles, residuals and solution as mi_gamg_solve and as the oracle."""
    eng = pkg.engine
    case = pkg.synthetic.box_case(24, 20, 16, symmetric=sym)
    w = orc.box_face_weights(case)
    addr, mat = build(pkg, ctx, case)
    args = dict(tolerance=1e-9, maxIter=60); args.update(kw)
    merge = args.pop("merge_levels", 1)
    G = eng.Gamg(addr, w, 10, merge_levels=merge)
    n = case.n_cells
    ctl = eng.gamg_controls(**args)
    psi = dev(np.zeros(n)); src = dev(case.source)
    out5 = (C.c_double * 5)()
    dropin.ref_dropin_gamg_solve.restype = None
    dropin.ref_dropin_gamg_solve(ctx.h, G.h, mat.h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_int(n), C.c_int(0 if sym else 1),
                                 C.c_void_p(psi.data_ptr()), C.c_void_p(src.data_ptr()), C.byref(ctl), out5)
    torch.cuda.synchronize()
    psi_ref = psi.cpu().numpy()
    pe = dev(np.zeros(n))
    perf_eng = G.solve(mat, pe, src, **args)
    psi_orc, perf_orc = orc.GamgHierarchy(case, w, 10, merge_levels=merge).solve(np.zeros(n), case.source, **args)
    assert int(out5[2]) == perf_eng["nIterations"] == perf_orc["nIterations"]
    assert bool(out5[3]) == bool(perf_eng["converged"]) == bool(perf_orc["converged"])
    for other in (perf_eng, perf_orc):
        assert abs(out5[0] - other["initialResidual"]) <= 1e-12 * abs(other["initialResidual"])
        assert abs(out5[1] - other["finalResidual"]) <= 1e-6 * abs(other["finalResidual"]) + 1e-16
    scale = np.max(np.abs(psi_orc))
    assert np.max(np.abs(psi_ref - psi_orc)) / scale < 1e-9 and np.max(np.abs(psi_ref - pe.cpu().numpy())) / scale < 1e-9


@pytest.mark.parametrize("kind,sym,precond", [(0, True, "diagonal"), (0, True, "AINV"), (1, False, "AINV"), (2, False, "diagonal"), (3, True, "none")])
def test_reference_solvers_on_engine_order_primitives(pkg, orc, ctx, dropin, kind, sym, precond):
    """Same reference solver sources, but their work vectors live in ENGINE order for the duration of the solve
    (mi_vec_to_engine once, mi_*_engine primitives, mi_vec_from_engine once): no permutation pass per operator.  Same answers."""
    case = pkg.synthetic.box_case(21, 17, 13, symmetric=sym)
    addr, mat = build(pkg, ctx, case)
    n = case.n_cells
    iters = 10
    psi_c, perf_c = run_ref(dropin, pkg, ctx, mat, kind, n, np.zeros(n), case.source, precond, iters, n_sweeps=2)
    psi = dev(np.zeros(n)); src = dev(case.source)
    out5 = (C.c_double * 5)()
    dropin.ref_dropin_solve_order.restype = None
    dropin.ref_dropin_solve_order(C.c_int(kind), ctx.h, mat.h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_int(n),
                                  C.c_void_p(psi.data_ptr()), C.c_void_p(src.data_ptr()), C.c_int(pkg.engine.PRECOND[precond]),
                                  C.c_double(0.0), C.c_double(0.0), C.c_int(iters), C.c_int(0), C.c_int(2), C.c_double(0.9), C.c_int(1), out5)
    torch.cuda.synchronize()
    assert int(out5[2]) == perf_c["nIterations"]
    assert abs(out5[0] - perf_c["initialResidual"]) <= 1e-12 * perf_c["initialResidual"]
    assert abs(out5[1] - perf_c["finalResidual"]) <= 1e-7 * perf_c["finalResidual"] + 1e-16
    assert np.max(np.abs(psi.cpu().numpy() - psi_c)) <= 1e-9 * np.max(np.abs(psi_c))
