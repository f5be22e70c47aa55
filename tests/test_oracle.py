"""CPU: pin the oracle with algebraic identities (parity is unpinned by the
reference, which has no tests; these are the cross-checks available without its binary)."""
import numpy as np
import pytest

from conftest import random_graph_case


def dense(case):
    n = case.n_cells
    A = np.zeros((n, n), dtype=np.longdouble)
    A[np.arange(n), np.arange(n)] = case.diag
    lo, up = case.lower_addr, case.upper_addr
    A[lo, up] = case.upper
    A[up, lo] = case.upper if case.lower is None else case.lower
    return A


@pytest.mark.parametrize("symmetric", [True, False])
@pytest.mark.parametrize("kind", ["box", "graph"])
def test_spmv_family_against_dense(pkg, orc, symmetric, kind):
    case = pkg.synthetic.box_case(7, 5, 4, symmetric=symmetric) if kind == "box" else random_graph_case(pkg, 300, symmetric=symmetric)
    S = orc.System([case])
    A = dense(case)
    x = pkg.synthetic.splitmix_uniform(11, case.n_cells) - 0.5
    scale = np.abs(A).sum(1).max()
    for got, ref in [(S.amul(x), A @ x), (S.tmul(x), A.T @ x), (S.amul_faceloop(x), A @ x),
                     (S.sumA(), A.sum(1)), (S.residual(x, case.source), case.source - A @ x),
                     (S.H(x), -(A - np.diag(np.diag(A))) @ x), (S.H1(), -(A - np.diag(np.diag(A))).sum(1))]:
        assert np.max(np.abs(got - ref.astype(np.float64))) < 4e-15 * scale
    # symmetric => x'Ay == y'Ax ; Tmul(A) == Amul(A')
    y = pkg.synthetic.splitmix_uniform(12, case.n_cells) - 0.5
    if symmetric:
        assert abs(y @ S.amul(x) - x @ S.amul(y)) < 1e-13 * scale
        assert np.array_equal(S.amul(x), S.tmul(x))
    lo, up = case.lower_addr, case.upper_addr
    fh = S.faceH(x)
    lower = case.upper if case.lower is None else case.lower
    assert np.allclose(fh, case.upper * x[up] - lower * x[lo], rtol=0, atol=1e-15 * scale)


def test_face_loop_order_vs_row_gather_order(pkg, orc):
    # upstream OpenFOAM face loop and RapidCFD row gather agree to a few ulp (SURVEY 8c iii)
    case = pkg.synthetic.box_case(12, 11, 10)
    S = orc.System([case])
    x = pkg.synthetic.splitmix_uniform(5, case.n_cells)
    a, b = S.amul(x), S.amul_faceloop(x)
    assert np.max(np.abs(a - b)) <= 8 * np.finfo(float).eps * np.max(np.abs(a) + 1)


@pytest.mark.parametrize("precond", ["none", "diagonal", "AINV", "DIC_upstream"])
def test_pcg_solves_and_history_is_consistent(pkg, orc, precond):
    case = pkg.synthetic.box_case(10, 9, 8)
    S = orc.System([case])
    A = dense(case).astype(np.float64)
    psi, perf = S.pcg(np.zeros(case.n_cells), case.source, precond, tolerance=1e-9, maxIter=500)
    assert perf["converged"] and not perf["singular"]
    assert perf["history"].shape[0] == perf["nIterations"] + 1
    assert perf["history"][-1] == perf["finalResidual"] < 1e-9
    # the reported residual is the true one: sum|b - A psi| / normFactor
    true = np.abs(case.source - A @ psi).sum() / perf["normFactor"]
    assert abs(true - perf["finalResidual"]) < 1e-6 * perf["finalResidual"] + 1e-14


def test_pcg_iteration_counts_ordering(pkg, orc):
    # 7-point Laplacian sanity: stronger preconditioners need fewer iterations
    case = pkg.synthetic.box_case(16, 16, 16)
    S = orc.System([case])
    it = {p: S.pcg(np.zeros(case.n_cells), case.source, p, tolerance=1e-7, maxIter=2000)[1]["nIterations"]
          for p in ["none", "diagonal", "AINV", "DIC_upstream"]}
    assert it["DIC_upstream"] < it["AINV"] < min(it["diagonal"], it["none"])


@pytest.mark.parametrize("solver", ["pbicg", "pbicgstab"])
@pytest.mark.parametrize("precond", ["diagonal", "AINV"])
def test_asymmetric_solvers(pkg, orc, solver, precond):
    case = pkg.synthetic.box_case(9, 8, 7, symmetric=False)
    S = orc.System([case])
    A = dense(case).astype(np.float64)
    kw = dict(replicate_quirk=False) if solver == "pbicgstab" else {}
    psi, perf = getattr(S, solver)(np.zeros(case.n_cells), case.source, precond, tolerance=1e-10, maxIter=300, **kw)
    assert perf["converged"]
    assert np.abs(A @ psi - case.source).sum() / perf["normFactor"] < 1e-8


def test_max_iter_quirk_and_min_iter(pkg, orc):
    # do { } while (nIterations++ < maxIter ...) runs maxIter+1 bodies (PCG.C:197-204)
    case = pkg.synthetic.box_case(8, 8, 8)
    S = orc.System([case])
    _, perf = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=0.0, maxIter=5)
    assert perf["nIterations"] == 6 and not perf["converged"]
    _, perf = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e30, maxIter=50, minIter=3)
    assert perf["nIterations"] == 3
    _, perf = S.pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e30, maxIter=50)
    assert perf["nIterations"] == 0 and perf["converged"]


def test_singular_detection(pkg, orc):
    case = pkg.synthetic.box_case(4, 4, 4)
    case.source[:] = 0.0  # r = 0 => wApA = 0 => singular break, no iteration counted
    _, perf = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=0.0, maxIter=5)
    assert perf["singular"] and perf["nIterations"] == 0


def test_decomposed_system_matches_serial(pkg, orc):
    # N-subdomain emulation of the one-rank-per-GPU run: same product, same PCG history
    syn = pkg.synthetic
    for symmetric in (True, False):
        case = syn.box_case(10, 8, 6, symmetric=symmetric)
        parts = syn.decompose_box(case, (2, 2, 1))
        S, SD = orc.System([case]), orc.System(parts)
        x = syn.splitmix_uniform(3, case.n_cells) - 0.5
        xg = np.concatenate([x[p.global_cells] for p in parts])
        ref = S.amul(x)
        got = SD.amul(xg)
        assert np.max(np.abs(got - np.concatenate([ref[p.global_cells] for p in parts]))) < 1e-15
        reft = S.tmul(x)
        assert np.max(np.abs(SD.tmul(xg) - np.concatenate([reft[p.global_cells] for p in parts]))) < 1e-15
        assert np.max(np.abs(SD.sumA() - np.concatenate([S.sumA()[p.global_cells] for p in parts]))) < 1e-15
    case = syn.box_case(10, 8, 6)
    parts = syn.decompose_box(case, (2, 1, 2))
    b = np.concatenate([case.source[p.global_cells] for p in parts])
    _, p1 = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=1e-9)
    _, p2 = orc.System(parts).pcg(np.zeros(case.n_cells), b, "diagonal", tolerance=1e-9)
    assert p1["nIterations"] == p2["nIterations"]
    # a 1-ulp change of summation order in Amul is amplified by CG: per-iteration relative
    # agreement degrades as the residual falls, agreement relative to the initial residual holds
    assert np.max(np.abs(p1["history"] - p2["history"])) < 1e-12 * p1["history"][0]
    assert np.max(np.abs(p1["history"] - p2["history"]) / p1["history"]) < 1e-6


def test_jacobi_fixed_point(pkg, orc):
    case = pkg.synthetic.box_case(6, 6, 6, dirichlet_all=True)
    S = orc.System([case])
    A = dense(case).astype(np.float64)
    exact = np.linalg.solve(A, case.source)
    assert np.max(np.abs(S.jacobi_smooth(exact, case.source, 3) - exact)) < 1e-12 * np.max(np.abs(exact))
    x = S.jacobi_smooth(np.zeros(case.n_cells), case.source, 400)
    assert np.max(np.abs(x - exact)) < 1e-3 * np.max(np.abs(exact))


def test_cpu_baseline_kernel_is_the_same_pcg(pkg, orc):
    # bench.py's cpu_baseline leg (one OpenMP thread per domain, face-loop Amul) runs the same diagonal PCG:
    # its sum|rA| after k iterations equals the parity oracle's residual, serial and decomposed
    syn = pkg.synthetic
    case = syn.box_case(20, 16, 12)
    _, p = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", tolerance=0.0, maxIter=29)
    ref = p["finalResidual"] * p["normFactor"]
    for parts in ((1, 1, 1), (1, 1, 3), (1, 1, 4)):
        subs = syn.decompose_box(case, parts) if parts != (1, 1, 1) else [case]
        S = orc.System(subs)
        n, sec, res = S.baseline_pcg(np.concatenate([s.source for s in subs]), 30)
        assert n == 30 and sec > 0
        assert abs(res - ref) < 1e-9 * ref


# ---- independent restatement of the Krylov loops (dense numpy, long double): pins the C oracle's arithmetic ------------
def _np_solver_history(case, solver, precond, n_iter, quirk=True):
    """PCG.C:68-208 / PBiCG.C:67-246 / PBiCGStab.C:67-300 written again from the reference text, on a dense long-double
    matrix: shares no code with oracle/ldu_oracle.c"""
    ld = np.longdouble
    A = dense(case).astype(ld)
    b = case.source.astype(ld)
    D = np.diag(A).copy()
    rD = 1 / D
    LU = A - np.diag(D)

    def M(r, transpose=False):
        if precond == "none":
            return r.copy()
        if precond == "diagonal":
            return rD * r
        return rD * (r - (LU.T if transpose else LU) @ (rD * r))      # AINVPreconditionerF.H:41-99
    psi = np.zeros_like(b)
    Apsi = A @ psi
    sumA = A.sum(axis=1)
    xref = psi.mean()
    nf = (np.abs(Apsi - xref * sumA) + np.abs(b - xref * sumA)).sum() + ld(1e-20)   # lduMatrixSolver.C:182-236
    r = b - Apsi
    hist = [float(np.abs(r).sum() / nf)]
    if solver == "pcg":
        p = np.zeros_like(b); wr_old = None
        for it in range(n_iter):
            w = M(r); wr = w @ r
            p = w if it == 0 else w + (wr / wr_old) * p
            w = A @ p
            alpha = wr / (w @ p)
            psi = psi + alpha * p; r = r - alpha * w; wr_old = wr
            hist.append(float(np.abs(r).sum() / nf))
    elif solver == "pbicg":
        rT = b - A.T @ psi
        p = pT = None; wr_old = None
        for it in range(n_iter):
            w, wT = M(r), M(rT, True); wr = w @ rT
            if it == 0:
                p, pT = w, wT
            else:
                beta = wr / wr_old; p, pT = w + beta * p, wT + beta * pT
            w, wT = A @ p, A.T @ pT
            alpha = wr / (w @ pT)
            psi = psi + alpha * p; r = r - alpha * w; rT = rT - alpha * wT; wr_old = wr
            hist.append(float(np.abs(r).sum() / nf))
    else:
        r0 = r.copy(); p = None; rr_old = alpha = omega = None; Ay = None
        for it in range(n_iter):
            rr = r0 @ r
            p = r.copy() if it == 0 else r + (rr / rr_old) * (alpha / omega) * (p - omega * Ay)
            y = M(p); Ay = A @ y
            alpha = rr / (r0 @ Ay)
            s = r - alpha * Ay
            z = M(s); t = A @ z
            omega = (t @ s) / (t @ t)
            psi = psi + alpha * y + omega * (y if quirk else z)
            r = s - omega * t; rr_old = rr
            hist.append(float(np.abs(r).sum() / nf))
    return np.array(hist), float(nf)


@pytest.mark.parametrize("solver,symmetric,precond", [("pcg", True, "diagonal"), ("pcg", True, "AINV"), ("pcg", True, "none"),
                                                      ("pbicg", False, "diagonal"), ("pbicg", False, "AINV"),
                                                      ("pbicgstab", False, "diagonal"), ("pbicgstab", False, "AINV")])
def test_krylov_histories_against_an_independent_restatement(pkg, orc, solver, symmetric, precond):
    case = pkg.synthetic.box_case(7, 6, 5, symmetric=symmetric)
    n_iter = 12
    S = orc.System([case])
    z = np.zeros(case.n_cells)
    kw = dict(tolerance=0.0, maxIter=n_iter - 1)
    for quirk in ((True, False) if solver == "pbicgstab" else (True,)):
        if solver == "pbicgstab":
            _, perf = S.pbicgstab(z, case.source, precond, replicate_quirk=quirk, **kw)
        else:
            _, perf = getattr(S, solver)(z, case.source, precond, **kw)
        ref, nf = _np_solver_history(case, solver, precond, n_iter, quirk)
        assert abs(perf["normFactor"] - nf) < 1e-14 * nf
        h = perf["history"]
        assert h.shape == ref.shape
        assert np.max(np.abs(h - ref) / ref) < 1e-9, (solver, precond, quirk)


# ---- pinned against the REFERENCE's own solver sources -------------------------------------------------------------
def _ref_solver_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_solvers.npz"))


def test_krylov_loops_equal_the_reference_sources(pkg, orc):
    """tests/golden/golden_ref_solvers.npz was produced by the reference's PCG::solve, PBiCG::solve, PBiCGStab::solve and
    smoothSolver::solve COMPILED FROM /root/reference (PCG.C, PBiCG.C, PBiCGStab.C, smoothSolver.C and their functor headers, against oracle/ref_shim/
    foam_solver_shim.H) running on this oracle's Amul/precondition/gSum primitives.  The oracle's own restatement of those
    loops must give the same bits: psi, residuals, iteration counts, converged/singular -- fixed iteration counts,
    converged runs, the minIter rule (incl. PBiCGStab's mid-iteration exit) and relTol."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = _ref_solver_golden()
    n = 0
    for key, case, kind, pre, kw in make_golden_ref.solver_runs(pkg):
        S = orc.System([case])
        x, p = make_golden_ref.oracle_solve(orc, S, case, kind, pre, kw)
        ref = G[key + "/perf"]
        assert np.array_equal(x, G[key + "/psi"]), key
        assert p["initialResidual"] == ref[0] and p["finalResidual"] == ref[1] and p["nIterations"] == int(ref[2]), key
        assert bool(p["converged"]) == bool(ref[3]) and bool(p["singular"]) == bool(ref[4]), key
        n += 1
    assert n == 36


def test_reference_solver_sources_live_when_built(pkg, orc):
    if not orc.ref_solvers_available():
        pytest.skip("oracle/_ref/libref_solvers.so not built (needs /root/reference)")
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    now = make_golden_ref.build_solvers(pkg, orc)
    G = _ref_solver_golden()
    assert sorted(now) == sorted(G.files)
    for k in G.files:
        assert np.array_equal(now[k], G[k]), k


def test_functor_literal_variant_is_within_rounding(pkg, orc):
    """The reference's Amul functor stages the first 3+3 products in tmpSum[] (lduMatrixATmul.C:42-138); whether nvcc fuses
    those adds is not knowable here.  The oracle's one-fma-per-term row sum and the literal reading agree to a few ulp of
    the row's magnitude on boxes AND on ragged rows (where the literal reading also changes the association), far inside
    the 1e-10 bar set for residual histories."""
    syn = pkg.synthetic
    for case in (syn.box_case(12, 10, 8), syn.box_case(9, 8, 7, symmetric=False), random_graph_case(pkg, 500, extra=4.0)):
        S = orc.System([case])
        x = syn.splitmix_uniform(4, case.n_cells) - 0.5
        a, b = S.amul(x), S.amul_functor_literal(x)
        lower = case.upper if case.lower is None else case.lower
        row_mag = np.abs(case.diag * x)
        np.add.at(row_mag, case.lower_addr, np.abs(case.upper * x[case.upper_addr])); np.add.at(row_mag, case.upper_addr, np.abs(lower * x[case.lower_addr]))
        assert np.max(np.abs(a - b) / row_mag) < 8 * np.finfo(float).eps
        assert np.max(np.abs(a - b)) > 0          # they ARE different roundings: the question is real


def test_row_functors_match_the_reference_headers(pkg, orc):
    """JacobiSmootherF.H and AINVPreconditionerF.H of the reference, compiled as host code (oracle/_ref/libref_functors.so),
    produced tests/golden/golden_ref_functors.npz.  The oracle's Jacobi sweep and AINV apply (plain and transposed) agree to
    a few ulp: the structure of the row arithmetic (sides, signs, transposition, rD placement) is the reference's; which
    mul/add pairs are fused is a compiler's choice (gcc there, explicit fma here, nvcc in the real thing), hence no bitwise bar."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_functors.npz"))
    for name, case in make_golden_ref.functor_cases(pkg).items():
        S = orc.System([case])
        x = pkg.synthetic.splitmix_uniform(4, case.n_cells) - 0.5
        for key, got in (("jacobi", S.jacobi_smooth(x, case.source, 1)), ("ainv", S.precondition("AINV", x)), ("ainvT", S.precondition("AINV", x, transpose=True))):
            ref = G[f"{name}/{key}"]
            assert np.max(np.abs(got - ref)) < 4e-15 * np.max(np.abs(ref)), (name, key)
    if orc.ref_functors_available():
        now = make_golden_ref.build_functors(pkg, orc)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k


def test_spmv_family_against_the_reference_source_run_on_the_host(pkg, orc):
    """lduMatrixATmul.C of the reference (Amul / Tmul / residual / sumA / H1 with callMultiply, matrixMultiplyFunctor<fast,3>,
    matrixOperation / matrixFastOperation of lduAddressingFunctors.H, lduMatrixFunctors.H and ops.H), compiled where it lies
    and RUN on the host over a sequential thrust (oracle/_ref/libref_atmul.so), produced tests/golden/golden_ref_atmul.npz for
    favourSpeed 0, 1, 2 (losort-indirect and pre-sorted paths).
      * sumA and H1 (sums of coefficients in row order) and the oracle's LITERAL reading of the multiply functor
        (orc_amul_functor_literal: staged products rounded separately, extras fused, nExtra last) give the reference's BITS,
        on boxes and ragged graphs, symmetric and asymmetric, on every favourSpeed path;
      * the oracle's default one-fma-per-term Amul / Tmul / residual stay within 2 ulp of the row magnitude of it;
      * the reference's FAST residual and H1 drop the neighbour-side terms beyond the third (nExtra is accumulated and never
        added, lduAddressingFunctors.H:132-139): no effect on hex meshes, and on ragged rows exactly those terms are missing.
        Oracle and engine compute the full row (reference behaviours not replicated)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_atmul.npz"))
    eps = np.finfo(float).eps
    for name, case in make_golden_ref.atmul_cases(pkg, orc).items():
        S = orc.System([case])
        n = case.n_cells
        x = pkg.synthetic.splitmix_uniform(5, n) - 0.5
        b = pkg.synthetic.splitmix_uniform(6, n) - 0.5
        lower = case.upper if case.lower is None else case.lower
        row_mag = np.abs(case.diag * x)
        np.add.at(row_mag, case.lower_addr, np.abs(case.upper * x[case.upper_addr])); np.add.at(row_mag, case.upper_addr, np.abs(lower * x[case.lower_addr]))
        for fs in (0, 1, 2):
            assert np.array_equal(S.amul_functor_literal(x), G[f"{name}/amul/fs{fs}"]), (name, fs)
            assert np.max(np.abs(S.amul(x) - G[f"{name}/amul/fs{fs}"]) / row_mag) < 2 * eps
            assert np.max(np.abs(S.tmul(x) - G[f"{name}/tmul/fs{fs}"]) / row_mag) < 2 * eps
        assert np.array_equal(S.sumA(), G[f"{name}/sumA"]) and np.array_equal(S.H1(), G[f"{name}/H1/fs0"])
        # AINVPreconditioner.C of the reference, run the same way: rD = 1/diag in the constructor, precondition / preconditionT
        # with the triangles swapped for the transpose; the oracle's apply agrees within rounding on both favourSpeed paths
        for fs in (0, 1):
            for key, tr in (("ainv", False), ("ainvT", True)):
                ref = G[f"{name}/{key}/fs{fs}"]
                assert np.max(np.abs(S.precondition("AINV", x, transpose=tr) - ref)) < 4e-15 * np.max(np.abs(ref)), (name, key, fs)
        if case.lower is not None:
            assert np.max(np.abs(G[f"{name}/ainv/fs0"] - G[f"{name}/ainvT/fs0"])) > 1e-6 * np.max(np.abs(G[f"{name}/ainv/fs0"]))
        assert np.max(np.abs(S.residual(x, b) - G[f"{name}/residual/fs0"]) / (row_mag + np.abs(b))) < 2 * eps
        # the fast paths: neighbour-side faces of a row in losort order, those from the fourth on are dropped by the reference
        losort = np.argsort(case.upper_addr, kind="stable")
        rank_in_row = np.arange(losort.shape[0]) - np.searchsorted(case.upper_addr[losort], case.upper_addr[losort])
        dropped = losort[rank_in_row >= 3]
        miss_r, miss_h = np.zeros(n), np.zeros(n)
        np.add.at(miss_r, case.upper_addr[dropped], lower[dropped] * x[case.lower_addr[dropped]])
        np.add.at(miss_h, case.upper_addr[dropped], lower[dropped])
        assert (dropped.shape[0] == 0) == name.startswith("box")
        assert np.max(np.abs(G[f"{name}/residual/fs1"] - (G[f"{name}/residual/fs0"] + miss_r)) / (row_mag + np.abs(b))) < 4 * eps
        assert np.max(np.abs(G[f"{name}/H1/fs1"] - (G[f"{name}/H1/fs0"] + miss_h))) < 4 * eps * np.max(np.abs(G[f"{name}/H1/fs0"]))
    if orc.ref_atmul_available():
        now = make_golden_ref.build_atmul(pkg, orc)
        assert set(now) == set(G.files)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k


def test_matrix_operations_equal_the_reference_source_run_on_the_host(pkg, orc):
    """lduMatrixOperations.C of the reference (sumDiag, negSumDiag, sumMagOffDiag, H; operator=, negate, +=, *= compiled and run
    too), compiled where it lies and run on the host like lduMatrixATmul.C (oracle/_ref/libref_ldu_ops.so), produced
    tests/golden/golden_ref_ldu_ops.npz.  The oracle's row sweeps (orc_row_face_op: which triangle goes with which side of
    the row, signs, magnitudes) and its H operator give the reference's BITS on boxes and ragged graphs, symmetric and
    asymmetric.  The reference's fast H (favourSpeed) stages the first 3+3 products and, like the fast residual, drops the
    neighbour-side terms beyond the third; the full row is what oracle and engine compute."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_ldu_ops.npz"))
    eps = np.finfo(float).eps
    for name, case in make_golden_ref.atmul_cases(pkg, orc).items():
        S = orc.System([case])
        n = case.n_cells
        x = pkg.synthetic.splitmix_uniform(5, n) - 0.5
        sweep = lambda kind, inout: orc.row_face_op(kind, n, case.lower_addr, case.upper_addr, case.lower, case.upper, inout)
        assert np.array_equal(sweep(0, case.diag), G[f"{name}/sumDiag"])
        assert np.array_equal(sweep(1, case.diag), G[f"{name}/negSumDiag"])
        assert np.array_equal(sweep(2, np.zeros(n)), G[f"{name}/sumMagOffDiag"])
        assert np.array_equal(S.H(x), G[f"{name}/H/fs0"])
        lower = case.upper if case.lower is None else case.lower
        row_mag = np.zeros(n)
        np.add.at(row_mag, case.lower_addr, np.abs(case.upper * x[case.upper_addr])); np.add.at(row_mag, case.upper_addr, np.abs(lower * x[case.lower_addr]))
        losort = np.argsort(case.upper_addr, kind="stable")
        rank_in_row = np.arange(losort.shape[0]) - np.searchsorted(case.upper_addr[losort], case.upper_addr[losort])
        dropped = losort[rank_in_row >= 3]
        miss = np.zeros(n)
        np.add.at(miss, case.upper_addr[dropped], lower[dropped] * x[case.lower_addr[dropped]])
        assert np.max(np.abs(G[f"{name}/H/fs1"] - (G[f"{name}/H/fs0"] + miss)) / row_mag) < 4 * eps
    if orc.ref_ldu_ops_available():
        now = make_golden_ref.build_ldu_ops(pkg, orc)
        assert set(now) == set(G.files)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k
        case = make_golden_ref.atmul_cases(pkg, orc)["graph_asym"]          # the coefficient algebra of the same file, run once
        x = pkg.synthetic.splitmix_uniform(5, case.n_cells) - 0.5
        d, u, lo = orc.ref_ldu_ops(case, "scale", x)                         # operator*=(field): row scaling, upper by sf[l], lower by sf[u]
        assert np.array_equal(d, case.diag * x) and np.array_equal(u, case.upper * x[case.lower_addr]) and np.array_equal(lo, case.lower * x[case.upper_addr])
        d, u, lo = orc.ref_ldu_ops(case, "addNegate")                        # B = A; B *= 0.5; A += B; A.negate()
        assert np.array_equal(d, -(1.5 * case.diag)) and np.array_equal(u, -(1.5 * case.upper)) and np.array_equal(lo, -(1.5 * case.lower))


def test_cg_iteration_count_obeys_the_spectral_bound(pkg, orc):
    """sanity of the oracle's PCG against theory.  For the SPD matrix -A of a small box the classical bound
    says the A-norm error falls by 2((sqrt(k)-1)/(sqrt(k)+1))^i; with the exact spectrum from numpy the unpreconditioned
    oracle CG must reach a 1e-8 residual within that many iterations (plus slack for the different norm), converge to the
    dense solution, and need FEWER iterations with the diagonal and AINV preconditioners' better-conditioned operators."""
    case = pkg.synthetic.box_case(9, 7, 5)
    n = case.n_cells
    A = np.zeros((n, n)); A[np.arange(n), np.arange(n)] = case.diag
    A[case.lower_addr, case.upper_addr] = case.upper; A[case.upper_addr, case.lower_addr] = case.upper
    lam = np.linalg.eigvalsh(-A) if np.all(np.linalg.eigvalsh(A) < 0) else np.linalg.eigvalsh(A)
    assert lam.min() > 0
    kappa = lam.max() / lam.min()
    eps = 1e-8
    bound = int(np.ceil(0.5 * np.sqrt(kappa) * np.log(2.0 / eps)))
    S = orc.System([case])
    psi, p = S.pcg(np.zeros(n), case.source, "none", tolerance=eps, maxIter=10 * n)
    assert p["converged"] and p["nIterations"] <= int(1.5 * bound) + 5, (p["nIterations"], bound, kappa)
    ref = np.linalg.solve(A, case.source)
    assert np.max(np.abs(psi - ref)) < 1e-6 * np.max(np.abs(ref))
    _, pd = S.pcg(np.zeros(n), case.source, "diagonal", tolerance=eps, maxIter=10 * n)
    _, pa = S.pcg(np.zeros(n), case.source, "AINV", tolerance=eps, maxIter=10 * n)
    assert pd["converged"] and pa["converged"] and pa["nIterations"] < pd["nIterations"] <= p["nIterations"] + 2


def test_arbitrarily_partitioned_system_matches_serial(pkg, orc):
    """The multi-domain oracle (the judge of every decomposed engine path) on ragged graphs cut by arbitrary cell-to-processor
    maps (synthetic.decompose: contiguous chunks, stripes, random labels): every operator and the Krylov histories equal the
    single-domain ones.  Interface terms are added after the face terms in the decomposed rows, so products agree to rounding,
    not bitwise."""
    from conftest import random_graph_case
    syn = pkg.synthetic
    for seed, n, nd, kind, sym in [(1, 400, 3, "chunks", True), (2, 350, 4, "stripes", False), (3, 500, 5, "random", True), (4, 450, 2, "random", False)]:
        case = random_graph_case(pkg, n, extra=2.0, seed=seed, symmetric=sym)
        c = np.arange(n)
        dom = {"chunks": c * nd // n, "stripes": (c // 5) % nd, "random": (syn.splitmix_uniform(70 + seed, n) * nd).astype(np.int64)}[kind]
        parts = syn.decompose(case, dom, nd)
        assert sum(p.n_cells for p in parts) == n and all(len(p.interfaces) >= 1 for p in parts)
        for d, p in enumerate(parts):                           # both sides of every patch pair see the same faces
            for k, itf in enumerate(p.interfaces):
                other = parts[itf.nbr_domain].interfaces[itf.nbr_patch]
                assert other.nbr_domain == d and other.nbr_patch == k and len(other.face_cells) == len(itf.face_cells)
                assert np.array_equal(itf.bou_coeffs, other.int_coeffs)
        S, SD = orc.System([case]), orc.System(parts)
        pick = lambda v: np.concatenate([v[p.global_cells] for p in parts])
        x = syn.splitmix_uniform(seed, n) - 0.5
        scale = np.max(np.abs(S.amul(x)))
        for name, a, b in (("amul", SD.amul(pick(x)), S.amul(x)), ("tmul", SD.tmul(pick(x)), S.tmul(x)), ("sumA", SD.sumA(), S.sumA()),
                           ("residual", SD.residual(pick(x), pick(case.source)), S.residual(x, case.source))):
            assert np.max(np.abs(a - pick(b))) < 1e-13 * scale, name
        b = pick(case.source)
        if sym:
            _, p1 = S.pcg(np.zeros(n), case.source, "diagonal", tolerance=1e-9); _, p2 = SD.pcg(np.zeros(n), b, "diagonal", tolerance=1e-9)
        else:
            _, p1 = S.pbicgstab(np.zeros(n), case.source, "diagonal", tolerance=1e-9); _, p2 = SD.pbicgstab(np.zeros(n), b, "diagonal", tolerance=1e-9)
        assert abs(p1["nIterations"] - p2["nIterations"]) <= 1
        k = min(len(p1["history"]), len(p2["history"]))
        assert np.max(np.abs(p1["history"][:k] - p2["history"][:k])) < 1e-9 * p1["history"][0]


def test_assembly_oracle_equals_the_reference_functors_run_on_the_host(pkg, orc):
    """The reference's fvMatrix-assembly functors -- fvMatrixPatchAddFunctor, fvMatrixAddBoundarySourceFunctor, the relax functors and
    the setValues functors of fvMatrix.C, surfaceIntegrateFunctor / surfaceIntegratePatchFunctor (fvcSurfaceIntegrate.C),
    surfaceInterpolationSchemeInterpolateFunctor (scalar and vector), gaussGradFunctor / gaussGradPatchFunctor, lduMatrixfaceHFunctor,
    LimitedSchemeCalcLimiterFunctor with limitedLinearLimiter<NVDTVD> and limitedSurfaceInterpolationSchemeWeightsFunctor -- compiled
    from /root/reference on the reference's own Vector / Scalar primitives (oracle/ref_shim/ref_fvm_tu.cpp -> oracle/_ref/libref_fvm.so)
    produced tests/golden/golden_ref_fvm.npz.  oracle/fvm_oracle.c (and faceH of ldu_oracle.c) must give the reference's BITS for every
    one of them, including fvMatrix::relax composed of the reference's own pieces; where the reference tree is present the record is
    re-derived live."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden_ref as mg
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_ref_fvm.npz"))
    cols = lambda a: [np.ascontiguousarray(a[:, k]) for k in range(3)]
    fn_of = {"add": 0, "subtract": 1, "relaxComponentZero": 0, "relaxMagComponentZero": 2, "relaxMaxComponentMag": 2, "relaxNegComponentZero": 1,
             "relaxNegComponentMin": 1, "surfaceIntegratePatch": 0}
    for name, case in mg.fvm_cases(pkg).items():
        q = mg.fvm_inputs(pkg, case)
        n, lo, up = case.n_cells, case.lower_addr, case.upper_addr
        eq = lambda got, key: np.array_equal(got, G[f"{name}/{key}"])
        assert eq(orc.surface_integrate(n, lo, up, q["ssf"]), "surfaceIntegrate")
        assert eq(orc.face_interpolate(lo, up, q["lam"], q["phi"]), "interpolate")
        assert eq(np.stack([orc.face_interpolate(lo, up, q["lam"], v) for v in cols(q["v3"])], 1), "interpolate_vector")
        assert eq(orc.flux_div(n, lo, up, q["lam"], cols(q["sf3"]), cols(q["v3"]), want_div=False), "Sf_dot_interpolate")
        g = orc.gauss_grad(n, lo, up, cols(q["sf3"]), q["ssf"])
        assert eq(np.stack(g, 1), "gaussGrad")
        gp = [orc.patch_add_product(q["fc"], psf, q["pf"], gk, 0) for psf, gk in zip(cols(q["psf3"]), g)]   # out += Sf[face]*issf[face]: one fma
        assert eq(np.stack(gp, 1), "gaussGrad_patch")
        assert eq(orc.System([case]).faceH(q["psi"]), "faceH")
        for k in (1.0, 0.33):
            w, lim = orc.limited_linear_weights(lo, up, k, q["cdw"], q["flux"], q["phi"], cols(q["g3"]), cols(q["C3"]))
            assert eq(lim, f"limitedLinear_{k}/limiter") and eq(w, f"limitedLinear_{k}/weights")
        for kind, fn in fn_of.items():
            assert eq(orc.patch_add(q["fc"], q["pf"], q["fld"], fn), f"patch/{kind}"), kind
        assert eq(orc.patch_add_product(q["fc"], q["pf"], q["q"], q["fld"], 0), "patch/boundarySource")
        assert eq(np.maximum(np.abs(q["fld"]), q["sumOff"]), "relaxDominance")
        fcs, ics, bcs, coupled = mg.relax_patches(q)
        d, s = orc.relax(n, lo, up, 0.7, case.diag, case.lower, case.upper, case.source, q["psi"], fcs, ics, bcs, coupled)
        assert eq(d, "relax/diag") and eq(s, "relax/source")
        sv = orc.set_values(n, lo, up, q["set_cells"], q["set_vals"][:q["set_cells"].shape[0]], q["psi"], case.diag, case.source, case.upper, case.lower)
        keep = np.ones(n, bool); keep[q["set_cells"]] = False
        assert np.array_equal(sv["source"][keep], G[f"{name}/setValues/source"][keep])
        assert np.array_equal(sv["source"][~keep], q["set_vals"][:q["set_cells"].shape[0]] * case.diag[q["set_cells"]])
        assert eq(sv["upper"], "setValues/upper") and eq(sv["lower"], "setValues/lower")
    if orc.ref_fvm_available():
        now = mg.build_fvm(pkg, orc)
        assert set(now) == set(G.files)
        for k in G.files:
            assert np.array_equal(now[k], G[k]), k
