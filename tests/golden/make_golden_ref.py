"""Golden vectors produced by the REFERENCE's own code: pairGAMGAgglomeration::agglomerate compiled from
/root/reference (oracle/_ref/libref_pair.so, oracle/Makefile target `ref`) is run level by level on a few meshes; its
coarse-cell maps are frozen in tests/golden/golden_ref_pair.npz; its PCG::solve / PBiCG::solve / PBiCGStab::solve / smoothSolver::solve (PCG.C,
PBiCG.C, PBiCGStab.C, smoothSolver.C + the functor headers, oracle/_ref/libref_solvers.so) run on the oracle's primitives and their psi and
solverPerformance are frozen in tests/golden/golden_ref_solvers.npz; its GAMGSolver::solve / Vcycle / initVcycle /
solveCoarsestLevel (GAMGSolverSolve.C, oracle/_ref/libref_gamg.so) run on the oracle's hierarchy and primitives ->
tests/golden/golden_ref_gamg.npz; its fvMatrix-assembly functors (oracle/_ref/libref_fvm.so) -> tests/golden/golden_ref_fvm.npz.  Needs the reference tree:
    python tests/golden/make_golden_ref.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402


def cases(pkg, orc):
    from conftest import random_graph_case
    syn = pkg.synthetic
    out = {}
    for name, dims in (("box_14x11x9", (14, 11, 9)), ("box_7x5x3", (7, 5, 3)), ("box_24x1x1", (24, 1, 1))):
        c = syn.box_case(*dims)
        out[name] = (c, orc.box_face_weights(c))
    g = random_graph_case(pkg, 800)
    out["graph_800"] = (g, 0.5 + syn.splitmix_uniform(77, g.n_faces))
    g = random_graph_case(pkg, 300, extra=4.0, seed=9)
    out["graph_300_ties"] = (g, np.ones(g.n_faces))          # all weights equal: pure tie-breaking
    return out


def reference_levels(orc, case, w, n_coarsest, forward):
    """drive the reference's agglomerate() through the levels the way its level loop does (pairGAMGAgglomerate.C:46-120),
    taking the coarse addressing / restricted weights of each level from the oracle's hierarchy"""
    H = orc.GamgHierarchy(case, w, n_coarsest, forward)
    lo, up, ww, n, fwd = case.lower_addr, case.upper_addr, np.asarray(w, dtype=np.float64), case.n_cells, forward
    maps = []
    for l in range(H.n_levels):
        m, nc, fwd = orc.ref_pair_agglomerate(n, lo, up, ww, fwd)
        maps.append(m)
        lv = H.level(l)
        cw = np.zeros(lv["n_coarse_faces"])
        keep = lv["face_restrict"] >= 0
        np.add.at(cw, lv["face_restrict"][keep], ww[keep])
        lo, up, ww, n = lv["lower"], lv["upper"], cw, lv["n_coarse"]
    return maps


def build(pkg, orc):
    out = {}
    for name, (case, w) in cases(pkg, orc).items():
        for forward in (True, False):
            for l, m in enumerate(reference_levels(orc, case, w, 4, forward)):
                out[f"{name}/fwd{int(forward)}/level{l}"] = m
    return out


SOLVER_RUNS = [(True, "pcg", ("none", "diagonal", "AINV")), (False, "pbicg", ("diagonal", "AINV")), (False, "pbicgstab", ("diagonal", "AINV"))]
SOLVER_CONTROLS = [dict(tolerance=0.0, maxIter=7), dict(tolerance=1e-9, maxIter=500), dict(tolerance=1e30, maxIter=50, minIter=3),
                   dict(tolerance=1e-3, relTol=0.1, maxIter=40)]


SMOOTH_CONTROLS = [dict(n_sweeps=2, tolerance=1e-4, maxIter=400), dict(n_sweeps=3, tolerance=0.0, maxIter=10), dict(n_sweeps=-4),
                   dict(n_sweeps=1, tolerance=1e30, maxIter=50, minIter=5)]


def solver_runs(pkg):
    for sym, kind, pres in SOLVER_RUNS:
        case = pkg.synthetic.box_case(9, 8, 7, symmetric=sym)
        for pre in pres:
            for k, kw in enumerate(SOLVER_CONTROLS):
                yield f"{kind}/{pre}/{k}", case, kind, pre, kw
    for sym in (True, False):      # smoothSolver.C (smoother = the reference's Jacobi "GaussSeidel")
        case = pkg.synthetic.box_case(9, 8, 7, symmetric=sym)
        for k, kw in enumerate(SMOOTH_CONTROLS):
            yield f"smooth/{'sym' if sym else 'asym'}/{k}", case, "smooth", None, kw


def oracle_solve(orc, S, case, kind, pre, kw):
    z = np.zeros(case.n_cells)
    if kind == "smooth":
        return S.smooth_solve(z, case.source, **kw)
    return getattr(S, kind)(z, case.source, pre, **kw)


def build_solvers(pkg, orc):
    """psi and solverPerformance of the REFERENCE's PCG::solve / PBiCG::solve / PBiCGStab::solve (compiled from /root/reference,
    oracle/_ref/libref_solvers.so) on the oracle's primitives"""
    out = {}
    for key, case, kind, pre, kw in solver_runs(pkg):
        S = orc.System([case])
        x, p = orc.ref_krylov_solve(kind, S, np.zeros(case.n_cells), case.source, pre or "diagonal", **kw)
        out[key + "/psi"] = x
        out[key + "/perf"] = np.array([p["initialResidual"], p["finalResidual"], p["nIterations"], p["converged"], p["singular"]], dtype=np.float64)
    return out


def gamg_runs(pkg, orc):
    syn = pkg.synthetic
    case = syn.box_case(16, 12, 12)
    asym = syn.box_case(16, 12, 12, symmetric=False)
    runs = [("sym/0", [case], dict(tolerance=1e-9, maxIter=100)), ("sym/1", [case], dict(tolerance=0.0, maxIter=4)),
            ("sym/2", [case], dict(tolerance=1e-9, maxIter=100, nPreSweeps=1)),
            ("sym/3", [case], dict(tolerance=1e-9, maxIter=100, nPreSweeps=2, preSweepsLevelMultiplier=2, scaleCorrection=0)),
            ("sym/4", [case], dict(tolerance=1e30, maxIter=20, minIter=2)),
            ("sym/5", [case], dict(tolerance=1e-9, maxIter=100, nPostSweeps=1, postSweepsLevelMultiplier=2, maxPostSweeps=3, nFinestSweeps=1)),
            ("asym/0", [asym], dict(tolerance=1e-9, maxIter=100)), ("asym/1", [asym], dict(tolerance=1e-9, maxIter=100, scaleCorrection=1, nPreSweeps=1)),
            ("decomposed_2x2x1/0", syn.decompose_box(case, (2, 2, 1)), dict(tolerance=1e-9, maxIter=100)),
            ("decomposed_1x1x3/0", syn.decompose_box(case, (1, 1, 3)), dict(tolerance=1e-9, maxIter=100, nPreSweeps=1)),
            ("cyclic/0", [syn.add_cyclic_y(syn.box_case(12, 12, 10))], dict(tolerance=1e-9, maxIter=100, nPreSweeps=1))]
    for key, subs, kw in runs:
        S = orc.System(subs)
        H = orc.GamgSysHierarchy(S, [orc.box_face_weights(s) for s in subs], 10)
        yield key, S, H, np.concatenate([s.source for s in subs]), kw


def build_gamg(pkg, orc):
    """psi and solverPerformance of the REFERENCE's GAMGSolver::solve (GAMGSolverSolve.C compiled from /root/reference,
    oracle/_ref/libref_gamg.so) on the oracle's hierarchy and primitives"""
    out = {}
    for key, S, H, src, kw in gamg_runs(pkg, orc):
        x, p = orc.ref_gamg_solve(H, np.zeros(S.n), src, **kw)
        out[key + "/psi"] = x
        out[key + "/perf"] = np.array([p["initialResidual"], p["finalResidual"], p["nIterations"], p["converged"], p["singular"]], dtype=np.float64)
    return out


def functor_cases(pkg):
    from conftest import random_graph_case
    syn = pkg.synthetic
    return {"box_sym": syn.box_case(10, 9, 8), "box_asym": syn.box_case(9, 8, 7, symmetric=False),
            "graph_sym": random_graph_case(pkg, 400, extra=4.0), "graph_asym": random_graph_case(pkg, 300, extra=3.0, symmetric=False)}


def build_functors(pkg, orc):
    """row results of the REFERENCE's JacobiSmootherFunctor / AINVPreconditionerFunctor (F.H headers host-compiled,
    oracle/_ref/libref_functors.so)"""
    out = {}
    for name, case in functor_cases(pkg).items():
        x = pkg.synthetic.splitmix_uniform(4, case.n_cells) - 0.5
        out[name + "/jacobi"] = orc.ref_jacobi_rows(case, 0.9, x, case.source)
        out[name + "/ainv"] = orc.ref_ainv_rows(case, x)
        out[name + "/ainvT"] = orc.ref_ainv_rows(case, x, True)
    return out


def gamg_functor_cases(pkg, orc):
    from conftest import random_graph_case
    syn = pkg.synthetic
    g = random_graph_case(pkg, 600, extra=3.0, symmetric=False)
    return {"box_sym": (syn.box_case(10, 9, 8), None), "box_asym": (syn.box_case(9, 8, 7, symmetric=False), None),
            "graph_asym": (g, 0.5 + syn.splitmix_uniform(77, g.n_faces)), "box_asym_merge2": (syn.box_case(8, 8, 6, symmetric=False), None)}


def build_gamg_functors(pkg, orc):
    """coarse matrices of the first levels + a restricted / prolonged field, computed by the REFERENCE's inter-level functors
    (GAMGSolverAgglomerateMatrixF.H, GAMGAgglomerationF.H host-compiled, oracle/_ref/libref_gamg_functors.so) on the
    oracle's restrict addressing"""
    out = {}
    for name, (case, w) in gamg_functor_cases(pkg, orc).items():
        H = orc.GamgHierarchy(case, orc.box_face_weights(case) if w is None else w, 6, merge_levels=2 if name.endswith("merge2") else 1)
        d, u, lo = case.diag, case.upper, case.lower
        for l in range(min(3, H.n_levels)):
            lv = H.level(l)
            x = pkg.synthetic.splitmix_uniform(40 + l, lv["n_fine"]) - 0.5
            out[f"{name}/{l}/restrict"] = orc.ref_gamg_restrict(lv["restrict"], lv["n_coarse"], x)
            out[f"{name}/{l}/prolong"] = orc.ref_gamg_prolong(lv["restrict"], out[f"{name}/{l}/restrict"])
            d, u, lo = orc.ref_gamg_agglomerate_matrix(lv, d, u, lo)
            out[f"{name}/{l}/diag"], out[f"{name}/{l}/upper"] = d, u
            if lo is not None:
                out[f"{name}/{l}/lower"] = lo
    return out


def atmul_cases(pkg, orc):
    from conftest import random_graph_case
    syn = pkg.synthetic
    return {"box_sym": syn.box_case(9, 8, 7), "box_asym": syn.box_case(8, 7, 6, symmetric=False),
            "graph_sym": random_graph_case(pkg, 700), "graph_asym": random_graph_case(pkg, 600, extra=3.0, symmetric=False)}


def build_atmul(pkg, orc):
    """lduMatrix::Amul / Tmul / residual / sumA / H1 computed by the REFERENCE's lduMatrixATmul.C (+ lduAddressingFunctors.H,
    lduMatrixFunctors.H, ops.H) compiled where it lies and run on the host (oracle/_ref/libref_atmul.so); fs = favourSpeed"""
    out = {}
    for name, case in atmul_cases(pkg, orc).items():
        x = pkg.synthetic.splitmix_uniform(5, case.n_cells) - 0.5
        b = pkg.synthetic.splitmix_uniform(6, case.n_cells) - 0.5
        for fs in (0, 1, 2):
            out[f"{name}/amul/fs{fs}"] = orc.ref_atmul(case, "amul", x, favour_speed=fs)
            out[f"{name}/tmul/fs{fs}"] = orc.ref_atmul(case, "tmul", x, favour_speed=fs)
            if fs < 2:      # AINVPreconditioner.C (ctor rD = 1/diag, precondition / preconditionT) of the same translation unit
                out[f"{name}/ainv/fs{fs}"] = orc.ref_atmul(case, "ainv", x, favour_speed=fs)
                out[f"{name}/ainvT/fs{fs}"] = orc.ref_atmul(case, "ainvT", x, favour_speed=fs)
        for fs in (0, 1):
            out[f"{name}/residual/fs{fs}"] = orc.ref_atmul(case, "residual", x, b, favour_speed=fs)
            out[f"{name}/H1/fs{fs}"] = orc.ref_atmul(case, "H1", favour_speed=fs)
        out[f"{name}/sumA"] = orc.ref_atmul(case, "sumA")
    return out


def build_ldu_ops(pkg, orc):
    """sumDiag / negSumDiag / sumMagOffDiag / H computed by the REFERENCE's lduMatrixOperations.C compiled where it lies and run
    on the host (oracle/_ref/libref_ldu_ops.so), on the cases of atmul_cases(); fs = favourSpeed (H only)"""
    out = {}
    for name, case in atmul_cases(pkg, orc).items():
        x = pkg.synthetic.splitmix_uniform(5, case.n_cells) - 0.5
        for which in ("sumDiag", "negSumDiag", "sumMagOffDiag"):
            out[f"{name}/{which}"] = orc.ref_ldu_ops(case, which)
        for fs in (0, 1):
            out[f"{name}/H/fs{fs}"] = orc.ref_ldu_ops(case, "H", x, favour_speed=fs)
    return out


def gamg_scale_cases(pkg, orc):
    syn = pkg.synthetic
    return {"box_sym": [syn.box_case(9, 8, 7)], "box_asym": [syn.box_case(8, 7, 6, symmetric=False)],
            "box_sym_2dom": syn.decompose_box(syn.box_case(10, 8, 6), (2, 1, 1)),
            "box_asym_4dom": syn.decompose_box(syn.box_case(8, 8, 6, symmetric=False), (2, 2, 1))}


def scale_factor(source, field, acf):
    """GAMGSolverScale.C:113-142 as the oracle restates it: both sums accumulated in order in long double, rounded to
    double, sf = num/stabilise(den, VSMALL)"""
    num = float(np.cumsum(np.asarray(source, np.longdouble) * np.asarray(field, np.longdouble))[-1])
    den = float(np.cumsum(np.asarray(acf, np.longdouble) * np.asarray(field, np.longdouble))[-1])
    return num / (den + 1e-300 if den >= 0 else den - 1e-300)


def build_gamg_scale(pkg, orc):
    """GAMGSolver::scale's pointwise update through the REFERENCE's GAMGSolverScaleFunctor (GAMGSolverScale.C compiled where
    it lies, oracle/_ref/libref_gamg_scale.so) on A*field of the oracle and the scaling factor of scale_factor()"""
    out = {}
    for name, subs in gamg_scale_cases(pkg, orc).items():
        S = orc.System(subs)
        field = pkg.synthetic.splitmix_uniform(61, S.n) - 0.5
        source = pkg.synthetic.splitmix_uniform(62, S.n) - 0.5
        _, acf = orc.gamg_sys_scale(S, field, source)
        sf = scale_factor(source, field, acf)
        out[f"{name}/sf"] = np.array([sf])
        out[f"{name}/field"] = orc.ref_gamg_scale_pointwise(sf, field, source, acf, np.concatenate([c.diag for c in subs]))
        out[f"{name}/terms"] = orc.ref_gamg_scale_terms(source, field)
    return out


def fvm_cases(pkg):
    from conftest import random_graph_case
    syn = pkg.synthetic
    return {"box_asym": syn.box_case(10, 9, 8, symmetric=False), "box_sym": syn.box_case(9, 7, 6), "graph_asym": random_graph_case(pkg, 600, extra=3.0, symmetric=False)}


def fvm_inputs(pkg, case):
    """seeded fields for the assembly functors: face fields, cell fields, AoS vectors, one ragged patch with repeated cells"""
    u = pkg.synthetic.splitmix_uniform
    n, nf = case.n_cells, case.n_faces
    npf = 240
    return dict(ssf=u(1, nf) - 0.5, lam=u(2, nf), phi=u(3, n) - 0.5, sf3=u(4, nf * 3).reshape(nf, 3) - 0.5, psi=u(5, n) - 0.5, v3=u(6, n * 3).reshape(n, 3) - 0.5,
                cdw=0.3 + 0.4 * u(7, nf), flux=u(8, nf) - 0.5, g3=u(9, n * 3).reshape(n, 3) - 0.5, C3=u(10, n * 3).reshape(n, 3),
                fc=(u(11, npf) * n).astype(np.int32), pf=u(12, npf) - 0.5, q=u(13, npf) - 0.5, fld=u(14, n) - 0.5, psf3=u(15, npf * 3).reshape(npf, 3) - 0.5,
                sumOff=np.abs(u(16, n)), set_cells=np.unique((u(17, 25) * n).astype(np.int32)), set_vals=u(18, 25) - 0.5)


def build_fvm(pkg, orc):
    """outputs of the REFERENCE's fvMatrix-assembly functors (fvMatrix.C, fvcSurfaceIntegrate.C, gaussGrad.C, surfaceInterpolationScheme.C,
    limitedSurfaceInterpolationScheme.C, LimitedScheme.C + NVDTVD.H / limitedLinear.H, lduMatrixTemplates.C on the reference's own primitives,
    oracle/_ref/libref_fvm.so)"""
    out = {}
    for name, case in fvm_cases(pkg).items():
        q = fvm_inputs(pkg, case)
        n, lo, up = case.n_cells, case.lower_addr, case.upper_addr
        lower = case.upper if case.lower is None else case.lower
        out[f"{name}/surfaceIntegrate"] = orc.ref_surface_integrate_rows(n, lo, up, q["ssf"], True)
        out[f"{name}/surfaceSum"] = orc.ref_surface_integrate_rows(n, lo, up, q["ssf"], False)
        out[f"{name}/interpolate"] = orc.ref_face_interpolate(lo, up, q["lam"], q["phi"])
        iv = orc.ref_face_interpolate(lo, up, q["lam"], q["v3"])
        out[f"{name}/interpolate_vector"] = iv
        out[f"{name}/Sf_dot_interpolate"] = orc.ref_face_dot(q["sf3"], iv)
        out[f"{name}/gaussGrad"] = orc.ref_gauss_grad_rows(n, lo, up, q["sf3"], q["ssf"])
        out[f"{name}/gaussGrad_patch"] = orc.ref_gauss_grad_patch_rows(q["fc"], q["psf3"], q["pf"], out[f"{name}/gaussGrad"])
        out[f"{name}/faceH"] = orc.ref_faceH(lo, up, lower, case.upper, q["psi"])
        for k in (1.0, 0.33):
            out[f"{name}/limitedLinear_{k}/limiter"], out[f"{name}/limitedLinear_{k}/weights"] = orc.ref_limited_linear(lo, up, k, q["cdw"], q["flux"], q["phi"], q["g3"], q["C3"])
        for kind in orc.REF_FVM_PATCH_KINDS:
            out[f"{name}/patch/{kind}"] = orc.ref_fvm_patch_rows(kind, q["fc"], q["pf"], q["fld"], q["q"] if kind == "boundarySource" else None)
        out[f"{name}/relaxDominance"] = orc.ref_fvm_relax_dominance(q["fld"], q["sumOff"])
        mask = np.zeros(n, np.uint8); mask[q["set_cells"]] = 1
        vals = np.zeros(n); vals[q["set_cells"]] = q["set_vals"][:q["set_cells"].shape[0]]
        s, uo, lw = orc.ref_set_values_source(n, lo, up, mask, vals, case.upper, lower, case.source)
        out[f"{name}/setValues/source"], out[f"{name}/setValues/upper"], out[f"{name}/setValues/lower"] = s, uo, lw
        out[f"{name}/relax/diag"], out[f"{name}/relax/source"] = reference_relax(orc, case, q, 0.7)
    return out


def relax_patches(q):
    """two patches for relax: the first coupled (processor-like), the second not"""
    h = q["fc"].shape[0] // 2
    return [q["fc"][:h], q["fc"][h:]], [q["pf"][:h], q["pf"][h:]], [q["q"][:h], q["q"][h:]], [1, 0]


def reference_relax(orc, case, q, alpha):
    """fvMatrix<scalar>::relax(alpha) (fvMatrix.C:1087-1345) composed of the REFERENCE's pieces in its order: sumMagOffDiag of
    lduMatrixOperations.C (libref_ldu_ops.so), the patch functors and the dominance functor of fvMatrix.C (libref_fvm.so); the field
    operations between them (D /= alpha, S += (D - D0)*psi: one rounding each) in numpy"""
    fcs, ics, bcs, coupled = relax_patches(q)
    D = case.diag.copy(); D0 = D.copy()
    sumOff = orc.ref_ldu_ops(case, "sumMagOffDiag")
    for fc, ic, bc, cpl in zip(fcs, ics, bcs, coupled):
        if cpl:
            D = orc.ref_fvm_patch_rows("relaxComponentZero", fc, ic, D)
            sumOff = orc.ref_fvm_patch_rows("relaxMagComponentZero", fc, bc, sumOff)
        else:
            D = orc.ref_fvm_patch_rows("relaxMaxComponentMag", fc, ic, D)
    D = orc.ref_fvm_relax_dominance(D, sumOff)
    D = D / alpha
    for fc, ic, bc, cpl in zip(fcs, ics, bcs, coupled):
        D = orc.ref_fvm_patch_rows("relaxNegComponentZero" if cpl else "relaxNegComponentMin", fc, ic, D)
    t = D - D0
    t = t * q["psi"]
    return D, case.source + t


if __name__ == "__main__":
    graft.build()
    pkg = graft.load_package()
    from oracle import oracle as orc
    assert orc.ref_pair_available(), "oracle/_ref/libref_pair.so missing: needs /root/reference (make -C oracle ref)"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_pair.npz"), **build(pkg, orc))
    assert orc.ref_solvers_available(), "oracle/_ref/libref_solvers.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_solvers.npz"), **build_solvers(pkg, orc))
    assert orc.ref_functors_available(), "oracle/_ref/libref_functors.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_functors.npz"), **build_functors(pkg, orc))
    assert orc.ref_gamg_available(), "oracle/_ref/libref_gamg.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_gamg.npz"), **build_gamg(pkg, orc))
    assert orc.ref_gamg_functors_available(), "oracle/_ref/libref_gamg_functors.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_gamg_functors.npz"), **build_gamg_functors(pkg, orc))
    assert orc.ref_gamg_scale_available(), "oracle/_ref/libref_gamg_scale.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_gamg_scale.npz"), **build_gamg_scale(pkg, orc))
    assert orc.ref_atmul_available(), "oracle/_ref/libref_atmul.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_atmul.npz"), **build_atmul(pkg, orc))
    assert orc.ref_ldu_ops_available(), "oracle/_ref/libref_ldu_ops.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_ldu_ops.npz"), **build_ldu_ops(pkg, orc))
    assert orc.ref_fvm_available(), "oracle/_ref/libref_fvm.so missing"
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ref_fvm.npz"), **build_fvm(pkg, orc))
    print("written")
