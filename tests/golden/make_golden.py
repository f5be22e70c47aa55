"""Regenerates tests/golden/golden_v2.npz (and, with --v1, golden_v1.npz) from the CPU oracle.

The reference holds no golden vectors, so these fixtures are the oracle's own
outputs on the synthetic cases, frozen so that any later change of the oracle or of the generators
is caught.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

CASES = {
    "box_sym_12x10x8": dict(dims=(12, 10, 8), symmetric=True),
    "box_asym_12x10x8": dict(dims=(12, 10, 8), symmetric=False),
    "cavity_32": dict(dims=(32, 32, 32), symmetric=True),  # BASELINE config 1 size (N = 32768, F = 95232)
}


def build(pkg, orc):
    out = {}
    for name, spec in CASES.items():
        case = pkg.synthetic.box_case(*spec["dims"], symmetric=spec["symmetric"])
        S = orc.System([case])
        x = pkg.synthetic.splitmix_uniform(2024, case.n_cells) - 0.5
        out[f"{name}/n"] = np.array([case.n_cells, case.n_faces])
        out[f"{name}/amul"] = S.amul(x)[:: max(1, case.n_cells // 257)]
        out[f"{name}/tmul"] = S.tmul(x)[:: max(1, case.n_cells // 257)]
        out[f"{name}/sumA"] = S.sumA()[:: max(1, case.n_cells // 257)]
        z = np.zeros(case.n_cells)
        if spec["symmetric"]:
            for pre in ("diagonal", "AINV", "DIC_upstream"):
                _, p = S.pcg(z, case.source, pre, tolerance=1e-8, maxIter=1000)
                out[f"{name}/pcg_{pre}"] = p["history"]
        else:
            for pre in ("diagonal", "AINV"):
                _, p = S.pbicg(z, case.source, pre, tolerance=1e-10, maxIter=300)
                out[f"{name}/pbicg_{pre}"] = p["history"]
                _, p = S.pbicgstab(z, case.source, pre, tolerance=1e-10, maxIter=300, replicate_quirk=True)
                out[f"{name}/pbicgstab_quirk_{pre}"] = p["history"]
                _, p = S.pbicgstab(z, case.source, pre, tolerance=1e-10, maxIter=300, replicate_quirk=False)
                out[f"{name}/pbicgstab_{pre}"] = p["history"]
    return out


def cyclic_case(pkg, symmetric):
    return pkg.synthetic.add_cyclic_y(pkg.synthetic.box_case(12, 10, 8, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.25)


def front_end_inputs(pkg, case, dims):
    """geometry + fields of the scheme front-end fixtures (uniform box, perturbed so that nothing is symmetric)"""
    syn = pkg.synthetic
    nx = dims[0]
    h = 1.0 / nx
    n, nf = case.n_cells, case.n_faces
    c = np.arange(n)
    C = [(c % nx + 0.5) * h, ((c // nx) % dims[1] + 0.5) * h, (c // (nx * dims[1]) + 0.5) * h]
    d = case.upper_addr.astype(np.int64) - case.lower_addr
    direction = np.where(d == 1, 0, np.where(d == nx, 1, 2))
    Sf = [np.where(direction == k, h * h * (1 + 0.1 * syn.splitmix_uniform(10 + k, nf)), 0.01 * h * h * (syn.splitmix_uniform(20 + k, nf) - 0.5)) for k in range(3)]
    vol = h ** 3 * (1.0 + 0.2 * syn.splitmix_uniform(1, n))
    phi = np.sin(3 * C[0]) * np.cos(2 * C[1]) + C[2] ** 2 + 0.05 * syn.splitmix_uniform(6, n)
    flux = syn.splitmix_uniform(5, nf) - 0.4
    cdw = 0.4 + 0.2 * syn.splitmix_uniform(7, nf)
    return dict(C=C, Sf=Sf, vol=vol, phi=phi, flux=flux, cdw=cdw)


def build_v2(pkg, orc):
    """second fixture set: coupled patches, GAMG (single domain, cyclic, decomposed), scheme front-end"""
    syn = pkg.synthetic
    out = {}
    for sym in (True, False):
        case = cyclic_case(pkg, sym)
        S = orc.System([case])
        tag = "cyclic_sym" if sym else "cyclic_asym"
        x = syn.splitmix_uniform(2024, case.n_cells) - 0.5
        step = max(1, case.n_cells // 257)
        out[f"{tag}/amul"] = S.amul(x)[::step]
        out[f"{tag}/tmul"] = S.tmul(x)[::step]
        out[f"{tag}/jacobi2"] = S.jacobi_smooth(x, case.source, 2)[::step]
        z = np.zeros(case.n_cells)
        if sym:
            out[f"{tag}/pcg_diagonal"] = S.pcg(z, case.source, "diagonal", tolerance=1e-8, maxIter=1000)[1]["history"]
        else:
            out[f"{tag}/pbicg_AINV"] = S.pbicg(z, case.source, "AINV", tolerance=1e-10, maxIter=300)[1]["history"]
        out[f"{tag}/gamg"] = orc.GamgSysHierarchy(S, [orc.box_face_weights(case)], 10).solve(z, case.source, tolerance=1e-9, maxIter=100)[1]["history"]
    case = syn.box_case(16, 12, 12)
    out["box_16x12x12/gamg"] = orc.GamgHierarchy(case, orc.box_face_weights(case), 10).solve(np.zeros(case.n_cells), case.source, tolerance=1e-9, maxIter=100)[1]["history"]
    subs = syn.decompose_box(case, (2, 2, 1))
    S = orc.System(subs)
    H = orc.GamgSysHierarchy(S, [orc.box_face_weights(s) for s in subs], 10)
    out["box_16x12x12_2x2x1/gamg"] = H.solve(np.zeros(S.n), np.concatenate([s.source for s in subs]), tolerance=1e-9, maxIter=100)[1]["history"]
    out["box_16x12x12_2x2x1/level0_coarse_cells"] = np.array([H.level(d, 0)["n_coarse"] for d in range(4)])
    # scheme front-end
    dims = (13, 11, 9)
    case = syn.box_case(*dims)
    f = front_end_inputs(pkg, case, dims)
    ssf = orc.face_interpolate(case.lower_addr, case.upper_addr, f["cdw"], f["phi"])
    g = orc.gauss_grad(case.n_cells, case.lower_addr, case.upper_addr, f["Sf"], ssf, f["vol"])
    stepc, stepf = max(1, case.n_cells // 257), max(1, case.n_faces // 257)
    for k in range(3):
        out[f"front_end/grad{k}"] = g[k][::stepc]
    for kk in (1.0, 0.33):
        w, lim = orc.limited_linear_weights(case.lower_addr, case.upper_addr, kk, f["cdw"], f["flux"], f["phi"], g, f["C"])
        out[f"front_end/limitedLinear_{kk}_w"] = w[::stepf]
        out[f"front_end/limitedLinear_{kk}_limiter"] = lim[::stepf]
    d, s = orc.fvm_ddt_euler(400.0, 1.2, f["vol"], f["phi"])
    out["front_end/ddt_diag"] = d[::stepc]; out["front_end/ddt_source"] = s[::stepc]
    lo, up, dg = orc.fvm_div(case.n_cells, case.lower_addr, case.upper_addr, orc.upwind_weights(f["flux"]), f["flux"])
    out["front_end/div_upwind_lower"] = lo[::stepf]; out["front_end/div_upwind_upper"] = up[::stepf]; out["front_end/div_upwind_diag"] = dg[::stepc]
    return out


def v3_cases(pkg):
    syn = pkg.synthetic
    return {"sym": syn.box_case(16, 12, 12), "asym": syn.box_case(14, 12, 10, symmetric=False)}


def build_v3(pkg, orc):
    """third fixture set: GAMG options (mergeLevels, directSolveCoarsest false), fvMatrix::flux / coupled H pieces, and a ragged
    graph cut by an arbitrary cell-to-processor map"""
    syn = pkg.synthetic
    out = {}
    for tag, case in v3_cases(pkg).items():
        w = orc.box_face_weights(case)
        z = np.zeros(case.n_cells)
        for merge in (2, 3):
            out[f"{tag}/gamg_merge{merge}"] = orc.GamgHierarchy(case, w, 10, merge_levels=merge).solve(z, case.source, tolerance=1e-9, maxIter=100)[1]["history"]
        out[f"{tag}/gamg_iterative_coarsest"] = orc.GamgHierarchy(case, w, 10).solve(z, case.source, tolerance=1e-9, maxIter=100, directSolveCoarsest=False)[1]["history"]
        S = orc.System([case])
        x = syn.splitmix_uniform(2024, case.n_cells) - 0.5
        stepf = max(1, case.n_faces // 257)
        out[f"{tag}/faceH"] = S.faceH(x)[::stepf]
        nx = case.dims[0]
        fc = np.nonzero(np.arange(case.n_cells) % nx == nx - 1)[0].astype(np.int32)
        ic, bc, nbr = syn.splitmix_uniform(31, fc.shape[0]) - 0.5, syn.splitmix_uniform(32, fc.shape[0]) - 0.5, syn.splitmix_uniform(33, fc.shape[0])
        out[f"{tag}/patch_flux_coupled"] = orc.patch_flux(fc, ic, bc, x, nbr)
        out[f"{tag}/patch_flux_plain"] = orc.patch_flux(fc, ic, bc, x, None)
        out[f"{tag}/patch_add_product"] = orc.patch_add_product(fc, bc, nbr, case.source, 0)[fc]
    cyc = cyclic_case(pkg, False)
    x = syn.splitmix_uniform(2024, cyc.n_cells) - 0.5
    step = max(1, cyc.n_cells // 257)
    out["cyclic_asym/H"] = orc.System([cyc]).H(x)[::step]          # face sums only: no interface terms
    out["cyclic_asym/H1"] = orc.System([cyc]).H1()[::step]
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from conftest import random_graph_case
    g = random_graph_case(pkg, 600, extra=2.0, seed=7)
    dom = (syn.splitmix_uniform(91, g.n_cells) * 3).astype(np.int64)
    subs = syn.decompose(g, dom, 3)
    wg = 0.5 + syn.splitmix_uniform(78, g.n_faces)
    S = orc.System(subs)
    H = orc.GamgSysHierarchy(S, [wg[s.global_faces] for s in subs], 6)
    out["graph_3way/gamg"] = H.solve(np.zeros(S.n), np.concatenate([s.source for s in subs]), tolerance=1e-9, maxIter=100)[1]["history"]
    out["graph_3way/level0_coarse_cells"] = np.array([H.level(d, 0)["n_coarse"] for d in range(3)])
    out["graph_3way/pcg_diagonal"] = S.pcg(np.zeros(S.n), np.concatenate([s.source for s in subs]), "diagonal", tolerance=1e-9, maxIter=1000)[1]["history"]
    return out


def v4_cases(pkg):
    syn = pkg.synthetic
    return {"ami_sym": syn.add_cyclic_ami_y(syn.box_case(14, 10, 8), shift=0.37, low_weight_every=9),
            "ami_asym_T": syn.add_cyclic_ami_y(syn.box_case(12, 10, 8, symmetric=False), shift=0.61, transform=0.8)}


def build_v4(pkg, orc):
    """fourth fixture set (round 2): cyclicAMI interfaces -- operators, Jacobi, Krylov histories, the agglomerated AMI of the
    first GAMG level and the GAMG history (iterative coarsest solve) -- on a non-conformal interface with low-weight faces and
    with a transformation factor"""
    syn = pkg.synthetic
    out = {}
    for tag, case in v4_cases(pkg).items():
        S = orc.System([case])
        n = case.n_cells
        x = syn.splitmix_uniform(404, n) - 0.5
        step = max(1, n // 257)
        out[f"{tag}/amul"] = S.amul(x)[::step]; out[f"{tag}/tmul"] = S.tmul(x)[::step]
        out[f"{tag}/residual"] = S.residual(x, case.source)[::step]
        out[f"{tag}/jacobi2"] = S.jacobi_smooth(x, case.source, 2)[::step]
        z = np.zeros(n)
        if case.lower is None:
            out[f"{tag}/pcg_dic"] = S.pcg(z, case.source, "AINV", tolerance=1e-9, maxIter=300)[1]["history"]
        else:
            out[f"{tag}/pbicg_dilu"] = S.pbicg(z, case.source, "AINV", tolerance=1e-8, maxIter=300)[1]["history"]
        import copy
        base = copy.copy(case); base.interfaces = []
        H = orc.GamgSysHierarchy(S, [orc.box_face_weights(base)], 10)
        out[f"{tag}/gamg"] = H.solve(z, case.source, tolerance=1e-9, maxIter=80, directSolveCoarsest=False)[1]["history"]
        for p, itf in enumerate(case.interfaces):
            P = H.patch(0, 0, p, itf.face_cells.shape[0])
            A = H.patch_ami(0, 0, p, P["face_cells"].shape[0])
            out[f"{tag}/level0_patch{p}_start"] = A["start"]; out[f"{tag}/level0_patch{p}_addr"] = A["addr"]; out[f"{tag}/level0_patch{p}_w"] = A["w"]
    return out


def v5_inputs(pkg):
    """fields of the fifth fixture set: a 12 x 10 x 8 box, densities, volumes, face weights / areas, a velocity, old fluxes"""
    syn = pkg.synthetic
    case = syn.box_case(12, 10, 8)
    n, nf = case.n_cells, case.n_faces
    u = syn.splitmix_uniform
    return dict(case=case, rho=0.9 + u(501, n), rho0=0.8 + u(502, n), vol=0.5 + u(503, n), psi0=u(504, n) - 0.5, su=u(505, n) - 0.5, sp=u(506, n),
                susp=u(507, n) - 0.5, vf=u(508, n) - 0.5, lam=u(509, nf), Sf=[u(510 + k, nf) - 0.5 for k in range(3)], U=[u(513 + k, n) - 0.5 for k in range(3)],
                aA=u(516, nf) - 0.5, aB=u(517, nf), phi0=u(518, nf) - 0.5, rdt=1.0 / 3e-4)


def build_v5(pkg, orc):
    """fifth fixture set (round 5): the compressible operators of rhoPimpleFoam -- fvm::ddt(rho, vf) (EulerDdtScheme.C:403-440), fvm::Su / Sp /
    SuSp (fvmSup.C:34-214), fvc::ddtCorr(rho, U, phi) (EulerDdtScheme.C:663-720, ddtScheme.C:139-174), the interpolated flux with its
    surfaceIntegrate (pEqn.H:49-71) -- whole arrays (the case is small)"""
    q = v5_inputs(pkg)
    case = q["case"]
    n, lo, up = case.n_cells, case.lower_addr, case.upper_addr
    out = {}
    d, s = orc.fvm_ddt_euler_rho(q["rdt"], q["rho"], q["rho0"], q["vol"], q["psi0"])
    out["ddt_rho/diag"], out["ddt_rho/source"] = d, s
    out["su/source"] = orc.fvm_su(q["vol"], q["su"], s)
    out["sp/diag"] = orc.fvm_sp(q["vol"], q["sp"], d)
    out["sp_scalar/diag"] = orc.fvm_sp(q["vol"], 0.25, d)
    out["susp/diag"], out["susp/source"] = orc.fvm_susp(q["vol"], q["susp"], q["vf"], d, s)
    out["ddtcorr/rho"] = orc.ddt_phi_corr(lo, up, q["rdt"], q["lam"], q["Sf"], q["U"], q["rho0"], q["phi0"])
    out["ddtcorr/plain"] = orc.ddt_phi_corr(lo, up, q["rdt"], q["lam"], q["Sf"], q["U"], None, q["phi0"])
    out["fluxdiv/phi"], out["fluxdiv/div"] = orc.flux_div(n, lo, up, q["lam"], q["Sf"], q["U"], scale=q["rho"], add_a=q["aA"], add_b=q["aB"], vol=q["vol"])
    out["fluxdiv_plain/phi"], out["fluxdiv_plain/div"] = orc.flux_div(n, lo, up, q["lam"], q["Sf"], q["U"])
    return out


if __name__ == "__main__":
    graft.build()
    pkg = graft.load_package()
    from oracle import oracle as orc
    here = os.path.dirname(os.path.abspath(__file__))
    if "--v1" in sys.argv:   # v1 is frozen; regenerate only on purpose
        np.savez_compressed(os.path.join(here, "golden_v1.npz"), **build(pkg, orc))
    if "--v2" in sys.argv or not os.path.exists(os.path.join(here, "golden_v2.npz")):   # frozen as well
        np.savez_compressed(os.path.join(here, "golden_v2.npz"), **build_v2(pkg, orc))
    if "--v3" in sys.argv or not os.path.exists(os.path.join(here, "golden_v3.npz")):   # frozen
        np.savez_compressed(os.path.join(here, "golden_v3.npz"), **build_v3(pkg, orc))
    if "--v4" in sys.argv or not os.path.exists(os.path.join(here, "golden_v4.npz")):   # frozen
        np.savez_compressed(os.path.join(here, "golden_v4.npz"), **build_v4(pkg, orc))
    np.savez_compressed(os.path.join(here, "golden_v5.npz"), **build_v5(pkg, orc))
    print("written")
