"""The fvMatrix-assembly operators at BASELINE size (216^3: 10 077 696 cells, 30 093 120 faces), one definition for both sides:

  * oracle_run(): every assembly operator of the C ABI evaluated by the CPU oracle (oracle/fvm_oracle.c, pinned by the
    reference's own functors: tests/test_oracle.py) -- run on the CPU box by tests/golden/make_full_size.py, which stores the
    sha256 of every result's BITS (section `assembly216` of tests/golden/full_size_v1.npz);
  * engine_run(): the same operators through the C ABI on the GPU (tests/test_gpu_full_size.py), compared bit for bit.

Two meshes (VERDICT r05 "next" 1b): `caller` = the box in the caller's lexicographic numbering (the row passes own fixed ranges of
1024 cells), `ordered` = the same box renumbered once into the engine's tile order (mi_addr_create_adopted; the row passes own
the layout's tiles: 9 842 blocks, 16-bit row tables with their escape lists, cut faces recomputed from the schemes' inputs).
Inputs are seeded fields on the mesh at hand (nothing is symmetric, nothing is uniform); coefficient arrays for relax / faceH /
setValues / the row sums are seeded too.  Names follow the reference: lower / upper / diag / source, faceFlux, deltaCoeffs ...
"""
import numpy as np

DIMS = (216, 216, 216)
VARIANTS = ("caller", "ordered")
N_PATCH = 60000          # faces of each of the two boundary patches (cells repeat: several patch faces per cell)
N_SET = 5000             # cells fixed by setValues


def mesh(pkg, variant, dims=DIMS):
    """-> dict(n, nf, lo, up, dims): the box in the caller's numbering, or renumbered into the engine's tile order (host part of
    mi_addr_create_adopted; no device)"""
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.box_case(*dims)
    lo, up = case.lower_addr, case.upper_addr
    if variant == "ordered":
        ad = eng.adopt_host(case.n_cells, lo, up)
        lo, up = ad["lower"], ad["upper"]
    return dict(n=case.n_cells, nf=int(lo.shape[0]), lo=np.ascontiguousarray(lo), up=np.ascontiguousarray(up), dims=dims)


def inputs(pkg, M):
    u = pkg.synthetic.splitmix_uniform
    n, nf = M["n"], M["nf"]
    q = dict(
        Lc=u(101, nf) - 0.5, Uc=u(102, nf) - 0.5, Dc=3.0 + u(103, n), src=u(104, n) - 0.5, psi=u(105, n) - 0.5,
        delta=1.0 + u(107, nf), gamma=0.5 + u(108, nf), w=u(109, nf), flux=u(110, nf) - 0.5, vol=0.5 + u(111, n),
        Sf=[u(120 + k, nf) - 0.5 for k in range(3)], lam=u(130, nf), V=[u(131 + k, n) - 0.5 for k in range(3)],
        rho=0.8 + u(135, n), rho0=0.7 + u(136, n), aA=u(137, nf) - 0.5, aB=u(138, nf), phi0=u(139, nf) - 0.5,
        cdw=0.3 + 0.4 * u(140, nf), g=[u(141 + k, n) - 0.5 for k in range(3)], C=[u(144 + k, n) for k in range(3)],
        cv=[u(150 + k, nf) - 0.5 for k in range(3)], sp=u(153, n), su=[u(154 + k, n) - 0.5 for k in range(3)], su2=u(157, n) - 0.5,
        start=[u(160 + k, n) for k in range(3)], rdt=1.0 / 3e-4,
        fc=[(u(170 + k, N_PATCH) * n).astype(np.int32) for k in range(2)], ic=[u(172 + k, N_PATCH) - 0.5 for k in range(2)],
        bc=[u(174 + k, N_PATCH) - 0.5 for k in range(2)], coupled=[1, 0],
        set_cells=np.unique((u(180, N_SET) * n).astype(np.int32)), ref_cell=int(u(182, 1)[0] * n))
    q["set_vals"] = u(181, q["set_cells"].shape[0]) - 0.5
    return q


def oracle_assemble(orc, M, q, ddt, div, lap, sp=None, su=(), n_rhs=1, want_mag=False):
    """the UNFUSED reference sequence of mi_fvm_assemble with the oracle's sweeps and numpy field operations (one rounding each):
    fvm::ddt -> fvm::div -> operator+ -> fvm::laplacian -> operator- -> [Sp] -> explicit terms (fvMatrix.C:1693-2030,
    lduMatrixOperations.C:235-396)"""
    n, lo, up = M["n"], M["lo"], M["up"]
    diag = None; lower = None; upper = None
    src = [np.zeros(n) for _ in range(n_rhs)]
    if ddt is not None:
        for r in range(n_rhs):
            if ddt.get("rho") is not None:
                diag, src[r] = orc.fvm_ddt_euler_rho(ddt["rdt"], ddt["rho"], ddt["rho_old"], q["vol"], ddt["psi_old"][r])
            else:
                diag, src[r] = orc.fvm_ddt_euler(ddt["rdt"], ddt.get("rho_value", 1.0), q["vol"], ddt["psi_old"][r])
    if div is not None:
        w = div["weights"] if div.get("weights") is not None else orc.upwind_weights(div["flux"])
        lower, upper, dB = orc.fvm_div(n, lo, up, w, div["flux"])
        diag = dB if diag is None else diag + dB
    if lap is not None:
        uL, dL = orc.fvm_laplacian(n, lo, up, lap["delta"], lap["gamma"])
        if upper is None:
            upper = -uL
        else:
            lower, upper = lower - uL, upper - uL
        diag = -dL if diag is None else diag - dL
    if sp is not None:
        t = q["vol"] * sp[0]
        diag = diag + t if sp[1] > 0 else diag - t
    for sign, fields in su:
        for r in range(n_rhs):
            t = q["vol"] * fields[r]
            src[r] = src[r] - t if sign > 0 else src[r] + t
    out = dict(upper=upper, diag=diag)
    if lower is not None:
        out["lower"] = lower
    for r in range(n_rhs):
        out[f"source{r}"] = src[r]
    if want_mag:
        out["sumMag"] = orc.row_face_op(2, n, lo, up, lower, upper, np.zeros(n))
    return out


def _assemble_specs(q):
    """the three assembled systems: momentum-like (everything on), pressure-like (symmetric), convection with given weights"""
    return {
        "assemble_momentum": dict(ddt=dict(rdt=q["rdt"], rho=q["rho"], rho_old=q["rho0"], psi_old=q["V"]), div=dict(flux=q["flux"], weights=None),
                                  lap=dict(delta=q["delta"], gamma=q["gamma"]), sp=(q["sp"], -1.0), su=[(1.0, q["su"]), (-1.0, q["g"])], n_rhs=3, want_mag=True),
        "assemble_pressure": dict(ddt=dict(rdt=q["rdt"], rho=q["rho"], rho_old=q["rho"], psi_old=[q["psi"]]), div=None,
                                  lap=dict(delta=q["delta"], gamma=q["gamma"]), su=[(1.0, [q["su2"]])], n_rhs=1, want_mag=False),
        "assemble_convection": dict(ddt=dict(rdt=q["rdt"], rho_value=1.2, psi_old=[q["psi"]]), div=dict(flux=q["flux"], weights=q["w"]), lap=None,
                                    n_rhs=1, want_mag=True),
    }


def oracle_run(pkg, orc, M, q):
    """name -> float64 array, every operator by the oracle"""
    syn = pkg.synthetic
    n, nf, lo, up = M["n"], M["nf"], M["lo"], M["up"]
    out = {}
    for kind in (0, 1, 2):
        out[f"row_face_op{kind}"] = orc.row_face_op(kind, n, lo, up, q["Lc"], q["Uc"], q["start"][kind])
    out["laplacian/upper"], out["laplacian/diag"] = orc.fvm_laplacian(n, lo, up, q["delta"], q["gamma"])
    out["div/lower"], out["div/upper"], out["div/diag"] = orc.fvm_div(n, lo, up, q["w"], q["flux"])
    out["surfaceIntegrate"] = orc.surface_integrate(n, lo, up, q["flux"], q["vol"])
    out["interpolate"] = orc.face_interpolate(lo, up, q["lam"], q["psi"])
    for k, g in enumerate(orc.gauss_grad(n, lo, up, q["Sf"], q["flux"], q["vol"])):
        out[f"gaussGrad{k}"] = g
    out["limitedLinear/weights"], out["limitedLinear/limiter"] = orc.limited_linear_weights(lo, up, 0.5, q["cdw"], q["flux"], q["psi"], q["g"], q["C"])
    out["ddtCorr/rho"] = orc.ddt_phi_corr(lo, up, q["rdt"], q["lam"], q["Sf"], q["V"], q["rho0"], q["phi0"])
    out["ddtCorr/plain"] = orc.ddt_phi_corr(lo, up, q["rdt"], q["lam"], q["Sf"], q["V"], None, q["phi0"])
    out["fluxDiv/phi"], out["fluxDiv/div"] = orc.flux_div(n, lo, up, q["lam"], q["Sf"], q["V"], scale=q["rho"], add_a=q["aA"], add_b=q["aB"], vol=q["vol"])
    out["snGradCorrection"] = orc.sngrad_correction_flux(lo, up, q["cv"], q["lam"], q["g"], q["gamma"])
    out["relax/diag"], out["relax/source"] = orc.relax(n, lo, up, 0.7, q["Dc"], q["Lc"], q["Uc"], q["src"], q["psi"], q["fc"], q["ic"], q["bc"], q["coupled"])
    case = syn.LduCase(n_cells=n, lower_addr=lo, upper_addr=up, diag=q["Dc"], upper=q["Uc"], lower=q["Lc"], source=q["src"], dims=M["dims"])
    out["faceH"] = orc.System([case]).faceH(q["psi"])
    sv = orc.set_values(n, lo, up, q["set_cells"], q["set_vals"], q["psi"], q["Dc"], q["src"], q["Uc"], q["Lc"], q["fc"], q["ic"], q["bc"])
    for k in ("psi", "source", "upper", "lower"):
        out[f"setValues/{k}"] = sv[k]
    out["setValues/icoeffs0"], out["setValues/bcoeffs1"] = sv["icoeffs"][0], sv["bcoeffs"][1]
    out["setReference/diag"], out["setReference/source"] = orc.set_reference(q["ref_cell"], 0.37, q["Dc"], q["src"])
    for name, spec in _assemble_specs(q).items():
        res = oracle_assemble(orc, M, q, **spec)
        for k, v in res.items():
            out[f"{name}/{k}"] = v
        if name == "assemble_momentum":          # fvMatrix<vector>::relax on the assembled system, sumMagOffDiag taken from the assembly pass
            d = res["diag"]
            srcs = []
            for r in range(3):
                dd, s = orc.relax(n, lo, up, 0.7, d, res["lower"], res["upper"], res[f"source{r}"], q["V"][r], q["fc"], q["ic"], q["bc"], q["coupled"])
                srcs.append(s)
            out["relaxMomentum/diag"] = dd
            for r in range(3):
                out[f"relaxMomentum/source{r}"] = srcs[r]
    return out


def engine_run(pkg, ctx, addr, M, q, device="cuda:0"):
    """name -> float64 host array, every operator through the C ABI on the GPU"""
    import torch
    eng = pkg.engine
    n, nf = M["n"], M["nf"]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    E = lambda m: torch.empty(m, dtype=torch.float64, device=device)
    asm = eng.Assembly(addr)
    out = {}

    def host(t):
        torch.cuda.synchronize()
        return t.cpu().numpy()
    d = {k: dev(v) for k, v in q.items() if isinstance(v, np.ndarray) and v.dtype == np.float64}
    for k in ("Sf", "V", "g", "C", "cv", "su", "start", "ic", "bc"):
        d[k] = [dev(x) for x in q[k]]
    for kind in (0, 1, 2):
        io = d["start"][kind].clone()
        asm.row_face_op(kind, d["Lc"], d["Uc"], io); out[f"row_face_op{kind}"] = host(io)
    uo, do = E(nf), E(n)
    asm.fvm_laplacian(d["delta"], d["gamma"], uo, do); out["laplacian/upper"], out["laplacian/diag"] = host(uo), host(do)
    lo_o = E(nf)
    asm.fvm_div(d["w"], d["flux"], lo_o, uo, do); out["div/lower"], out["div/upper"], out["div/diag"] = host(lo_o), host(uo), host(do)
    asm.surface_integrate(d["flux"], d["vol"], do); out["surfaceIntegrate"] = host(do)
    asm.face_interpolate(d["lam"], d["psi"], uo); out["interpolate"] = host(uo)
    g = [E(n) for _ in range(3)]
    asm.gauss_grad(d["Sf"], d["flux"], d["vol"], g)
    for k in range(3):
        out[f"gaussGrad{k}"] = host(g[k])
    asm.limited_linear_weights(0.5, d["cdw"], d["flux"], d["psi"], d["g"], d["C"], uo, lo_o)
    out["limitedLinear/weights"], out["limitedLinear/limiter"] = host(uo), host(lo_o)
    asm.ddt_phi_corr(q["rdt"], d["lam"], d["Sf"], d["V"], d["rho0"], d["phi0"], uo); out["ddtCorr/rho"] = host(uo)
    asm.ddt_phi_corr(q["rdt"], d["lam"], d["Sf"], d["V"], None, d["phi0"], uo); out["ddtCorr/plain"] = host(uo)
    asm.flux_div(d["lam"], d["Sf"], d["V"], uo, do, cell_scale=d["rho"], add_a=d["aA"], add_b=d["aB"], vol=d["vol"])
    out["fluxDiv/phi"], out["fluxDiv/div"] = host(uo), host(do)
    asm.sngrad_correction_flux(d["cv"], d["lam"], d["g"], d["gamma"], uo); out["snGradCorrection"] = host(uo)
    patches = [eng.Patch(ctx, n, fc) for fc in q["fc"]]
    dg, sr = d["Dc"].clone(), d["src"].clone()
    asm.relax(0.7, dg, d["Lc"], d["Uc"], sr, d["psi"], patches, d["ic"], d["bc"], q["coupled"])
    out["relax/diag"], out["relax/source"] = host(dg), host(sr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(d["Dc"], d["Uc"], d["Lc"])
    mat.faceH(d["psi"], uo); out["faceH"] = host(uo)
    del mat
    ps, sr, ic, bc = d["psi"].clone(), d["src"].clone(), [x.clone() for x in d["ic"]], [x.clone() for x in d["bc"]]
    labels = torch.from_numpy(q["set_cells"]).to(device)
    asm.set_values(labels, dev(q["set_vals"]), ps, d["Dc"], sr, d["Uc"], d["Lc"], uo, lo_o, patches, ic, bc)
    out["setValues/psi"], out["setValues/source"], out["setValues/upper"], out["setValues/lower"] = host(ps), host(sr), host(uo), host(lo_o)
    out["setValues/icoeffs0"], out["setValues/bcoeffs1"] = host(ic[0]), host(bc[1])
    dg, sr = d["Dc"].clone(), d["src"].clone()
    asm.set_reference(q["ref_cell"], 0.37, dg, sr); out["setReference/diag"], out["setReference/source"] = host(dg), host(sr)
    for name, spec in _assemble_specs(q).items():
        nr = spec["n_rhs"]
        srcs = [E(n) for _ in range(nr)]
        mag = E(n) if spec["want_mag"] else None
        dd = spec["ddt"]
        ddt = dict(r_delta_t=dd["rdt"], vol=d["vol"], psi_old=[dev(x) for x in dd["psi_old"]], rho_value=dd.get("rho_value", 1.0))
        if dd.get("rho") is not None:
            ddt["rho"], ddt["rho_old"] = dev(dd["rho"]), dev(dd["rho_old"])
        div = None if spec["div"] is None else dict(flux=d["flux"], weights=None if spec["div"]["weights"] is None else d["w"])
        lap = None if spec["lap"] is None else dict(delta_coeffs=d["delta"], gamma_magsf=d["gamma"])
        sp = None if spec.get("sp") is None else (d["sp"], spec["sp"][1])
        su = [(sign, [dev(f) for f in fields]) for sign, fields in spec.get("su", ())]
        asm.assemble(uo, do, lower_out=lo_o if div is not None else None, sources_out=srcs, ddt=ddt, div=div, laplacian=lap, sp=sp, su=su, sum_mag_out=mag)
        out[f"{name}/upper"], out[f"{name}/diag"] = host(uo), host(do)
        if div is not None:
            out[f"{name}/lower"] = host(lo_o)
        for r in range(nr):
            out[f"{name}/source{r}"] = host(srcs[r])
        if mag is not None:
            out[f"{name}/sumMag"] = host(mag)
        if name == "assemble_momentum":
            asm.relax_multi(0.7, do, None, None, srcs, d["V"], sum_mag=mag, patches=patches, internal_coeffs=d["ic"], boundary_coeffs=d["bc"], coupled=q["coupled"])
            out["relaxMomentum/diag"] = host(do)
            for r in range(3):
                out[f"relaxMomentum/source{r}"] = host(srcs[r])
    return out
