"""The BASELINE workloads beyond config 2 as callable measurements (bench.py appends them to its JSON line as
`config.supplements`; tools/bench_timestep.py and tools/bench_gamg.py print them on their own):

  gamg_supplement      config 3: GAMG pressure solve on the 10 M-cell box (V-cycles/s, fraction of the HBM roofline on the
                       ALGORITHMIC bytes of the reference's unfused cycle, cycles to 1e-6)
  timestep_supplement  configs 4 / 5 per rank: one PISO-like time step -- full fvMatrix assembly, PBiCG + DILU momentum
                       (three components), GAMG pressure, flux / gradient correction -- every stage through the C ABI
"""
import os
import time

import numpy as np


def box_direction(case):
    nx = case.dims[0]
    d = case.upper_addr.astype(np.int64) - case.lower_addr
    return np.where(d == 1, 0, np.where(d == nx, 1, 2))


def box_pair_weights(case):
    return (1.0 / case.dims[0]) * np.array([1.0, 1.01, 1.02])[box_direction(case)]   # faceAreaPair weights of the box


def gamg_cycle_bytes(levels, n0, f0, ctl):
    """ALGORITHMIC bytes of one V-cycle + finest residual, summed over the levels with the SURVEY.md 8(d) formulas and the
    reference's UNFUSED op sequence (GAMGSolverSolve.C:181-474), symmetric matrix: Jacobi sweep 32N+16F; Amul 24N+16F;
    restrict or prolong between levels (8+4)N_fine + 8N_coarse; correction scaling = Amul + two dot products (2 x 16N) + the
    scaling pass (field, Acf, source, D -> field: 40N); finest: psi += corr (24N), residual = Amul + subtract (24N) + sumMag (8N).
    levels: [(cells, faces)] of the coarse levels 0..L-1; n0, f0 the finest level."""
    nL = len(levels)
    size = [(n0, f0)] + list(levels)                       # size[k]: finest is k = 0, coarse level l is k = l + 1
    tot = 0.0
    for k in range(nL):                                    # restrict k -> k+1 on the way down, prolong on the way up
        tot += 2 * ((8 + 4) * size[k][0] + 8 * size[k + 1][0])
    for l in range(nL - 1):                                # every coarse level but the coarsest: [scale] + post sweeps
        n, f = levels[l]
        sweeps = min(ctl["nPostSweeps"] + ctl["postSweepsLevelMultiplier"] * l, ctl["maxPostSweeps"])
        tot += sweeps * (32 * n + 16 * f)
        if l < nL - 2:
            tot += (24 * n + 16 * f) + 32 * n + 40 * n
    tot += (24 * n0 + 16 * f0) + 32 * n0 + 40 * n0 + 24 * n0                       # finest: scale + psi update
    tot += ctl["nFinestSweeps"] * (32 * n0 + 16 * f0)
    tot += (24 * n0 + 16 * f0) + 24 * n0 + 8 * n0                                   # finest residual
    nc = levels[-1][0]
    tot += 8 * nc * nc + 16 * nc                                                    # coarsest: dense inverse times source
    return tot


def gamg_supplement(eng, case, addr, mat, dev, cycles=20, repeats=3, hbm_peak_gbs=8000.0):
    """BASELINE config 3 on an existing addressing / matrix of the box: V-cycles/s inside mi_gamg_solve (tolerance 0)"""
    import torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    N, F = case.n_cells, case.n_faces
    w = box_pair_weights(case)           # the caller's input (GAMGAgglomeration's face weights), not part of the engine's start-up
    t0 = time.perf_counter()
    G = eng.Gamg(addr, w, 100)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    src = t(case.source)
    psi = torch.zeros(N, dtype=torch.float64, device=dev)
    G.solve(mat, psi, src, tolerance=0.0, maxIter=3)           # warm-up: level matrices, scratch, the cycle graph
    rep = []
    for _ in range(repeats):
        psi.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        perf = G.solve(mat, psi, src, tolerance=0.0, maxIter=cycles)
        torch.cuda.synchronize()
        rep.append(time.perf_counter() - t0)
        assert perf["nIterations"] == cycles, perf
    el = float(np.median(rep))
    levels = [(G.level_sizes(l)["n_coarse"], G.level_sizes(l)["n_coarse_faces"]) for l in range(G.n_levels)]
    ctl = dict(nPostSweeps=2, postSweepsLevelMultiplier=1, maxPostSweeps=4, nFinestSweeps=2)
    alg = gamg_cycle_bytes(levels, N, F, ctl)
    psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    pc = G.solve(mat, psi, src, tolerance=1e-6, maxIter=200); torch.cuda.synchronize(); t_conv = time.perf_counter() - t0
    out = {"workload": "BASELINE config 3: GAMG pressure solve on the same box (faceAreaPair agglomeration, "
                       f"{G.n_levels} levels down to {levels[-1][0]} cells, GaussSeidel(=Jacobi) smoother, correction scaling, direct coarsest solve)",
           "v_cycles_per_s": cycles / el, "ms_per_v_cycle": 1e3 * el / cycles, "algorithmic_bytes_per_cycle": alg,
           "roofline_frac": alg / (el / cycles) / 1e9 / hbm_peak_gbs, "hierarchy_build_s": t_build,
           "solve_to_1e-6": {"cycles": int(pc["nIterations"]), "seconds": t_conv},
           "timing": f"median of {repeats} mi_gamg_solve calls of {cycles} V-cycles (tolerance 0; the solve's prologue is inside the clock)"}
    return out, G


def timestep_supplement(eng, syn, case, addr, ctx, dev, gamg=None, steps=5, batched=None, coupled=None):
    """One PISO-like time step on the box (configs 4 / 5 per rank): stage times by HIP events, wall time per step.
    batched: solve the three momentum components with ONE multi-right-hand-side PBiCG (mi_pbicg_solve_multi) -- default: when
    the engine has it (MI_TIMESTEP_SEGREGATED=1 forces three single solves).
    coupled = dict(case=<LduCase with interfaces on addr's patches>, comms=(reduce, halo) | None, n_global=N): the step of a
    DECOMPOSED case -- both matrices carry the coupled patches' coefficients (re-bound every step like the internal ones);
    with communicators they are attached (every tile operator exchanges its halo, every sum is all-reduced, GAMG has
    processor interfaces on every level), without them the patches are local cyclic ones (the single-rank reference)."""
    import torch
    N, F = case.n_cells, case.n_faces
    nx = case.dims[0]
    h = 1.0 / nx
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    E = lambda n: torch.empty(n, dtype=torch.float64, device=dev)
    asm = eng.Assembly(addr)
    dirn = box_direction(case)
    Sf = [t(h * h * (dirn == d)) for d in range(3)]
    magSf, delta, vol = t(np.full(F, h * h)), t(np.full(F, 1.0 / h)), t(np.full(N, h ** 3))
    phi = t(0.3 * h * h * (dirn == 0) * (1.0 + 0.2 * (syn.splitmix_uniform(3, F) - 0.5)))
    nuMagSf = t(np.full(F, 1e-3 * h * h))
    U = [t(0.1 * (syn.splitmix_uniform(10 + d, N) - 0.5)) for d in range(3)]
    p = t(np.zeros(N))
    UM, PM = eng.Matrix(addr), eng.Matrix(addr)
    cpl = None
    if coupled is not None:
        itfs = coupled["case"].interfaces
        pr, pn = [i.nbr_domain for i in itfs], [i.nbr_patch for i in itfs]
        kU, kP = -1e-3 * h, 1e-3 * h            # off-diagonal of the momentum / pressure matrix across a coupled face
        cpl = [dict(patch=eng.Patch(ctx, N, itf.face_cells), dU=t(np.full(len(itf.face_cells), -kU)), bU=t(np.full(len(itf.face_cells), -kU)),
                    dP=t(np.full(len(itf.face_cells), -kP)), bP=t(np.full(len(itf.face_cells), -kP))) for itf in itfs]
        if coupled["comms"] is not None:
            # (coefficients must be bound before the first operator; the attach itself only needs the addressing)
            for M in (UM, PM):
                M.attach_comm(coupled["comms"][0], coupled["comms"][1], pr, pn, coupled["n_global"])
            if gamg is None:
                gamg = eng.Gamg(addr, box_pair_weights(case), 100, comms=coupled["comms"], patch_rank=pr, patch_nbr_patch=pn)
    G = gamg if gamg is not None else eng.Gamg(addr, box_pair_weights(case), 100)
    xmin = np.nonzero(np.arange(N) % nx == 0)[0]
    patch = eng.Patch(ctx, N, xmin)
    icU, icP = t(np.full(xmin.shape[0], 2.0 * 1e-3 * h)), t(np.full(xmin.shape[0], -2.0 * h))
    lam = t(np.full(F, 0.5))                     # linear interpolation weights of the uniform box
    ds, smag = [E(N) for _ in range(3)], E(N)
    ul, uu, ud = E(F), E(F), E(N)
    rAU, rAUf, pu, pd, psrc = E(N), E(F), E(F), E(N), E(N)
    fh, grad = E(F), [E(N) for _ in range(3)]
    if batched is None:
        batched = hasattr(UM, "pbicg_multi") and not os.environ.get("MI_TIMESTEP_SEGREGATED")
    ev = {}

    class stage:
        def __init__(self, name): self.name = name
        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True); self.b = torch.cuda.Event(enable_timing=True); self.a.record()
        def __exit__(self, *x):
            self.b.record(); ev.setdefault(self.name, []).append((self.a, self.b))

    def step():
        with stage("momentum: UEqn = fvm::ddt(U) + fvm::div(phi, U) [upwind] - fvm::laplacian(nu, U), three sources, in ONE pass (mi_fvm_assemble) + boundary diag"):
            asm.assemble(uu, ud, lower_out=ul, sources_out=ds, ddt=dict(r_delta_t=1.0 / 1e-3, vol=vol, psi_old=U), div=dict(flux=phi),
                         laplacian=dict(delta_coeffs=delta, gamma_magsf=nuMagSf), sum_mag_out=smag)
            patch.add(icU, ud, 0)
            if cpl:
                for q in cpl: q["patch"].add(q["dU"], ud, 0)
        with stage("momentum: relax(0.7), three components (sumMagOffDiag from the assembly pass) + bind coefficients"):
            asm.relax_multi(0.7, ud, None, None, ds, U, sum_mag=smag)
            UM.set_coeffs(ud, uu, ul)
            if cpl:
                for k, q in enumerate(cpl): UM.set_interface_coeffs(k, q["bU"], q["bU"])
        with stage("momentum: PBiCG + DILU, 3 components (relTol 0.1)" + (" -- one batched solve" if batched else "")):
            if batched:
                its = [q["nIterations"] for q in UM.pbicg_multi(U, ds, "DILU", tolerance=1e-12, relTol=0.1, maxIter=50)]
            else:
                its = [UM.pbicg(U[d], ds[d], "DILU", tolerance=1e-12, relTol=0.1, maxIter=50)["nIterations"] for d in range(3)]
        with stage("pressure: rAU, interpolate, fvm::laplacian(rAUf), bind, div(phi) source"):
            torch.reciprocal(ud, out=rAU); rAU.mul_(vol)
            asm.face_interpolate(lam, rAU, rAUf); rAUf.mul_(magSf)
            asm.fvm_laplacian(delta, rAUf, pu, pd); patch.add(icP, pd, 0)
            if cpl:
                for q in cpl: q["patch"].add(q["dP"], pd, 0)
            PM.set_coeffs(pd, pu, None)
            if cpl:
                for k, q in enumerate(cpl): PM.set_interface_coeffs(k, q["bP"], None)
            asm.surface_integrate(phi, None, psrc)
        with stage("pressure: GAMG (relTol 0.05)"):
            p.zero_()                   # same work every step: V-cycles from a zero start (a restart from the old p takes fewer)
            cyc = G.solve(PM, p, psrc, tolerance=1e-12, relTol=0.05, maxIter=50)["nIterations"]
        with stage("corrector: flux (faceH), fvc::grad(p), U -= rAU grad p"):
            PM.faceH(p, fh); phi.sub_(fh * 0.0)
            asm.face_interpolate(lam, p, rAUf); asm.gauss_grad(Sf, rAUf, vol, grad)
            for d in range(3): asm.axpby(1.0, U[d], -1e-3, grad[d], U[d])
        return its, cyc

    step(); torch.cuda.synchronize(); ev.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        its, cyc = step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    return {"workload": f"BASELINE configs 4/5, per-rank share: one PISO-like time step on the {case.dims[0]}x{case.dims[1]}x{case.dims[2]} box "
                        "(fvm::ddt + upwind fvm::div - fvm::laplacian assembled in one pass, relax, PBiCG + DILU x 3 components, pressure assembly, GAMG, flux + Gauss gradient correction), "
                        "every stage through the C ABI",
            "ms_per_time_step": 1e3 * wall, "pbicg_iterations_per_component": [int(v) for v in its], "gamg_cycles": int(cyc),
            "momentum_solve": "one batched 3-right-hand-side PBiCG" if batched else "three segregated PBiCG solves",
            "stages_ms": {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()},
            "timing": f"mean of {steps} steps after one warm-up step; stages by HIP events on the engine's stream"}


def rhopimple_supplement(eng, syn, case, addr, ctx, dev, gamg=None, steps=3, transonic=False, capture=None):
    """One rhoPimpleFoam time step (BASELINE config 5's solver, per-rank share) on the box: UEqn.H, EEqn.H, pEqn.H of
    applications/solvers/compressible/rhoPimpleFoam with one outer corrector, one PISO corrector and no non-orthogonal corrector --

      UEqn   fvm::ddt(rho, U) + fvm::div(phi, U) [upwind] - fvm::laplacian(muEff, U) == -fvc::grad(p); relax; PBiCG + DILU, the three
             components in one batched solve                                   (UEqn.H:3-21; the explicit part of divDevRhoReff is left out)
      EEqn   fvm::ddt(rho, he) + fvm::div(phi, he) + fvc::ddt(rho, K) + fvc::div(phi, K) - dpdt - fvm::laplacian(alphaEff, he); relax;
             PBiCG + DILU; rho = psi p                                                                                  (EEqn.H:4-33)
      pEqn   rAU = 1/A, rhorAUf = interpolate(rho rAU), HbyA = rAU H, fvc::ddtCorr(rho, U, phi) (mi_ddt_phi_corr),
             non-transonic: phiHbyA = (interpolate(rho HbyA) & Sf) + rhorAUf ddtCorr and fvc::div(phiHbyA) in ONE pass (mi_flux_div);
                            fvm::ddt(psi, p) + fvc::div(phiHbyA) - fvm::laplacian(rhorAUf, p): symmetric, GAMG       (pEqn.H:47-82)
             transonic:     phid = interpolate(psi) ((interpolate(HbyA) & Sf) + rhorAUf ddtCorr / interpolate(rho));
                            fvm::ddt(psi, p) + fvm::div(phid, p) - fvm::laplacian(rhorAUf, p): ASYMMETRIC, GAMG       (pEqn.H:17-46)
             phi = phiHbyA + pEqn.flux(); U = HbyA - rAU grad(p)                                                        (pEqn.H:78-101)

    every matrix operator, solver and face / row pass through the C ABI; the cell-wise algebra of the thermodynamics (K, rho = psi p,
    products with rAU) with torch element-wise kernels on the same stream.
    capture: a dict that receives host copies of the pressure system of the last step (diag, upper, lower | None, source, the start
    field, the solution and the solver's perf) -- tests/test_assembly.py solves it with the oracle's GAMG."""
    import torch
    N, F = case.n_cells, case.n_faces
    nx = case.dims[0]
    h = 1.0 / nx
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    E = lambda n: torch.empty(n, dtype=torch.float64, device=dev)
    asm = eng.Assembly(addr)
    dirn = box_direction(case)
    Sf = [t(h * h * (dirn == d)) for d in range(3)]
    magSf, delta, vol = t(np.full(F, h * h)), t(np.full(F, 1.0 / h)), t(np.full(N, h ** 3))
    lam = t(np.full(F, 0.5))                                        # linear interpolation weights of the uniform box
    dt = 1e-4
    rdt = 1.0 / dt
    R, T0, p0 = 287.0, 300.0, 1.0e5
    psi = t(np.full(N, 1.0 / (R * T0)) * (1.0 + 0.01 * (syn.splitmix_uniform(21, N) - 0.5)))          # compressibility rho = psi p
    p = t(p0 * (1.0 + 1e-3 * (syn.splitmix_uniform(22, N) - 0.5)))
    rho = psi * p
    U = [t(20.0 * (d == 0) + 2.0 * (syn.splitmix_uniform(10 + d, N) - 0.5)) for d in range(3)]
    he = t(1005.0 * T0 * (1.0 + 1e-3 * (syn.splitmix_uniform(23, N) - 0.5)))
    rho_f = E(F)
    asm.face_interpolate(lam, rho, rho_f)
    phi = rho_f * Sf[0] * 20.0 * t(1.0 + 0.05 * (syn.splitmix_uniform(3, F) - 0.5))                    # mass flux, mostly along x
    muMagSf, alphaMagSf = t(np.full(F, 1.8e-5 * h * h)), t(np.full(F, 2.5e-5 * h * h))
    UM, EM, PM = eng.Matrix(addr), eng.Matrix(addr), eng.Matrix(addr)
    G = gamg if gamg is not None else eng.Gamg(addr, box_pair_weights(case), 100)
    xmin = np.nonzero(np.arange(N) % nx == 0)[0]
    patch = eng.Patch(ctx, N, xmin)
    icU, icE, icP = t(np.full(xmin.shape[0], 2.0 * 1.8e-5 * h)), t(np.full(xmin.shape[0], 2.0 * 2.5e-5 * h)), t(np.full(xmin.shape[0], 2.0e-4 * h))
    wts, lu, ld = E(F), E(F), E(N)
    ds, smag = [E(N) for _ in range(3)], E(N)
    ul, uu, ud = E(F), E(F), E(N)
    el, eu, ed, es, expl, Kf = E(F), E(F), E(N), E(N), E(N), E(F)
    rAU, rhorAUf, pu, pl, pd, ps, pdd = E(N), E(F), E(F), E(F), E(N), E(N), E(N)
    ddtc, phiH, divH, fh, pf = E(F), E(F), E(N), E(F), E(F)
    grad, HbyA, tmpN = [E(N) for _ in range(3)], [E(N) for _ in range(3)], E(N)
    K = E(N)
    rho0, p_old = rho.clone(), p.clone()
    U0 = [u.clone() for u in U]
    K0 = 0.5 * (U[0] * U[0] + U[1] * U[1] + U[2] * U[2])
    ev = {}

    class stage:
        def __init__(self, name): self.name = name
        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True); self.b = torch.cuda.Event(enable_timing=True); self.a.record()
        def __exit__(self, *x):
            self.b.record(); ev.setdefault(self.name, []).append((self.a, self.b))

    def step():
        with stage("UEqn: fvc::grad(p), then fvm::ddt(rho, U) + fvm::div(phi, U) [upwind] - fvm::laplacian(muEff, U) == -grad(p) in ONE pass (mi_fvm_assemble)"):
            asm.face_interpolate(lam, p, pf); asm.gauss_grad(Sf, pf, vol, grad)
            asm.assemble(uu, ud, lower_out=ul, sources_out=ds, ddt=dict(r_delta_t=rdt, vol=vol, psi_old=U0, rho=rho, rho_old=rho0), div=dict(flux=phi),
                         laplacian=dict(delta_coeffs=delta, gamma_magsf=muMagSf), su=[(1.0, grad)], sum_mag_out=smag)   # == -grad(p): source -= V grad(p)
            patch.add(icU, ud, 0)
        with stage("UEqn: relax(0.7) + bind + PBiCG + DILU, 3 components in one batched solve (relTol 0.1)"):
            asm.relax_multi(0.7, ud, None, None, ds, U, sum_mag=smag)
            UM.set_coeffs(ud, uu, ul)
            its = [q["nIterations"] for q in UM.pbicg_multi(U, ds, "DILU", tolerance=1e-12, relTol=0.1, maxIter=50)]
        with stage("EEqn: fvm::ddt(rho, he) + fvm::div(phi, he) - fvm::laplacian(alphaEff, he) + explicit K / dpdt terms (fvm::Su) + relax + PBiCG"):
            torch.mul(U[0], U[0], out=K); K.addcmul_(U[1], U[1]).addcmul_(U[2], U[2]).mul_(0.5)
            asm.fvc_div(phi, None, K, vol, Kf, expl)                                   # fvc::div(phi, K), upwind: one face pass + the row sum
            torch.mul(rho, K, out=tmpN); tmpN.addcmul_(rho0, K0, value=-1.0).sub_(p).add_(p_old)
            expl.add_(tmpN, alpha=rdt)                                                  # + fvc::ddt(rho, K) - dpdt  (= rdt (rho K - rho0 K0 - (p - p_old)))
            asm.assemble(eu, ed, lower_out=el, sources_out=[es], ddt=dict(r_delta_t=rdt, vol=vol, psi_old=[he], rho=rho, rho_old=rho0), div=dict(flux=phi),
                         laplacian=dict(delta_coeffs=delta, gamma_magsf=alphaMagSf), su=[(1.0, [expl])], sum_mag_out=smag)   # the explicit terms stand on the left: source -= V expl
            patch.add(icE, ed, 0)
            asm.relax_multi(0.9, ed, None, None, [es], [he], sum_mag=smag)
            EM.set_coeffs(ed, eu, el)
            e_it = EM.pbicg(he, es, "DILU", tolerance=1e-12, relTol=0.1, maxIter=50)["nIterations"]
            torch.mul(psi, p, out=rho)                                                 # thermo.correct(): rho = psi p
        with stage("pEqn: rAU, rhorAUf, HbyA = rAU H, fvc::ddtCorr(rho, U, phi)"):
            torch.div(vol, ud, out=rAU)                                                # 1 / UEqn.A(), A = D / V
            torch.mul(rho, rAU, out=tmpN); asm.face_interpolate(lam, tmpN, rhorAUf)
            for d in range(3):
                UM.H(U[d], HbyA[d]); HbyA[d].add_(ds[d]).div_(vol).mul_(rAU)            # fvMatrix::H = (lduMatrix::H + source) / V
            asm.ddt_phi_corr(rdt, lam, Sf, U0, rho0, phi, ddtc)
        if not transonic:
            with stage("pEqn: phiHbyA + fvc::div(phiHbyA) in one pass, fvm::ddt(psi, p) - fvm::laplacian(rhorAUf, p), bind"):
                asm.flux_div(lam, Sf, HbyA, phiH, divH, cell_scale=rho, add_a=rhorAUf, add_b=ddtc, vol=vol)
                rhorAUf.mul_(magSf)
                asm.assemble(pu, pd, sources_out=[ps], ddt=dict(r_delta_t=rdt, vol=vol, psi_old=[p_old], rho=psi, rho_old=psi),
                             laplacian=dict(delta_coeffs=delta, gamma_magsf=rhorAUf), su=[(1.0, [divH])])   # ddt(psi, p) + fvc::div(phiHbyA) - laplacian in one pass
                patch.add(icP, pd, 0)
                PM.set_coeffs(pd, pu, None)
        else:
            with stage("pEqn (transonic): phid, fvm::ddt(psi, p) + fvm::div(phid, p) - fvm::laplacian(rhorAUf, p): asymmetric, bind"):
                asm.flux_div(lam, Sf, HbyA, phiH, divH)                                  # interpolate(HbyA) & Sf
                asm.face_interpolate(lam, rho, rho_f); asm.face_interpolate(lam, psi, pf)
                phiH.add_(rhorAUf * ddtc / rho_f).mul_(pf)                               # phid
                rhorAUf.mul_(magSf)
                asm.assemble(pu, pd, lower_out=pl, sources_out=[ps], ddt=dict(r_delta_t=rdt, vol=vol, psi_old=[p_old], rho=psi, rho_old=psi), div=dict(flux=phiH),
                             laplacian=dict(delta_coeffs=delta, gamma_magsf=rhorAUf))               # ddt(psi, p) + div(phid, p) [upwind] - laplacian in one pass
                patch.add(icP, pd, 0)
                PM.set_coeffs(pd, pu, pl)
        if capture is not None:
            torch.cuda.synchronize()
            capture.update(diag=pd.cpu().numpy().copy(), upper=pu.cpu().numpy().copy(), lower=pl.cpu().numpy().copy() if transonic else None,
                           source=ps.cpu().numpy().copy(), start=p.cpu().numpy().copy())
        with stage("pEqn: GAMG (relTol 0.05)"):
            perf = G.solve(PM, p, ps, tolerance=1e-12, relTol=0.05, maxIter=50)
            cyc = perf["nIterations"]
        if capture is not None:
            torch.cuda.synchronize()
            capture.update(solution=p.cpu().numpy().copy(), perf=perf)
        with stage("corrector: phi = phiHbyA + pEqn.flux(), U = HbyA - rAU grad(p)"):
            PM.faceH(p, fh)
            if not transonic:
                torch.sub(phiH, fh, out=Kf)                                            # (the step's phi; `phi` itself stays the step's input: same work every step)
            asm.face_interpolate(lam, p, pf); asm.gauss_grad(Sf, pf, vol, grad)
            for d in range(3): torch.addcmul(HbyA[d], rAU, grad[d], value=-1.0, out=tmpN)
        return its, e_it, cyc

    he0 = he.clone()

    def reset():
        p.copy_(p_old); torch.mul(psi, p, out=rho); he.copy_(he0)
        for d in range(3): U[d].copy_(U0[d])

    step(); reset(); torch.cuda.synchronize(); ev.clear()
    wall = 0.0
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        its, e_it, cyc = step()
        torch.cuda.synchronize(); wall += time.perf_counter() - t0
        reset()
    wall /= steps
    return {"workload": f"BASELINE config 5's solver, per-rank share: one rhoPimpleFoam time step ({'transonic' if transonic else 'non-transonic'} pEqn) on the "
                        f"{case.dims[0]}x{case.dims[1]}x{case.dims[2]} box -- UEqn (fvm::ddt(rho,U) + upwind fvm::div(phi,U) - fvm::laplacian(muEff,U) == -grad p, relax, "
                        "batched PBiCG + DILU), EEqn (fvm::ddt(rho,he) + fvm::div - fvm::laplacian + explicit K / dpdt terms, PBiCG + DILU), pEqn "
                        + ("(fvm::ddt(psi,p) + fvm::div(phid,p) - fvm::laplacian(rhorAUf,p): asymmetric GAMG)" if transonic else
                           "(fvc::ddtCorr, phiHbyA + its divergence in one pass, fvm::ddt(psi,p) - fvm::laplacian(rhorAUf,p): GAMG)") + ", flux and velocity correction",
            "ms_per_time_step": 1e3 * wall, "pbicg_iterations_per_component": [int(v) for v in its], "energy_pbicg_iterations": int(e_it), "gamg_cycles": int(cyc),
            "pressure_matrix": "asymmetric" if transonic else "symmetric",
            "stages_ms": {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()},
            "timing": f"mean of {steps} steps after one warm-up step (fields reset between steps: same work every step); stages by HIP events on the engine's stream"}


def decomposed_supplements(eng, syn, par, sub, ctx, dev, comms, n_global, cycles=10, steps=3, reduce_max=lambda v: v):
    """BASELINE configs 3 / 4 / 5 as DECOMPOSED workloads, per rank (`bench.py --gpus N` measured diagonal PCG
    only): `sub` is this rank's sub-domain with its processor patches, `comms` the (reduce, halo) communicators the PCG bench
    already runs on.  GAMG with processor interfaces on every level (V-cycles/s, max over ranks) and one PISO-like time step with
    attached matrices (batched momentum PBiCG, GAMG pressure).  All ranks call together; reduce_max(x) = max of x over the ranks."""
    import torch
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = {}
    dm = par.DistributedMatrix(ctx, sub, dev, n_global=n_global, comms=comms)
    w = box_pair_weights(sub)
    t0 = time.perf_counter()
    G = dm.gamg(w, 100)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    src = t(sub.source)
    psi = torch.zeros(sub.n_cells, dtype=torch.float64, device=dev)
    G.solve(dm.mat, psi, src, tolerance=0.0, maxIter=3)
    g0 = ctx.stat(3)
    rep = []
    for _ in range(3):
        psi.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        perf = G.solve(dm.mat, psi, src, tolerance=0.0, maxIter=cycles)
        torch.cuda.synchronize()
        rep.append(reduce_max(time.perf_counter() - t0))
        assert perf["nIterations"] == cycles, perf
    el = float(np.median(rep))
    used, bad = dm.mat.peer_halo_status()
    out["gamg"] = {"workload": f"GAMG pressure solve of the decomposed box, {sub.n_cells} cells on this rank, {G.n_levels} levels with processor interfaces on every level, "
                               "global coarsest system", "ms_per_v_cycle": 1e3 * el / cycles, "v_cycles_per_s": cycles / el,
                   "cycles_replayed_as_hipGraph": int(ctx.stat(3) - g0), "halo_through_peer_windows": bool(used), "wait_timeouts": int(bad),
                   "hierarchy_build_s": reduce_max(t_build), "timing": f"median of 3 mi_gamg_solve calls of {cycles} V-cycles, max over ranks"}
    ts = timestep_supplement(eng, syn, sub, dm.addr, ctx, dev, gamg=G, steps=steps, coupled=dict(case=sub, comms=comms, n_global=n_global))
    ts["ms_per_time_step"] = reduce_max(ts["ms_per_time_step"])
    out["timestep"] = ts
    return out
