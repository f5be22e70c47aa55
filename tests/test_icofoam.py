"""GPU (-m gpu): rapidcfd-dev_amd/foam/icoFoam.C -- BASELINE config 1's application (icoFoam, lid-driven cavity) on the engine.

The test writes a cavity case the way OpenFOAM lays one out (constant/polyMesh, constant/transportProperties, system/controlDict,
system/fvSchemes, system/fvSolution, 0/U, 0/p), runs the application, and walks the SAME statements of icoFoam.C on the oracle
(numpy + oracle/*.c: fvm::ddt / div / laplacian, the boundary coefficients of fixedValue patches, PBiCG + DILU for the momentum
components, rAU, H, ddtCorr, phiHbyA and its divergence, fvm::laplacian(rAU, p) with setReference, PCG + DIC, flux, continuity errors, the
velocity correction).  Every solver line of every step (solver name, initial and final residual, iteration count), the continuity errors
and the U and p files the application writes back into the case must agree."""
import os
import re
import subprocess

import numpy as np
import pytest

from test_polymesh import PKG, HEADER, make_box_mesh, geometry, write_case, LINE, read_vol_field

pytestmark = pytest.mark.gpu


def write_cavity(case_dir, dims, nu, delta_t, n_steps, div_scheme, write_format="binary", corrected=False):
    pts, faces, owner, neighbour, patches = make_box_mesh(dims, seed=5 if corrected else None)   # corrected: a distorted, non-orthogonal box
    n = int(owner.max()) + 1
    write_case(case_dir, pts, faces, owner, neighbour, patches, np.zeros(n), False)
    hd = lambda cls, loc, obj: HEADER.format(fmt="ascii", cls=cls, note="", obj=obj).replace('location    "constant/polyMesh"', f'location    "{loc}"')
    os.makedirs(os.path.join(case_dir, "system"), exist_ok=True)
    open(os.path.join(case_dir, "constant", "transportProperties"), "w").write(hd("dictionary", "constant", "transportProperties") + f"nu              nu [0 2 -1 0 0 0 0] {nu!r};\n")
    open(os.path.join(case_dir, "system", "controlDict"), "w").write(hd("dictionary", "system", "controlDict") + f"""application     icoFoam;
startFrom       startTime;
startTime       0;
stopAt          endTime;
endTime         {delta_t * n_steps!r};
deltaT          {delta_t!r};
writeControl    timeStep;
writeInterval   {n_steps};
writeFormat     {write_format};
writePrecision  12;
""")
    open(os.path.join(case_dir, "system", "fvSchemes"), "w").write(hd("dictionary", "system", "fvSchemes") + f"""ddtSchemes {{ default Euler; }}
gradSchemes {{ default Gauss linear; }}
divSchemes {{ default none; div(phi,U) Gauss {div_scheme}; }}
laplacianSchemes {{ default Gauss linear {'corrected' if corrected else 'orthogonal'}; }}
interpolationSchemes {{ default linear; }}
snGradSchemes {{ default {'corrected' if corrected else 'orthogonal'}; }}
""")
    open(os.path.join(case_dir, "system", "fvSolution"), "w").write(hd("dictionary", "system", "fvSolution") + """solvers
{
    p       { solver PCG; preconditioner DIC; tolerance 1e-08; relTol 0.05; }
    pFinal  { $p; relTol 0; }
    "U.*"   { solver PBiCG; preconditioner DILU; tolerance 1e-09; relTol 0; }
}
PISO { nCorrectors 2; nNonOrthogonalCorrectors NONORTH; pRefCell 0; pRefValue 0; }
""".replace("NONORTH", "1" if corrected else "0"))
    bU = "".join(f"    {name}\n    {{\n        type            {'fixedValue' if name != 'walls' else 'noSlip'};\n"
                 + ("        value           uniform (0 1 0);\n" if name == "inlet" else "        value           uniform (0 0 0);\n" if name == "outlet" else "") + "    }\n"
                 for name, _, _, _ in patches)
    open(os.path.join(case_dir, "0", "U"), "w").write(hd("volVectorField", "0", "U") + "dimensions      [0 1 -1 0 0 0 0];\n\ninternalField   uniform (0 0 0);\n\nboundaryField\n{\n" + bU + "}\n")
    bp = "".join(f"    {name}\n    {{\n        type            zeroGradient;\n    }}\n" for name, _, _, _ in patches)
    open(os.path.join(case_dir, "0", "p"), "w").write(hd("volScalarField", "0", "p") + "dimensions      [0 2 -2 0 0 0 0];\n\ninternalField   uniform 0;\n\nboundaryField\n{\n" + bp + "}\n")
    return pts, faces, owner, neighbour, patches


def oracle_icofoam(pkg, orc, pts, faces, owner, neighbour, patches, nu, delta_t, n_steps, scheme, corrected=False):
    """icoFoam.C on the oracle; returns the solver lines [(name, field, initial, final, iterations)], the continuity errors and U, p"""
    syn = pkg.synthetic
    G = geometry(pts, faces, owner, neighbour)
    n, nI = int(owner.max()) + 1, len(neighbour)
    lo, up = owner[:nI].astype(np.int32), neighbour.astype(np.int32)
    V, lam, delta, magSf = G["V"], G["weights"], G["delta"], G["magSf"][:nI]
    Sf = [np.ascontiguousarray(G["Sf"][:nI, k]) for k in range(3)]
    P = []
    for name, ptype, cnt, start in patches:
        fc = owner[start:start + cnt].astype(np.int32)
        ub = np.tile(np.array([0.0, 1.0, 0.0]) if name == "inlet" else np.zeros(3), (cnt, 1))
        sfb = G["Sf"][start:start + cnt]
        P.append(dict(fc=fc, ub=ub, sf=[np.ascontiguousarray(sfb[:, k]) for k in range(3)],
                      phi=ub[:, 0] * sfb[:, 0] + ub[:, 1] * sfb[:, 1] + ub[:, 2] * sfb[:, 2],
                      diff=nu * G["magSf"][start:start + cnt] * G["delta_b"][start - nI:start - nI + cnt]))
    U = [np.zeros(n) for _ in range(3)]
    p = np.zeros(n)
    phi = orc.flux_div(n, lo, up, lam, Sf, U, want_div=False)
    r_dt = 1.0 / delta_t
    lines, cont, cumulative, totalV = [], [], 0.0, float(np.sum(V))

    def grad_p():
        g = orc.gauss_grad(n, lo, up, Sf, orc.face_interpolate(lo, up, lam, p), None)
        for q in P:
            for k in range(3):
                g[k] = orc.patch_add_product(q["fc"], q["sf"][k], p[q["fc"]], g[k], 0)
        return [x / V for x in g]

    centres = [np.ascontiguousarray(G["C"][:, k]) for k in range(3)]
    nhat = G["Sf"][:nI] / magSf[:, None]
    cv = nhat - (G["C"][up] - G["C"][lo]) * delta[:, None]                            # nonOrthCorrectionVectors (surfaceInterpolation.C:498-580)
    cv = [np.ascontiguousarray(cv[:, k]) for k in range(3)]

    def full_grad(vf, patch_values):
        g = orc.gauss_grad(n, lo, up, Sf, orc.face_interpolate(lo, up, lam, vf), None)
        for q, pv in zip(P, patch_values):
            for k in range(3):
                g[k] = orc.patch_add_product(q["fc"], q["sf"][k], vf[q["fc"]] if pv is None else pv, g[k], 0)
        return [x / V for x in g]
    for step in range(n_steps):
        Uold, phiOld = [u.copy() for u in U], phi.copy()
        if scheme == "upwind":
            w = orc.upwind_weights(phi)
        elif scheme.startswith("limitedLinear"):     # one limiter for the three components, from magSqr(U) and its Gauss gradient
            m2 = (U[0] * U[0] + U[1] * U[1]) + U[2] * U[2]
            g = orc.gauss_grad(n, lo, up, Sf, orc.face_interpolate(lo, up, lam, m2), None)
            for q in P:
                mb = (q["ub"][:, 0] * q["ub"][:, 0] + q["ub"][:, 1] * q["ub"][:, 1]) + q["ub"][:, 2] * q["ub"][:, 2]
                for k in range(3):
                    g[k] = orc.patch_add_product(q["fc"], q["sf"][k], mb, g[k], 0)
            w, _ = orc.limited_linear_weights(lo, up, float(scheme.split()[1]), lam, phi, m2, [x / V for x in g], centres)
        else:
            w = lam
        lB, uB, dB = orc.fvm_div(n, lo, up, w, phi)
        uL, dL = orc.fvm_laplacian(n, lo, up, delta, nu * magSf)
        lower, upper = lB - uL, uB - uL
        gp = grad_p()
        mats = []
        for k in range(3):
            dD, sD = orc.fvm_ddt_euler(r_dt, 1.0, V, Uold[k])
            diag, source = (dD + dB) - dL, sD
            ic = [q["diff"] for q in P]
            bc = [q["diff"] * q["ub"][:, k] - q["phi"] * q["ub"][:, k] for q in P]
            if corrected:   # - fvm::laplacian(nu, U) corrected: source += V*div(nu |Sf| correction(U_k))
                gU = full_grad(Uold[k], [q["ub"][:, k].copy() for q in P])
                cf = orc.sngrad_correction_flux(lo, up, cv, lam, gU, -(nu * magSf))
                source = orc.submul(V, orc.surface_integrate(n, lo, up, cf, V), source)
            mats.append(dict(diag=diag, source=source, ic=ic, bc=bc))
        for k in range(3):
            M = mats[k]
            dtot, stot = M["diag"].copy(), M["source"] - V * gp[k]
            for q, ic, bc in zip(P, M["ic"], M["bc"]):
                dtot = orc.patch_add(q["fc"], ic, dtot, 0); stot = orc.patch_add(q["fc"], bc, stot, 0)
            U[k], perf = orc.System([syn.LduCase(n, lo, up, dtot, upper, lower, stot)]).pbicg(U[k], stot, "AINV", tolerance=1e-9, relTol=0.0)
            lines.append(("AINVPBiCG", "Ux Uy Uz".split()[k], perf["initialResidual"], perf["finalResidual"], perf["nIterations"]))
        for corr in range(2):
            A = mats[0]["diag"].copy()
            for q in P:                                                               # D() = diag + cmptAv(internalCoeffs): (x + y + z)/3 as floating point does it
                A = orc.patch_add(q["fc"], ((q["diff"] + q["diff"]) + q["diff"]) / 3.0, A, 0)
            rAU = 1.0 / (A / V)
            HbyA = []
            for k in range(3):
                M = mats[k]
                H = orc.System([syn.LduCase(n, lo, up, M["diag"], upper, lower, M["source"])]).H(U[k]) + M["source"]
                for q, bc in zip(P, M["bc"]):
                    H = orc.patch_add(q["fc"], bc, H, 0)
                HbyA.append(rAU * (H / V))
            rAUf = orc.face_interpolate(lo, up, lam, rAU)
            ddtc = orc.ddt_phi_corr(lo, up, r_dt, lam, Sf, Uold, None, phiOld)
            phiHbyA, div = orc.flux_div(n, lo, up, lam, Sf, HbyA, None, rAUf, ddtc, None, True)
            for q in P:
                div = orc.patch_add(q["fc"], q["phi"], div, 0)
            for non_orth in range(2 if corrected else 1):
                upP, dP = orc.fvm_laplacian(n, lo, up, delta, rAUf * magSf)
                sP = div.copy()
                if corrected:
                    cfp = orc.sngrad_correction_flux(lo, up, cv, lam, grad_p(), rAUf * magSf)
                    sP = orc.submul(V, orc.surface_integrate(n, lo, up, cfp, V), sP)
                sP[0] += dP[0] * 0.0; dP = dP.copy(); dP[0] += dP[0]                    # setReference(0, 0): source += diag*value; diag += diag
                final = corr == 1 and non_orth == (1 if corrected else 0)
                p, perf = orc.System([syn.LduCase(n, lo, up, dP, upP, None, sP)]).pcg(p, sP, "AINV", tolerance=1e-8, relTol=0.0 if final else 0.05)
                lines.append(("AINVPCG", "p", perf["initialResidual"], perf["finalResidual"], perf["nIterations"]))
            flux = orc.System([syn.LduCase(n, lo, up, dP, upP, None, sP)]).faceH(p)
            phi = phiHbyA - ((flux + cfp) if corrected else flux)
            ce = orc.surface_integrate(n, lo, up, phi, None)
            for q in P:
                ce = orc.patch_add(q["fc"], q["phi"], ce, 0)
            loc, glob = float(np.sum(np.abs(ce))) * delta_t / totalV, float(np.sum(ce)) * delta_t / totalV
            cumulative += glob
            cont.append((loc, glob, cumulative))
            gp = grad_p()
            U = [HbyA[k] - rAU * gp[k] for k in range(3)]
    return lines, cont, U, p


def test_icoFoam_refuses_what_it_does_not_assemble(pkg, tmp_path):
    """a `corrected` Laplacian, another ddt scheme, a p patch that is not zeroGradient: an error that says so, not a silently different equation"""
    case_dir = str(tmp_path / "cavity")
    write_cavity(case_dir, (4, 3, 2), 0.01, 0.005, 1, "linear", "ascii")
    sch = os.path.join(case_dir, "system", "fvSchemes")
    good = open(sch).read()
    for bad, msg in ((good.replace("Gauss linear orthogonal", "Gauss linear limited 0.5"), "Gauss linear corrected | uncorrected | orthogonal"),
                     (good.replace("default Euler", "default backward"), "only Euler"), (good.replace("Gauss linear;  ", "Gauss QUICK;"), None)):
        open(sch, "w").write(bad)
        out = subprocess.run([os.path.join(PKG, "icoFoam"), case_dir], capture_output=True, text=True, timeout=120)
        if msg:
            assert out.returncode != 0 and msg in out.stderr, out.stderr
    open(sch, "w").write(good)
    pf = os.path.join(case_dir, "0", "p")
    text = open(pf).read()
    open(pf, "w").write(text.replace("type            zeroGradient;", "type            fixedValue; value uniform 0;", 1))
    out = subprocess.run([os.path.join(PKG, "icoFoam"), case_dir], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "zeroGradient patches only" in out.stderr


CONT = re.compile(r"time step continuity errors : sum local = (\S+), global = (\S+), cumulative = (\S+)")


@pytest.mark.parametrize("dims, n_steps, div_scheme, write_format, corrected", [((10, 8, 6), 3, "linear", "binary", False), ((10, 8, 6), 3, "upwind", "ascii", False),
                                                                                ((12, 10, 8), 4, "limitedLinear 1", "binary", False),
                                                                                ((12, 9, 7), 3, "linear", "binary", True),            # a distorted box, `corrected` Laplacians, one non-orthogonal corrector
                                                                                ((32, 32, 32), 2, "linear", "binary", False)])        # BASELINE config 1's size
def test_icoFoam_cavity_matches_the_oracle_statement_for_statement(pkg, orc, tmp_path, dims, n_steps, div_scheme, write_format, corrected):
    nu, delta_t = 0.01, 0.005
    case_dir = str(tmp_path / "cavity")
    pts, faces, owner, neighbour, patches = write_cavity(case_dir, dims, nu, delta_t, n_steps, div_scheme, write_format, corrected)
    out = subprocess.run([os.path.join(PKG, "icoFoam"), case_dir], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout[-2000:]
    got = [(m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))) for m in map(LINE.match, out.stdout.splitlines()) if m]
    cont = [tuple(map(float, m.groups())) for m in map(CONT.match, out.stdout.splitlines()) if m]
    ref_lines, ref_cont, refU, refp = oracle_icofoam(pkg, orc, pts, faces, owner, neighbour, patches, nu, delta_t, n_steps, div_scheme, corrected)
    assert len(got) == len(ref_lines) == n_steps * (7 if corrected else 5) and len(cont) == len(ref_cont) == n_steps * 2
    for g, r in zip(got, ref_lines):
        assert g[0] == r[0] and g[1] == r[1], (g, r)
        assert g[4] == r[4], (g, r)                                                   # the same iteration counts
        assert abs(g[2] - r[2]) <= 1e-7 * max(r[2], 1e-12) + 1e-14 and abs(g[3] - r[3]) <= 1e-6 * max(r[2], 1e-12) + 1e-14, (g, r)
    for g, r in zip(cont, ref_cont):
        assert abs(g[0] - r[0]) <= 1e-6 * r[0] + 1e-16 and abs(g[1] - r[1]) < 1e-15 and abs(g[2] - r[2]) < 1e-15, (g, r)
    assert all(c[0] < 1e-6 for c in cont[1::2])                                       # the final corrector closes continuity to the solver's tolerance
    tn = f"{n_steps * delta_t:.10g}"
    fU, fp = read_vol_field(os.path.join(case_dir, tn, "U")), read_vol_field(os.path.join(case_dir, tn, "p"))
    assert fU["header"]["class"] == "volVectorField" and fU["header"]["format"] == write_format and fp["header"]["class"] == "volScalarField"
    tolU = 1e-7 if write_format == "binary" else 1e-6
    Uref = np.stack(refU, axis=1)
    assert fU["internalField"].shape == Uref.shape and np.max(np.abs(fU["internalField"] - Uref)) <= tolU * np.max(np.abs(Uref))
    assert np.max(np.abs(fp["internalField"] - refp)) <= tolU * np.max(np.abs(refp))
    assert np.max(np.abs(Uref)) > 1e-3                                                # the lid really drives a flow
    if corrected:                                                                      # ... the non-orthogonal correction really corrects
        _, _, Uun, _ = oracle_icofoam(pkg, orc, pts, faces, owner, neighbour, patches, nu, delta_t, n_steps, div_scheme, False)
        assert np.max(np.abs(np.stack(Uun, axis=1) - Uref)) > 1e-6 * np.max(np.abs(Uref))
    if div_scheme.startswith("limitedLinear"):                                         # ... and the limiter really limits: not the linear scheme's result
        _, _, Ulin, _ = oracle_icofoam(pkg, orc, pts, faces, owner, neighbour, patches, nu, delta_t, n_steps, "linear")
        assert np.max(np.abs(np.stack(Ulin, axis=1) - Uref)) > 1e-7 * np.max(np.abs(Uref))
    bf = dict(fU["boundaryField"])
    assert bf["inlet"]["type"] == "fixedValue" and bf["inlet"]["value"] == ("uniform", [0.0, 1.0, 0.0]) and bf["walls"] == {"type": "noSlip"}
