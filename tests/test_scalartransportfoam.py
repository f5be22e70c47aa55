"""GPU (-m gpu): rapidcfd-dev_amd/foam/scalarTransportFoam.C -- applications/solvers/basic/scalarTransportFoam on the engine.

A channel case laid out as OpenFOAM lays one out (fixedValue inflow, zeroGradient outflow and walls; a uniform velocity in 0/U), the
application's  solve(fvm::ddt(T) + fvm::div(phi, T) - fvm::laplacian(DT, T))  per time step and non-orthogonal corrector, and the same
statements walked on the oracle: one fused assembly on the device against ddt + div - laplacian array by array, the patch coefficients of
gaussConvectionScheme / gaussLaplacianScheme for inflow AND outflow patches, upwind / linear / limitedLinear (the scalar NVD/TVD limiter
from T and its gradient), the `corrected` Laplacian on a distorted box, PBiCG + DILU.  Every solver line and the written T must agree."""
import os
import subprocess

import numpy as np
import pytest

from test_polymesh import PKG, HEADER, make_box_mesh, geometry, write_case, LINE, read_vol_field

pytestmark = pytest.mark.gpu


def write_channel(case_dir, dims, DT, delta_t, n_steps, div_scheme, corrected, n_non_orth, smooth_start=False):
    pts, faces, owner, neighbour, patches = make_box_mesh(dims, seed=5 if corrected else None)
    n = int(owner.max()) + 1
    # a TVD limiter is decided by SIGNS of differences: from a uniform start they are signs of rounding noise (the geometry of the application and
    # numpy's agree to the last bits, not to the bit), so the limited case starts from a smooth non-uniform field
    Cc = geometry(pts, faces, owner, neighbour)["C"]
    T0 = 0.6 + 0.25 * np.sin(3.0 * Cc[:, 0] + 2.0 * Cc[:, 1] + Cc[:, 2]) if smooth_start else None
    write_case(case_dir, pts, faces, owner, neighbour, patches, np.zeros(n), False)
    hd = lambda cls, loc, obj: HEADER.format(fmt="ascii", cls=cls, note="", obj=obj).replace('location    "constant/polyMesh"', f'location    "{loc}"')
    os.makedirs(os.path.join(case_dir, "system"), exist_ok=True)
    open(os.path.join(case_dir, "constant", "transportProperties"), "w").write(hd("dictionary", "constant", "transportProperties") + f"DT              DT [0 2 -1 0 0 0 0] {DT!r};\n")
    open(os.path.join(case_dir, "system", "controlDict"), "w").write(hd("dictionary", "system", "controlDict") +
        f"application scalarTransportFoam;\nstartTime 0;\nendTime {delta_t * n_steps!r};\ndeltaT {delta_t!r};\nwriteFormat binary;\nwritePrecision 12;\n")
    open(os.path.join(case_dir, "system", "fvSchemes"), "w").write(hd("dictionary", "system", "fvSchemes") + f"""ddtSchemes {{ default Euler; }}
gradSchemes {{ default Gauss linear; }}
divSchemes {{ default none; div(phi,T) Gauss {div_scheme}; }}
laplacianSchemes {{ default none; laplacian(DT,T) Gauss linear {'corrected' if corrected else 'uncorrected'}; }}
interpolationSchemes {{ default linear; }}
snGradSchemes {{ default {'corrected' if corrected else 'uncorrected'}; }}
""")
    open(os.path.join(case_dir, "system", "fvSolution"), "w").write(hd("dictionary", "system", "fvSolution") +
        "solvers { T { solver PBiCG; preconditioner DILU; tolerance 1e-10; relTol 0; } }\nSIMPLE { nNonOrthogonalCorrectors " + str(n_non_orth) + "; }\n")
    bU = "".join(f"    {name}\n    {{\n        type            {dict(inlet='fixedValue', outlet='zeroGradient', walls='noSlip')[name]};\n"
                 + ("        value           uniform (1 0.2 0);\n" if name == "inlet" else "") + "    }\n" for name, _, _, _ in patches)
    open(os.path.join(case_dir, "0", "U"), "w").write(hd("volVectorField", "0", "U") + "dimensions      [0 1 -1 0 0 0 0];\n\ninternalField   uniform (1 0.2 0);\n\nboundaryField\n{\n" + bU + "}\n")
    cnt_in = [pt[2] for pt in patches if pt[0] == "inlet"][0]
    tin = 1.0 + 0.5 * np.sin(np.arange(cnt_in))                        # a non-uniform inflow profile: `value nonuniform List<scalar>`
    bT = ""
    for name, _, _, _ in patches:
        bT += f"    {name}\n    {{\n        type            {'fixedValue' if name == 'inlet' else 'zeroGradient'};\n"
        if name == "inlet":
            bT += f"        value           nonuniform List<scalar> \n{cnt_in}\n(\n" + "\n".join(repr(float(v)) for v in tin) + "\n)\n;\n"
        bT += "    }\n"
    internal = "uniform 0;" if T0 is None else f"nonuniform List<scalar> \n{n}\n(\n" + "\n".join(repr(float(v)) for v in T0) + "\n)\n;"
    open(os.path.join(case_dir, "0", "T"), "w").write(hd("volScalarField", "0", "T") + "dimensions      [0 0 0 1 0 0 0];\n\ninternalField   " + internal + "\n\nboundaryField\n{\n" + bT + "}\n")
    return pts, faces, owner, neighbour, patches, tin, (np.zeros(n) if T0 is None else T0)


def oracle_scalar_transport(pkg, orc, pts, faces, owner, neighbour, patches, tin, T0, DT, delta_t, n_steps, scheme, corrected, n_non_orth):
    syn = pkg.synthetic
    G = geometry(pts, faces, owner, neighbour)
    n, nI = int(owner.max()) + 1, len(neighbour)
    lo, up = owner[:nI].astype(np.int32), neighbour.astype(np.int32)
    V, lam, delta, magSf = G["V"], G["weights"], G["delta"], G["magSf"][:nI]
    Sf = [np.ascontiguousarray(G["Sf"][:nI, k]) for k in range(3)]
    centres = [np.ascontiguousarray(G["C"][:, k]) for k in range(3)]
    nhat = G["Sf"][:nI] / magSf[:, None]
    cv = nhat - (G["C"][up] - G["C"][lo]) * delta[:, None]
    cv = [np.ascontiguousarray(cv[:, k]) for k in range(3)]
    u0 = np.array([1.0, 0.2, 0.0])
    U = [np.full(n, u0[k]) for k in range(3)]
    phi = orc.flux_div(n, lo, up, lam, Sf, U, want_div=False)
    P = []
    for name, ptype, cnt, start in patches:
        fc = owner[start:start + cnt].astype(np.int32)
        sfb = G["Sf"][start:start + cnt]
        ub = np.tile(u0, (cnt, 1)) if name in ("inlet", "outlet") else np.zeros((cnt, 3))
        phib = ub[:, 0] * sfb[:, 0] + ub[:, 1] * sfb[:, 1] + ub[:, 2] * sfb[:, 2]
        diff = DT * G["magSf"][start:start + cnt] * G["delta_b"][start - nI:start - nI + cnt]
        fixed = name == "inlet"
        tb = tin if fixed else None
        P.append(dict(fc=fc, sf=[np.ascontiguousarray(sfb[:, k]) for k in range(3)], tb=tb,
                      ic=diff if fixed else phib, bc=(diff * tb - phib * tb) if fixed else np.zeros(cnt)))

    def grad(T):
        g = orc.gauss_grad(n, lo, up, Sf, orc.face_interpolate(lo, up, lam, T), None)
        for q in P:
            for k in range(3):
                g[k] = orc.patch_add_product(q["fc"], q["sf"][k], T[q["fc"]] if q["tb"] is None else q["tb"], g[k], 0)
        return [x / V for x in g]

    T = T0.copy()
    uL, dL = orc.fvm_laplacian(n, lo, up, delta, DT * magSf)
    lines = []
    for step in range(n_steps):
        Told = T.copy()
        for non_orth in range(n_non_orth + 1):
            gT = grad(T) if (corrected or scheme.startswith("limitedLinear")) else None
            if scheme == "upwind":
                w = orc.upwind_weights(phi)
            elif scheme.startswith("limitedLinear"):
                w, _ = orc.limited_linear_weights(lo, up, float(scheme.split()[1]), lam, phi, T, gT, centres)
            else:
                w = lam
            lB, uB, dB = orc.fvm_div(n, lo, up, w, phi)
            dD, sD = orc.fvm_ddt_euler(1.0 / delta_t, 1.0, V, Told)
            lower, upper, diag, source = lB - uL, uB - uL, (dD + dB) - dL, sD
            if corrected:
                cf = orc.sngrad_correction_flux(lo, up, cv, lam, gT, -(DT * magSf))
                source = orc.submul(V, orc.surface_integrate(n, lo, up, cf, V), source)
            for q in P:
                diag = orc.patch_add(q["fc"], q["ic"], diag, 0); source = orc.patch_add(q["fc"], q["bc"], source, 0)
            T, perf = orc.System([syn.LduCase(n, lo, up, diag, upper, lower, source)]).pbicg(T, source, "AINV", tolerance=1e-10, relTol=0.0)
            lines.append(("AINVPBiCG", "T", perf["initialResidual"], perf["finalResidual"], perf["nIterations"]))
    return lines, T


@pytest.mark.parametrize("dims, n_steps, scheme, corrected, n_non_orth", [((12, 8, 6), 4, "upwind", False, 0), ((12, 8, 6), 4, "linear", False, 0),
                                                                         ((14, 9, 7), 5, "limitedLinear 1", False, 0), ((12, 9, 7), 3, "linear", True, 2)])
def test_scalarTransportFoam_matches_the_oracle_statement_for_statement(pkg, orc, tmp_path, dims, n_steps, scheme, corrected, n_non_orth):
    DT, delta_t = 0.01, 0.01
    case_dir = str(tmp_path / "channel")
    pts, faces, owner, neighbour, patches, tin, T0 = write_channel(case_dir, dims, DT, delta_t, n_steps, scheme, corrected, n_non_orth, smooth_start=scheme.startswith("limitedLinear"))
    out = subprocess.run([os.path.join(PKG, "scalarTransportFoam"), case_dir], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr + out.stdout[-1500:]
    got = [(m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))) for m in map(LINE.match, out.stdout.splitlines()) if m]
    ref, Tref = oracle_scalar_transport(pkg, orc, pts, faces, owner, neighbour, patches, tin, T0, DT, delta_t, n_steps, scheme, corrected, n_non_orth)
    assert len(got) == len(ref) == n_steps * (n_non_orth + 1)
    for g, r in zip(got, ref):
        assert g[:2] == r[:2] and g[4] == r[4], (g, r)
        assert abs(g[2] - r[2]) <= 1e-7 * max(r[2], 1e-12) + 1e-14 and abs(g[3] - r[3]) <= 1e-6 * max(r[2], 1e-12) + 1e-14, (g, r)
    f = read_vol_field(os.path.join(case_dir, f"{n_steps * delta_t:.10g}", "T"))
    assert f["header"]["class"] == "volScalarField" and np.max(np.abs(f["internalField"] - Tref)) <= 1e-8 * np.max(np.abs(Tref))
    assert 0.05 < np.max(Tref) < 2.0 and np.min(Tref) > -0.2                          # the inflow profile has entered the channel
    bf = dict(f["boundaryField"])
    assert bf["inlet"]["type"] == "fixedValue" and np.array_equal(bf["inlet"]["value"], tin) and bf["outlet"] == {"type": "zeroGradient"}
    if scheme.startswith("limitedLinear"):                                            # the limiter limits: neither the linear nor the upwind result
        for other in ("linear", "upwind"):
            _, To = oracle_scalar_transport(pkg, orc, pts, faces, owner, neighbour, patches, tin, T0, DT, delta_t, n_steps, other, corrected, n_non_orth)
            assert np.max(np.abs(To - Tref)) > 1e-5 * np.max(np.abs(Tref))
    if corrected:                                                                     # ... and the correction is not a no-op on this mesh
        _, Tun = oracle_scalar_transport(pkg, orc, pts, faces, owner, neighbour, patches, tin, T0, DT, delta_t, n_steps, scheme, False, n_non_orth)
        assert np.max(np.abs(Tun - Tref)) > 1e-7 * np.max(np.abs(Tref))
