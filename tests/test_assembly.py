"""fvMatrix assembly sweeps: oracle consistency (CPU) and engine parity, bit-exact (gpu)."""
import numpy as np
import pytest


def box_patches(dims):
    """faceCells of the six boundary patches of a lexicographic box (x-min, x-max, y-min, ...)."""
    nx, ny, nz = dims
    c = np.arange(nx * ny * nz)
    i, j, k = c % nx, (c // nx) % ny, c // (nx * ny)
    return [c[i == 0], c[i == nx - 1], c[j == 0], c[j == ny - 1], c[k == 0], c[k == nz - 1]]


def test_laplacian_plus_boundary_diag_reproduces_the_synthetic_matrix(pkg, orc):
    # fvm::laplacian(gamma=1) on uniform cubes: upper = deltaCoeffs*|Sf| = (1/h)*h^2 = h; diag = -sum;
    # fixedValue x-min patch: internalCoeffs = -gamma*|Sf|*2/h = -2h added by addBoundaryDiag
    dims = (9, 7, 5)
    case = pkg.synthetic.box_case(*dims, vary=0.0)
    h = 1.0 / dims[0]
    nf = case.n_faces
    upper, diag = orc.fvm_laplacian(case.n_cells, case.lower_addr, case.upper_addr, np.full(nf, 1.0 / h), np.full(nf, h * h))
    patch = box_patches(dims)[0]
    diag = orc.patch_add(patch, np.full(patch.shape[0], -2.0 * h), diag, 0)
    assert np.allclose(upper, case.upper, rtol=1e-15, atol=0)
    assert np.allclose(diag, case.diag, rtol=1e-14, atol=0)


def test_div_is_conservative_and_upwind(pkg, orc):
    dims = (8, 6, 5)
    case = pkg.synthetic.box_case(*dims)
    phi = pkg.synthetic.splitmix_uniform(3, case.n_faces) - 0.3
    w = (phi >= 0).astype(float)  # upwind weights
    lower, upper, diag = orc.fvm_div(case.n_cells, case.lower_addr, case.upper_addr, w, phi)
    assert np.all(lower <= 0) and np.all(upper <= 0)
    # column sums vanish on interior: diag = -sum(lower of own faces) - sum(upper of neighbour faces)
    ref = np.zeros(case.n_cells)
    np.subtract.at(ref, case.lower_addr, lower); np.subtract.at(ref, case.upper_addr, upper)
    assert np.max(np.abs(ref - diag)) < 1e-15
    # divergence of the flux through surfaceIntegrate == A*1 of the convection matrix (sign convention)
    div = orc.surface_integrate(case.n_cells, case.lower_addr, case.upper_addr, phi)
    row = diag.copy(); np.add.at(row, case.lower_addr, upper); np.add.at(row, case.upper_addr, lower)
    assert np.max(np.abs(row - div)) < 1e-14


def test_relax_makes_the_matrix_dominant_and_keeps_the_fixed_point(pkg, orc):
    dims = (7, 6, 5)
    case = pkg.synthetic.box_case(*dims, symmetric=False)
    patches = box_patches(dims)[:2]
    ic = [np.full(patches[0].shape[0], 0.02), np.full(patches[1].shape[0], -0.01)]
    bc = [np.full(patches[0].shape[0], 0.03), np.full(patches[1].shape[0], 0.04)]
    psi = pkg.synthetic.splitmix_uniform(5, case.n_cells)
    d, s = orc.relax(case.n_cells, case.lower_addr, case.upper_addr, 0.7, case.diag, case.lower, case.upper, case.source, psi,
                     patches, ic, bc, [0, 1])
    sum_off = orc.row_face_op(2, case.n_cells, case.lower_addr, case.upper_addr, case.lower, case.upper, np.zeros(case.n_cells))
    assert np.all(d * 0.7 >= sum_off - 0.05)                      # dominance before the 1/alpha scaling (patch terms aside)
    assert np.allclose(s - case.source, (d - case.diag) * psi, rtol=1e-12, atol=1e-300)   # S += (D - D0) psi


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(13, 11, 9), (40, 3, 2), (2, 2, 2)])
def test_engine_assembly_bit_exact(pkg, orc, dims):
    import torch
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    case = syn.box_case(*dims, symmetric=False)
    n, nf, lo, up = case.n_cells, case.n_faces, case.lower_addr, case.upper_addr
    addr = eng.Addressing(ctx, n, lo, up)
    asm = eng.Assembly(addr)
    E = lambda m: torch.empty(m, dtype=torch.float64, device="cuda:0")
    # row face ops
    for kind in (0, 1, 2):
        start = syn.splitmix_uniform(kind, n)
        io = dev(start)
        asm.row_face_op(kind, dev(case.lower), dev(case.upper), io)
        assert np.array_equal(host(io), orc.row_face_op(kind, n, lo, up, case.lower, case.upper, start))
        io = dev(start)
        asm.row_face_op(kind, None, dev(case.upper), io)   # symmetric: lower aliases upper
        assert np.array_equal(host(io), orc.row_face_op(kind, n, lo, up, None, case.upper, start))
    # laplacian
    delta, gam = 1.0 + syn.splitmix_uniform(7, nf), 0.5 + syn.splitmix_uniform(8, nf)
    uo, do = E(nf), E(n)
    asm.fvm_laplacian(dev(delta), dev(gam), uo, do)
    ru, rd = orc.fvm_laplacian(n, lo, up, delta, gam)
    assert np.array_equal(host(uo), ru) and np.array_equal(host(do), rd)
    # div
    w, phi = syn.splitmix_uniform(9, nf), syn.splitmix_uniform(10, nf) - 0.5
    lo_o, uo, do = E(nf), E(nf), E(n)
    asm.fvm_div(dev(w), dev(phi), lo_o, uo, do)
    rl, ru, rd = orc.fvm_div(n, lo, up, w, phi)
    assert np.array_equal(host(lo_o), rl) and np.array_equal(host(uo), ru) and np.array_equal(host(do), rd)
    # surfaceIntegrate (with and without volumes), face interpolation
    vol = 0.5 + syn.splitmix_uniform(11, n)
    out = E(n)
    asm.surface_integrate(dev(phi), None, out); assert np.array_equal(host(out), orc.surface_integrate(n, lo, up, phi))
    asm.surface_integrate(dev(phi), dev(vol), out); assert np.array_equal(host(out), orc.surface_integrate(n, lo, up, phi, vol))
    cellf = syn.splitmix_uniform(12, n)
    sf = E(nf)
    asm.face_interpolate(dev(w), dev(cellf), sf); assert np.array_equal(host(sf), orc.face_interpolate(lo, up, w, cellf))
    # patches: addBoundaryDiag / addBoundarySource / relax
    patches = box_patches(dims)
    ph = [eng.Patch(ctx, n, p) for p in patches]
    diag = dev(case.diag)
    ref = case.diag.copy()
    for k, (p, h) in enumerate(zip(patches, ph)):
        pf = syn.splitmix_uniform(20 + k, p.shape[0]) - 0.5
        h.add(dev(pf), diag, k % 3)
        ref = orc.patch_add(p, pf, ref, k % 3)
    assert np.array_equal(host(diag), ref)
    ic = [syn.splitmix_uniform(30 + k, p.shape[0]) - 0.5 for k, p in enumerate(patches)]
    bc = [syn.splitmix_uniform(40 + k, p.shape[0]) - 0.5 for k, p in enumerate(patches)]
    coupled = [0, 1, 0, 1, 0, 0]
    psi = syn.splitmix_uniform(50, n)
    d, s = dev(case.diag), dev(case.source)
    asm.relax(0.7, d, dev(case.lower), dev(case.upper), s, dev(psi), ph, [dev(a) for a in ic], [dev(a) for a in bc], coupled)
    rd, rs = orc.relax(n, lo, up, 0.7, case.diag, case.lower, case.upper, case.source, psi, patches, ic, bc, coupled)
    assert np.array_equal(host(d), rd) and np.array_equal(host(s), rs)
    # coupled part of addBoundarySource (fvMatrix::H): source[faceCells] += boundaryCoeffs * patchNeighbourField
    srcd, sref = dev(case.source), case.source.copy()
    for k, (p, h) in enumerate(zip(patches, ph)):
        nbr = syn.splitmix_uniform(70 + k, p.shape[0])
        h.add_product(dev(bc[k]), dev(nbr), srcd, k % 2)
        sref = orc.patch_add_product(p, bc[k], nbr, sref, k % 2)
    assert np.array_equal(host(srcd), sref)
    # boundary part of fvMatrix::flux: coupled patches multiply boundaryCoeffs by the neighbour field, the others do not
    for k, (p, h) in enumerate(zip(patches, ph)):
        nbr = syn.splitmix_uniform(60 + k, p.shape[0]) if coupled[k] else None
        out = E(p.shape[0])
        h.flux(dev(ic[k]), dev(bc[k]), dev(psi), out, None if nbr is None else dev(nbr))
        assert np.array_equal(host(out), orc.patch_flux(p, ic[k], bc[k], psi, nbr))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fixed256", "fixed1024", "tiles", "tiles_unstaged", "fixed_unstaged"])
@pytest.mark.parametrize("name", ["box", "graph"])
def test_row_passes_bit_exact_for_every_block_shape(pkg, orc, monkeypatch, name, mode):
    """The row passes (row face ops, fvm::laplacian, fvm::div, surfaceIntegrate, Gauss grad) own blocks of consecutive cells:
    fixed ranges of 256 / 1024 cells in the caller's numbering, the layout's tiles under ordered addressing; neighbour-side
    faces inside the block come from LDS, cut faces are gathered / recomputed.  Same bits as the oracle in every shape,
    and on the unstaged path a block takes when its faces do not fit the LDS arrays."""
    import torch
    from conftest import random_graph_case
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    case = syn.box_case(31, 23, 19, symmetric=False) if name == "box" else random_graph_case(pkg, 9000, extra=3.0, seed=5, symmetric=False)
    if mode.startswith("tiles"):
        a0 = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        case = syn.renumber(case, a0.cell_perm())
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, ordered=True, tile_cell_start=a0.tile_starts())
        assert addr.is_ordered and addr.n_tiles > 4
    else:
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        assert not addr.is_ordered
    for var in ("MI_ROW_BS", "MI_GRAD_BS"):
        monkeypatch.setenv(var, "1024" if mode == "fixed1024" else "256")
    if mode.endswith("unstaged"):
        monkeypatch.setenv("MI_ROW_CAP", "64")
    n, nf, lo, up = case.n_cells, case.n_faces, case.lower_addr, case.upper_addr
    asm = eng.Assembly(addr)
    E = lambda m: torch.empty(m, dtype=torch.float64, device="cuda:0")
    for kind in (0, 1, 2):
        start = syn.splitmix_uniform(kind, n)
        io = dev(start)
        asm.row_face_op(kind, dev(case.lower), dev(case.upper), io)
        assert np.array_equal(host(io), orc.row_face_op(kind, n, lo, up, case.lower, case.upper, start)), kind
        io = dev(start)
        asm.row_face_op(kind, None, dev(case.upper), io)
        assert np.array_equal(host(io), orc.row_face_op(kind, n, lo, up, None, case.upper, start)), kind
    delta, gam = 1.0 + syn.splitmix_uniform(7, nf), 0.5 + syn.splitmix_uniform(8, nf)
    uo, do = E(nf), E(n)
    asm.fvm_laplacian(dev(delta), dev(gam), uo, do)
    ru, rd = orc.fvm_laplacian(n, lo, up, delta, gam)
    assert np.array_equal(host(uo), ru) and np.array_equal(host(do), rd)
    w, phi = syn.splitmix_uniform(9, nf), syn.splitmix_uniform(10, nf) - 0.5
    lo_o, uo, do = E(nf), E(nf), E(n)
    asm.fvm_div(dev(w), dev(phi), lo_o, uo, do)
    rl, ru, rd = orc.fvm_div(n, lo, up, w, phi)
    assert np.array_equal(host(lo_o), rl) and np.array_equal(host(uo), ru) and np.array_equal(host(do), rd)
    vol = 0.5 + syn.splitmix_uniform(11, n)
    out = E(n)
    asm.surface_integrate(dev(phi), None, out); assert np.array_equal(host(out), orc.surface_integrate(n, lo, up, phi))
    asm.surface_integrate(dev(phi), dev(vol), out); assert np.array_equal(host(out), orc.surface_integrate(n, lo, up, phi, vol))
    Sf = [syn.splitmix_uniform(20 + k, nf) - 0.5 for k in range(3)]
    g = [E(n) for _ in range(3)]
    asm.gauss_grad([dev(x) for x in Sf], dev(phi), dev(vol), g)
    for a, b in zip(g, orc.gauss_grad(n, lo, up, Sf, phi, vol)):
        assert np.array_equal(host(a), b)
    # round 5: the flux of an interpolated (scaled) cell vector and its surfaceIntegrate in one row pass (pEqn.H:49-71)
    lam = syn.splitmix_uniform(30, nf)
    V = [syn.splitmix_uniform(31 + k, n) - 0.5 for k in range(3)]
    rho, aA, aB = 0.8 + syn.splitmix_uniform(35, n), syn.splitmix_uniform(36, nf) - 0.5, syn.splitmix_uniform(37, nf)
    po, dv = E(nf), E(n)
    for kw in (dict(), dict(scale=rho), dict(scale=rho, add_a=aA, add_b=aB, vol=vol), dict(add_a=aA)):
        asm.flux_div(dev(lam), [dev(x) for x in Sf], [dev(x) for x in V], po, dv, cell_scale=dev(kw["scale"]) if "scale" in kw else None,
                     add_a=dev(kw["add_a"]) if "add_a" in kw else None, add_b=dev(kw["add_b"]) if "add_b" in kw else None,
                     vol=dev(kw["vol"]) if "vol" in kw else None)
        rp, rd = orc.flux_div(n, lo, up, lam, Sf, V, **kw)
        assert np.array_equal(host(po), rp) and np.array_equal(host(dv), rd), sorted(kw)
    # the fused passes recompute cut faces from their inputs: an output aliasing an input is refused
    d = dev(delta)
    with pytest.raises(eng.MiError):
        asm.fvm_laplacian(d, dev(gam), d, do)
    l = dev(lam)
    with pytest.raises(eng.MiError):
        asm.flux_div(l, [dev(x) for x in Sf], [dev(x) for x in V], l, dv)


def test_compressible_operators_oracle_against_plain_numpy(pkg, orc):
    """fvm::ddt(rho, vf), fvm::Su / Sp / SuSp, fvc::ddtCorr(rho, U, phi) and the interpolated flux + its surfaceIntegrate
    (EulerDdtScheme.C:403-440, 663-720; fvmSup.C:34-214; ddtScheme.C:139-174; pEqn.H:49-71): the C restatements against the
    formulas written out with numpy field operations -- one rounding per operation, so bit for bit (the fused multiply-adds of
    the interpolate / dot product are evaluated in exact rational arithmetic and rounded once)"""
    from fractions import Fraction

    def fma(a, b, c):       # one rounding: exact rational arithmetic, then float() rounds to nearest even
        return float(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))
    syn = pkg.synthetic
    case = syn.box_case(7, 6, 5)
    n, nf, lo, up = case.n_cells, case.n_faces, case.lower_addr, case.upper_addr
    u = lambda seed, m: syn.splitmix_uniform(seed, m)
    rho, rho0, vol, psi0 = 0.9 + u(1, n), 0.8 + u(2, n), 0.5 + u(3, n), u(4, n) - 0.5
    rdt = 1.0 / 3e-4
    d, s = orc.fvm_ddt_euler_rho(rdt, rho, rho0, vol, psi0)
    assert np.array_equal(d, (rdt * rho) * vol) and np.array_equal(s, ((rdt * rho0) * psi0) * vol)
    d1, s1 = orc.fvm_ddt_euler(rdt, 1.0, vol, psi0)
    d2, s2 = orc.fvm_ddt_euler_rho(rdt, np.ones(n), np.ones(n), vol, psi0)
    assert np.array_equal(d1, d2) and np.array_equal(s1, s2)                  # rho = 1: the plain fvm::ddt(vf)
    su, sp, susp, vf = u(5, n) - 0.5, u(6, n), u(7, n) - 0.5, u(8, n) - 0.5
    assert np.array_equal(orc.fvm_su(vol, su, s), s - vol * su)
    assert np.array_equal(orc.fvm_sp(vol, sp, d), d + vol * sp) and np.array_equal(orc.fvm_sp(vol, 0.25, d), d + vol * 0.25)
    dd, ss = orc.fvm_susp(vol, susp, vf, d, s)
    assert np.array_equal(dd, d + vol * np.maximum(susp, 0.0)) and np.array_equal(ss, s - (vol * np.minimum(susp, 0.0)) * vf)
    lam, Sf, U = u(9, nf), [u(10 + k, nf) - 0.5 for k in range(3)], [u(13 + k, n) - 0.5 for k in range(3)]
    aA, aB = u(16, nf) - 0.5, u(17, nf)

    def flux(scale):
        V = [scale * c for c in U] if scale is not None else U
        out = np.empty(nf)
        for f in range(nf):
            P, N = lo[f], up[f]
            i = [fma(lam[f], V[k][P] - V[k][N], V[k][N]) for k in range(3)]
            out[f] = fma(i[2], Sf[2][f], fma(i[0], Sf[0][f], i[1] * Sf[1][f]))   # Vector operator& as the compiled reference rounds it
        return out
    phi, div = orc.flux_div(n, lo, up, lam, Sf, U, scale=rho0, add_a=aA, add_b=aB, vol=vol)
    ref = flux(rho0) + aA * aB
    assert np.array_equal(phi, ref) and np.array_equal(div, orc.surface_integrate(n, lo, up, ref, vol))
    phi0 = u(18, nf) - 0.5
    got = orc.ddt_phi_corr(lo, up, rdt, lam, Sf, U, rho0, phi0)
    corr = phi0 - flux(rho0)
    coeff = 1.0 - np.minimum(np.abs(corr) / (np.abs(phi0) + 1e-15), 1.0)
    assert np.array_equal(got, (coeff * rdt) * corr)
    assert np.all((coeff >= 0) & (coeff <= 1)) and np.any(coeff == 0) and np.any(coeff > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(13, 11, 9), (3, 2, 2)])
def test_engine_compressible_operators_bit_exact(pkg, orc, dims):
    """rhoPimpleFoam's operators (BASELINE config 5; VERDICT r04 "missing" 3): mi_fvm_ddt_euler_rho, mi_fvm_su / sp / susp,
    mi_ddt_phi_corr, mi_flux_div against the oracle, bit for bit"""
    import torch
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    case = syn.box_case(*dims)
    n, nf, lo, up = case.n_cells, case.n_faces, case.lower_addr, case.upper_addr
    asm = eng.Assembly(eng.Addressing(ctx, n, lo, up))
    E = lambda m: torch.empty(m, dtype=torch.float64, device="cuda:0")
    u = lambda seed, m: syn.splitmix_uniform(seed, m)
    rho, rho0, vol, psi0 = 0.9 + u(1, n), 0.8 + u(2, n), 0.5 + u(3, n), u(4, n) - 0.5
    rdt = 1.0 / 3e-4
    d, s = E(n), E(n)
    asm.fvm_ddt_euler_rho(rdt, dev(rho), dev(rho0), dev(vol), dev(psi0), d, s)
    rd, rs = orc.fvm_ddt_euler_rho(rdt, rho, rho0, vol, psi0)
    assert np.array_equal(host(d), rd) and np.array_equal(host(s), rs)
    su, sp, susp, vf = u(5, n) - 0.5, u(6, n), u(7, n) - 0.5, u(8, n) - 0.5
    asm.fvm_su(dev(vol), dev(su), s); rs = orc.fvm_su(vol, su, rs); assert np.array_equal(host(s), rs)
    asm.fvm_sp(dev(vol), dev(sp), d); rd = orc.fvm_sp(vol, sp, rd); assert np.array_equal(host(d), rd)
    asm.fvm_sp(dev(vol), 0.25, d); rd = orc.fvm_sp(vol, 0.25, rd); assert np.array_equal(host(d), rd)
    asm.fvm_susp(dev(vol), dev(susp), dev(vf), d, s); rd, rs = orc.fvm_susp(vol, susp, vf, rd, rs)
    assert np.array_equal(host(d), rd) and np.array_equal(host(s), rs)
    lam, Sf, U = u(9, nf), [u(10 + k, nf) - 0.5 for k in range(3)], [u(13 + k, n) - 0.5 for k in range(3)]
    phi0, out = u(18, nf) - 0.5, E(nf)
    for r0 in (rho0, None):
        asm.ddt_phi_corr(rdt, dev(lam), [dev(x) for x in Sf], [dev(x) for x in U], dev(r0) if r0 is not None else None, dev(phi0), out)
        assert np.array_equal(host(out), orc.ddt_phi_corr(lo, up, rdt, lam, Sf, U, r0, phi0))
    ddt = orc.ddt_phi_corr(lo, up, rdt, lam, Sf, U, rho0, phi0)
    rAUf = u(19, nf)
    po, dv = E(nf), E(n)
    asm.flux_div(dev(lam), [dev(x) for x in Sf], [dev(x) for x in U], po, dv, cell_scale=dev(rho), add_a=dev(rAUf), add_b=dev(ddt))
    rp, rdv = orc.flux_div(n, lo, up, lam, Sf, U, scale=rho, add_a=rAUf, add_b=ddt)
    assert np.array_equal(host(po), rp) and np.array_equal(host(dv), rdv)


@pytest.mark.gpu
@pytest.mark.parametrize("transonic", [False, True])
def test_rhopimple_step_and_its_pressure_solve_against_the_oracle(pkg, orc, transonic):
    """tools/workloads.py: rhopimple_supplement (BASELINE config 5's solver: UEqn.H / EEqn.H / pEqn.H of rhoPimpleFoam through the
    C ABI) on a small box.  The pressure system it assembled and bound -- symmetric for the non-transonic pEqn, ASYMMETRIC for the
    transonic one (fvm::ddt(psi,p) + fvm::div(phid,p) - fvm::laplacian(rhorAUf,p), pEqn.H:36-46) -- is solved again by the oracle's
    GAMG from the same start field: same cycle count, histories within 1e-10, same solution (GAMGSolverSolve.C:59-160)."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import workloads
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    case = syn.box_case(24, 20, 16)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    cap = {}
    out = workloads.rhopimple_supplement(eng, syn, case, addr, ctx, torch.device("cuda:0"), steps=1, transonic=transonic, capture=cap)
    assert out["ms_per_time_step"] > 0 and len(out["stages_ms"]) == 7 and out["gamg_cycles"] >= 1 and len(out["pbicg_iterations_per_component"]) == 3
    assert out["pressure_matrix"] == ("asymmetric" if transonic else "symmetric")
    assert (cap["lower"] is not None) == transonic
    if transonic:
        assert np.max(np.abs(cap["lower"] - cap["upper"])) > 1e-6 * np.max(np.abs(cap["upper"]))        # really asymmetric
    pcase = syn.LduCase(case.n_cells, case.lower_addr, case.upper_addr, cap["diag"], cap["upper"], cap["lower"], cap["source"])
    pcase.dims = case.dims
    H = orc.GamgHierarchy(pcase, workloads.box_pair_weights(case), 100)
    ref_psi, ref = H.solve(cap["start"], cap["source"], tolerance=1e-12, relTol=0.05, maxIter=50)
    perf = cap["perf"]
    assert perf["nIterations"] == ref["nIterations"] and perf["converged"] == ref["converged"]
    h, hr = perf["history"], ref["history"]
    assert h.shape == hr.shape and np.max(np.abs(h - hr)) < 1e-10 * hr[0]
    assert np.max(np.abs(cap["solution"] - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


@pytest.mark.gpu
def test_assemble_then_solve_matches_oracle_end_to_end(pkg, orc):
    """config-5 style step on a small box: fused fvm::laplacian + boundary diag on the GPU feeds PCG."""
    import torch
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    dims = (16, 14, 12)
    case = syn.box_case(*dims)   # supplies addressing and source
    n, nf = case.n_cells, case.n_faces
    h = 1.0 / dims[0]
    delta = np.full(nf, 1.0 / h)
    gam = h * h * (1.0 + 0.1 * syn.splitmix_uniform(12345, nf))
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
    asm = eng.Assembly(addr)
    upper = torch.empty(nf, dtype=torch.float64, device="cuda:0"); diag = torch.empty(n, dtype=torch.float64, device="cuda:0")
    asm.fvm_laplacian(dev(delta), dev(gam), upper, diag)
    patch = box_patches(dims)[0]
    eng.Patch(ctx, n, patch).add(dev(np.full(patch.shape[0], -2.0 * h)), diag, 0)
    mat = eng.Matrix(addr); mat.set_coeffs(diag, upper, None)
    psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    perf = mat.pcg(psi, dev(case.source), "diagonal", tolerance=1e-8, maxIter=500)
    ru, rd = orc.fvm_laplacian(n, case.lower_addr, case.upper_addr, delta, gam)
    rd = orc.patch_add(patch, np.full(patch.shape[0], -2.0 * h), rd, 0)
    ref_case = syn.LduCase(n, case.lower_addr, case.upper_addr, rd, ru, None, case.source)
    _, ref = orc.System([ref_case]).pcg(np.zeros(n), case.source, "diagonal", tolerance=1e-8, maxIter=500)
    assert perf["nIterations"] == ref["nIterations"]
    assert np.max(np.abs(perf["history"] - ref["history"])) < 1e-10


# ---- scheme front-end: ddt, upwind / limitedLinear weights, Gauss gradient ---------------------------------------
def box_geometry(dims):
    """cell centres, internal-face area vectors Sf and linear weights of the uniform lexicographic box"""
    nx, ny, nz = dims
    h = 1.0 / nx
    c = np.arange(nx * ny * nz)
    C = [(c % nx + 0.5) * h, ((c // nx) % ny + 0.5) * h, (c // (nx * ny) + 0.5) * h]
    return h, C


def face_dirs(case, dims):
    d = case.upper_addr.astype(np.int64) - case.lower_addr
    return np.where(d == 1, 0, np.where(d == dims[0], 1, 2))


def test_gauss_grad_of_a_linear_field_is_exact_inside(pkg, orc):
    dims = (9, 8, 7)
    case = pkg.synthetic.box_case(*dims)
    h, C = box_geometry(dims)
    phi = 2.0 * C[0] - 3.0 * C[1] + 0.5 * C[2]
    direction = face_dirs(case, dims)
    Sf = [np.where(direction == k, h * h, 0.0) for k in range(3)]
    ssf = orc.face_interpolate(case.lower_addr, case.upper_addr, np.full(case.n_faces, 0.5), phi)
    g = orc.gauss_grad(case.n_cells, case.lower_addr, case.upper_addr, Sf, ssf, np.full(case.n_cells, h ** 3))
    nx, ny, nz = dims
    c = np.arange(case.n_cells); i, j, k = c % nx, (c // nx) % ny, c // (nx * ny)
    inner = (i > 0) & (i < nx - 1) & (j > 0) & (j < ny - 1) & (k > 0) & (k < nz - 1)
    for comp, exact in zip(g, (2.0, -3.0, 0.5)):
        assert np.max(np.abs(comp[inner] - exact)) < 1e-11


def test_limited_linear_reduces_to_its_limits(pkg, orc):
    dims = (10, 6, 5)
    case = pkg.synthetic.box_case(*dims)
    h, C = box_geometry(dims)
    syn = pkg.synthetic
    flux = syn.splitmix_uniform(5, case.n_faces) - 0.4
    cdw = np.full(case.n_faces, 0.5)
    # smooth linear field with its exact gradient: r = 1 -> limiter 1 -> central weights
    phi = 1.0 + C[0] + 2 * C[1]
    g = [np.full(case.n_cells, 1.0), np.full(case.n_cells, 2.0), np.zeros(case.n_cells)]
    w, lim = orc.limited_linear_weights(case.lower_addr, case.upper_addr, 1.0, cdw, flux, phi, g, C)
    assert np.all(lim == 1.0) and np.all(w == 0.5)
    # local extremum (gradient opposes the face difference): r < 0 -> limiter 0 -> upwind
    g2 = [-x for x in g]
    w2, lim2 = orc.limited_linear_weights(case.lower_addr, case.upper_addr, 1.0, cdw, flux, phi, g2, C)
    xy = face_dirs(case, dims) != 2      # z faces: phi does not vary -> gradf = 0 -> the 1000-branch with sign(0) = +1 -> limiter 1
    assert np.all(lim2[xy] == 0.0) and np.array_equal(w2[xy], orc.upwind_weights(flux)[xy]) and np.all(lim2[~xy] == 1.0)
    # uniform field: gradf = 0 -> the 1000-branch, sign(0) = +1
    w3, lim3 = orc.limited_linear_weights(case.lower_addr, case.upper_addr, 1.0, cdw, flux, np.ones(case.n_cells), g, C)
    assert np.all((lim3 == 0.0) | (lim3 == 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(13, 11, 9), (3, 2, 2)])
def test_engine_scheme_front_end_bit_exact(pkg, orc, dims):
    import torch
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    host = lambda t: t.cpu().numpy()
    case = syn.box_case(*dims)
    n, nf = case.n_cells, case.n_faces
    addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
    A = eng.Assembly(addr)
    h, C = box_geometry(dims)
    E = lambda m: torch.empty(m, dtype=torch.float64, device="cuda:0")
    # ddt
    vol = h ** 3 * (1.0 + 0.2 * syn.splitmix_uniform(1, n)); psi0 = syn.splitmix_uniform(2, n) - 0.5
    d, s = E(n), E(n)
    A.fvm_ddt_euler(1.0 / 2.5e-3, 1.2, dev(vol), dev(psi0), d, s)
    rd, rs = orc.fvm_ddt_euler(1.0 / 2.5e-3, 1.2, vol, psi0)
    assert np.array_equal(host(d), rd) and np.array_equal(host(s), rs)
    # upwind / limitedLinear
    flux = syn.splitmix_uniform(5, nf) - 0.4
    w = E(nf); A.upwind_weights(dev(flux), w)
    assert np.array_equal(host(w), orc.upwind_weights(flux))
    phi = np.sin(3 * C[0]) * np.cos(2 * C[1]) + C[2] ** 2 + 0.05 * syn.splitmix_uniform(6, n)
    direction = face_dirs(case, dims)
    Sf = [np.where(direction == k, h * h * (1 + 0.1 * syn.splitmix_uniform(10 + k, nf)), 0.01 * h * h * (syn.splitmix_uniform(20 + k, nf) - 0.5)) for k in range(3)]
    cdw = 0.4 + 0.2 * syn.splitmix_uniform(7, nf)
    ssf = orc.face_interpolate(case.lower_addr, case.upper_addr, cdw, phi)
    g = [E(n) for _ in range(3)]
    A.gauss_grad([dev(x) for x in Sf], dev(ssf), dev(vol), g)
    rg = orc.gauss_grad(n, case.lower_addr, case.upper_addr, Sf, ssf, vol)
    for a, b in zip(g, rg):
        assert np.array_equal(host(a), b)
    gn = [E(n) for _ in range(3)]
    A.gauss_grad([dev(x) for x in Sf], dev(ssf), None, gn)
    for a, b in zip(gn, orc.gauss_grad(n, case.lower_addr, case.upper_addr, Sf, ssf, None)):
        assert np.array_equal(host(a), b)
    for k in (1.0, 0.33, 0.0):
        wl, lim = E(nf), E(nf)
        A.limited_linear_weights(k, dev(cdw), dev(flux), dev(phi), [dev(x) for x in rg], [dev(x) for x in C], wl, lim)
        rw, rl = orc.limited_linear_weights(case.lower_addr, case.upper_addr, k, cdw, flux, phi, rg, C)
        assert np.array_equal(host(lim), rl) and np.array_equal(host(wl), rw)
    # operator+= / -= on coefficient arrays
    x, y = syn.splitmix_uniform(8, nf) - 0.5, syn.splitmix_uniform(9, nf) - 0.5
    out = E(nf); A.axpby(1.0, dev(x), -0.75, dev(y), out)
    assert np.array_equal(host(out), orc.axpby(1.0, x, -0.75, y))
    xd = dev(x); A.axpby(2.0, xd, 1.0, dev(y), xd)           # in place
    assert np.array_equal(host(xd), orc.axpby(2.0, x, 1.0, y))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["fixed256", "fixed1024", "tiles", "tiles_unstaged"])
@pytest.mark.parametrize("name", ["box", "graph"])
def test_fused_assembly_equals_the_unfused_sequence_bit_for_bit(pkg, orc, monkeypatch, name, mode):
    """mi_fvm_assemble -- [fvm::ddt] + [fvm::div] - [fvm::laplacian] [+- fvm::Sp] [+- explicit terms] in ONE row pass -- against (a) the
    oracle's unfused sequence (tests/assembly_full_size.py: oracle_assemble) and (b) the engine's own scheme-by-scheme calls combined
    with mi_vec_axpby the way fvMatrix::operator+ / - combine them (fvMatrix.C:1693-2030), for three systems (momentum-like with three
    right-hand sides, upwind convection, Sp and two explicit terms; pressure-like symmetric; convection with given weights), in every
    block shape of the row passes; plus fvMatrix<vector>::relax fed with the pass's sumMagOffDiag, setValues (reference and upstream
    semantics) and setReference."""
    import torch
    import assembly_full_size as afs
    from conftest import random_graph_case
    eng, syn = pkg.engine, pkg.synthetic
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to("cuda:0")
    host = lambda t: (torch.cuda.synchronize(), t.cpu().numpy())[1]
    case = syn.box_case(31, 23, 19, symmetric=False) if name == "box" else random_graph_case(pkg, 9000, extra=3.0, seed=5, symmetric=False)
    if mode.startswith("tiles"):
        a0 = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
        case = syn.renumber(case, a0.cell_perm())
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, ordered=True, tile_cell_start=a0.tile_starts())
        assert addr.is_ordered
    else:
        addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    monkeypatch.setenv("MI_ROW_BS", "1024" if mode == "fixed1024" else "256")
    if mode.endswith("unstaged"):
        monkeypatch.setenv("MI_ROW_CAP", "64")
    M = dict(n=case.n_cells, nf=case.n_faces, lo=case.lower_addr, up=case.upper_addr, dims=(1, 1, 1))
    q = afs.inputs(pkg, M)
    got = afs.engine_run(pkg, ctx, addr, M, q)
    ref = afs.oracle_run(pkg, orc, M, q)
    assert sorted(got) == sorted(ref)
    for k in sorted(ref):
        assert np.array_equal(got[k], ref[k]), k
    # (b) the engine's own unfused sequence for the momentum-like system
    n, nf = M["n"], M["nf"]
    asm = eng.Assembly(addr)
    E = lambda m: torch.empty(m, dtype=torch.float64, device="cuda:0")
    wts, cl, cu, cd, lu, ld, dd, ds = E(nf), E(nf), E(nf), E(n), E(nf), E(n), E(n), [E(n) for _ in range(3)]
    flux, vol = dev(q["flux"]), dev(q["vol"])
    asm.upwind_weights(flux, wts); asm.fvm_div(wts, flux, cl, cu, cd); asm.fvm_laplacian(dev(q["delta"]), dev(q["gamma"]), lu, ld)
    for r in range(3):
        asm.fvm_ddt_euler_rho(q["rdt"], dev(q["rho"]), dev(q["rho0"]), vol, dev(q["V"][r]), dd, ds[r])
        asm.fvm_su(vol, dev(q["su"][r]), ds[r])                                   # + su: source -= V*su
        t = vol * dev(q["g"][r]); ds[r].add_(t)                                   # == g: source += V*g
    asm.axpby(1.0, cl, -1.0, lu, cl); asm.axpby(1.0, cu, -1.0, lu, cu)
    asm.axpby(1.0, dd, 1.0, cd, dd); asm.axpby(1.0, dd, -1.0, ld, dd)
    t = vol * dev(q["sp"]); dd.sub_(t)
    for key, t in (("lower", cl), ("upper", cu), ("diag", dd), ("source0", ds[0]), ("source1", ds[1]), ("source2", ds[2])):
        assert np.array_equal(host(t), got["assemble_momentum/" + key]), key
    # fvc::div(faceFlux, vf) (gaussConvectionScheme::fvcDiv): upwind and given weights, against the oracle's interpolate -> product -> surfaceIntegrate
    Kf, dv = E(nf), E(n)
    for wname in (None, "w"):
        asm.fvc_div(flux, None if wname is None else dev(q[wname]), dev(q["psi"]), vol, Kf, dv)
        wts_h = orc.upwind_weights(q["flux"]) if wname is None else q[wname]
        ref_f = q["flux"] * orc.face_interpolate(M["lo"], M["up"], wts_h, q["psi"])
        assert np.array_equal(host(Kf), ref_f) and np.array_equal(host(dv), orc.surface_integrate(n, M["lo"], M["up"], ref_f, q["vol"])), wname
    # upstream semantics of setValues against the oracle
    sv = orc.set_values(n, M["lo"], M["up"], q["set_cells"], q["set_vals"], q["psi"], q["Dc"], q["src"], q["Uc"], None, upstream=True)
    ps, sr, uo, lo_o = dev(q["psi"]), dev(q["src"]), E(nf), E(nf)
    asm.set_values(torch.from_numpy(q["set_cells"]).to("cuda:0"), dev(q["set_vals"]), ps, dev(q["Dc"]), sr, dev(q["Uc"]), None, uo, lo_o, upstream=True)
    assert np.array_equal(host(sr), sv["source"]) and np.array_equal(host(uo), sv["upper"]) and np.array_equal(host(lo_o), sv["lower"]) and np.array_equal(host(ps), sv["psi"])
    assert np.array_equal(sv["upper"], sv["lower"])          # upstream: a symmetric matrix stays symmetric
    with pytest.raises(eng.MiError):
        asm.assemble(flux, dd, lower_out=cl, sources_out=[], div=dict(flux=flux))       # a coefficient output aliasing an input


# ---- non-orthogonal correction of fvm::laplacian (gaussLaplacianSchemes.C:64-90) --------------------
def skewed_mesh(dims):
    """a distorted hex box (non-planar faces, non-orthogonal) in OpenFOAM's ordering with its geometric fields"""
    from test_polymesh import geometry, make_box_mesh
    pts, faces, owner, neighbour, patches = make_box_mesh(dims)
    G = geometry(pts, faces, owner, neighbour)
    nI = len(neighbour)
    lo, up = owner[:nI].astype(np.int32), neighbour.astype(np.int32)
    nhat = G["Sf"][:nI] / G["magSf"][:nI, None]
    d = G["C"][up] - G["C"][lo]
    corr = nhat - d * G["delta"][:, None]            # surfaceInterpolation.C:498-580: unitArea - delta*nonOrthDeltaCoeffs
    return dict(n=int(owner.max()) + 1, nI=nI, lo=lo, up=up, owner=owner, patches=patches, G=G, corr=[np.ascontiguousarray(corr[:, k]) for k in range(3)])


def nonorth_source_correction(orc, M, phi, boundary_value, gamma_magsf):
    """the reference's op sequence with the oracle's sweeps: grad = gaussGrad(linear interpolate(phi)) incl. boundary faces,
    flux = gammaMagSf*(corrVecs & interpolate(grad)), div = surfaceIntegrate(flux), returns V*div (what is subtracted from source)"""
    G, n, nI, lo, up = M["G"], M["n"], M["nI"], M["lo"], M["up"]
    Sf = [np.ascontiguousarray(G["Sf"][:nI, k]) for k in range(3)]
    ssf = orc.face_interpolate(lo, up, G["weights"], phi)
    g = orc.gauss_grad(n, lo, up, Sf, ssf, None)
    for name, ptype, cnt, start in M["patches"]:                      # boundary faces: Sf_p * (boundary value of phi)
        fc = M["owner"][start:start + cnt]
        bv = boundary_value(name, ptype, fc, start, cnt)
        for k in range(3):
            g[k] = orc.patch_add_product(fc, np.ascontiguousarray(G["Sf"][start:start + cnt, k]), bv, g[k], 0)
    g = [x / G["V"] for x in g]
    flux = orc.sngrad_correction_flux(lo, up, M["corr"], G["weights"], g, gamma_magsf)
    div = orc.surface_integrate(n, lo, up, flux, G["V"])
    return g, flux, G["V"] * div


def test_nonorth_correction_recovers_the_exact_face_gradient_of_a_linear_field(pkg, orc):
    # n = nonOrthDeltaCoeffs*d + k: for phi = a.x the uncorrected face gradient plus the correction is a.n exactly, so the corrected
    # Laplacian of a linear field has zero net flux in every interior cell -- what the explicit correction is for
    M = skewed_mesh((7, 6, 5))
    G, n, nI, lo, up = M["G"], M["n"], M["nI"], M["lo"], M["up"]
    a = np.array([0.7, -1.3, 0.45])
    phi = G["C"] @ a
    exact_bv = lambda name, ptype, fc, start, cnt: G["Cf"][start:start + cnt] @ a
    g, flux, vdiv = nonorth_source_correction(orc, M, phi, exact_bv, G["magSf"][:nI])
    interior = np.ones(n, bool); interior[M["owner"][nI:]] = False
    # Gauss gradient with linear interpolation is not exact on a skewed mesh (face centre != interpolation point): a few per cent
    assert np.max(np.abs(np.stack(g, 1)[interior] - a)) < 0.1 * np.abs(a).max()
    # with the EXACT gradient in every cell the identity is exact:
    gex = [np.full(n, a[k]) for k in range(3)]
    flux_ex = orc.sngrad_correction_flux(lo, up, M["corr"], G["weights"], gex, G["magSf"][:nI])
    unc = G["delta"] * G["magSf"][:nI] * (phi[up] - phi[lo])
    assert np.max(np.abs(unc + flux_ex - G["Sf"][:nI] @ a)) < 1e-13
    # and the correction is what separates the two on this mesh (it is not a no-op)
    assert np.max(np.abs(flux_ex)) > 1e-3 * np.max(np.abs(unc))
    # coupled-patch form with both sides equal reduces to the cell value: lambda*g + (1-lambda)*g
    fc = lo[:50]
    pf = orc.patch_sngrad_correction_flux(fc, [c[:50] for c in M["corr"]], G["weights"][:50], gex, [x[fc] for x in gex], G["magSf"][:50])
    assert np.max(np.abs(pf - G["magSf"][:50] * (np.stack(M["corr"], 1)[:50] @ a))) < 1e-14
    assert np.array_equal(orc.submul(G["V"], vdiv / G["V"], np.zeros(n)), -(G["V"] * (vdiv / G["V"])))


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(9, 8, 7), (3, 2, 2)])
def test_engine_nonorth_correction_bit_exact_and_corrected_solve(pkg, orc, dims):
    """fvm::laplacian with the explicit non-orthogonal correction on a distorted mesh: every sweep bit for bit against the
    oracle, then three non-orthogonal correctors (assemble with grad of the previous p, solve) against the same loop on the oracle"""
    import torch
    syn, eng = pkg.synthetic, pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    host = lambda t: t.cpu().numpy()
    E = lambda m: torch.empty(m, dtype=torch.float64, device="cuda:0")
    M = skewed_mesh(dims)
    G, n, nI, lo, up = M["G"], M["n"], M["nI"], M["lo"], M["up"]
    addr = eng.Addressing(ctx, n, lo, up)
    A = eng.Assembly(addr)
    patches = {name: (eng.Patch(ctx, n, M["owner"][start:start + cnt]), start, cnt, ptype) for name, ptype, cnt, start in M["patches"]}
    Sf = [np.ascontiguousarray(G["Sf"][:nI, k]) for k in range(3)]
    gms = G["magSf"][:nI] * (1.0 + 0.3 * syn.splitmix_uniform(4, nI))          # gamma*|Sf| with a varying gamma
    # boundary conditions of the test problem: inlet/outlet fixedValue 0, walls zeroGradient
    bv_host = lambda name, ptype, fc, start, cnt, phi: (np.zeros(cnt) if ptype == "patch" else phi[fc])

    def engine_correction(phi_d):
        ssf = E(nI); A.face_interpolate(dev(G["weights"]), phi_d, ssf)
        g = [E(n) for _ in range(3)]
        A.gauss_grad([dev(x) for x in Sf], ssf, None, g)
        for name, (P, start, cnt, ptype) in patches.items():
            bv = E(cnt)
            if ptype == "patch": bv.zero_()
            else: P.internal_field(phi_d, bv)
            for k in range(3):
                P.add_product(dev(G["Sf"][start:start + cnt, k]), bv, g[k], 0)
        V = dev(G["V"])
        gd = [E(n) for _ in range(3)]
        for k in range(3):
            eng._chk(eng.lib().mi_vec_div(ctx.h, n, eng._ptr(g[k]), eng._ptr(V), eng._ptr(gd[k])))
        flux = E(nI); A.sngrad_correction_flux([dev(c) for c in M["corr"]], dev(G["weights"]), gd, dev(gms), flux)
        div = E(n); A.surface_integrate(flux, V, div)
        return gd, flux, div

    phi = np.sin(3 * G["C"][:, 0]) * np.cos(2 * G["C"][:, 1]) + G["C"][:, 2] ** 2
    gd, flux, div = engine_correction(dev(phi))
    rg, rflux, rvdiv = nonorth_source_correction(orc, M, phi, lambda nm, pt, fc, s, c: bv_host(nm, pt, fc, s, c, phi), gms)
    for a, b in zip(gd, rg):
        assert np.array_equal(host(a), b)
    assert np.array_equal(host(flux), rflux)
    assert np.array_equal(host(div) * G["V"], rvdiv)
    src = syn.splitmix_uniform(9, n) - 0.5
    sd = dev(src.copy()); A.submul(dev(G["V"]), div, sd)
    assert np.array_equal(host(sd), orc.submul(G["V"], rvdiv / G["V"], src))
    # coupled-patch form (the first 64 internal faces posed as a patch whose neighbour values are the upper cells')
    m = min(64, nI)
    P = eng.Patch(ctx, n, lo[:m])
    nb = [x[up[:m]] for x in rg]
    pf = E(m); P.sngrad_correction_flux([dev(c[:m]) for c in M["corr"]], dev(G["weights"][:m]), [dev(x) for x in rg], [dev(x) for x in nb], dev(gms[:m]), pf)
    assert np.array_equal(host(pf), orc.patch_sngrad_correction_flux(lo[:m], [c[:m] for c in M["corr"]], G["weights"][:m], rg, nb, gms[:m]))

    # ---- nNonOrthogonalCorrectors = 3: laplacian(gamma, p) == S, corrected --------------------------------------------------
    up_c, diag_c = E(nI), E(n)
    A.fvm_laplacian(dev(G["delta"]), dev(gms), up_c, diag_c)
    ru, rd = orc.fvm_laplacian(n, lo, up, G["delta"], gms)
    ic = {}
    for name, (P, start, cnt, ptype) in patches.items():
        if ptype != "patch":
            continue
        ic[name] = -(G["magSf"][start:start + cnt] * G["delta_b"][start - nI:start - nI + cnt])    # fixedValue: internalCoeffs = -gamma|Sf|deltaCoeffs
        P.add(dev(ic[name]), diag_c, 0)
        rd = orc.patch_add(M["owner"][start:start + cnt], ic[name], rd, 0)
    assert np.array_equal(host(diag_c), rd) and np.array_equal(host(up_c), ru)
    mat = eng.Matrix(addr); mat.set_coeffs(diag_c, up_c, None)
    case = syn.LduCase(n, lo, up, rd, ru, None, np.zeros(n))
    S = orc.System([case])
    S0 = -(G["V"] * (1.0 + np.sin(5 * G["C"][:, 0])))
    p_d = torch.zeros(n, dtype=torch.float64, device="cuda:0"); p_h = np.zeros(n)
    for corr in range(3):
        _, _, div = engine_correction(p_d)
        src_d = dev(S0.copy()); A.submul(dev(G["V"]), div, src_d)
        _, _, rvdiv = nonorth_source_correction(orc, M, p_h, lambda nm, pt, fc, s, c: bv_host(nm, pt, fc, s, c, p_h), gms)
        src_h = orc.submul(G["V"], rvdiv / G["V"], S0)
        perf = mat.pcg(p_d, src_d, "DIC", tolerance=1e-10, maxIter=500)
        p_h, ref = S.pcg(p_h, src_h, "DIC", tolerance=1e-10, maxIter=500)
        assert perf["nIterations"] == ref["nIterations"] and np.max(np.abs(perf["history"] - ref["history"])) < 1e-10 * ref["history"][0]
        assert np.max(np.abs(host(p_d) - p_h)) < 1e-9 * np.max(np.abs(p_h))
    assert corr == 2 and np.max(np.abs(p_h)) > 0
