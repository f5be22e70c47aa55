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
