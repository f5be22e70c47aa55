"""The compiled C++ host mirror of the OpenFOAM interface (rapidcfd-dev_amd/foam): CPU checks that it
builds, links the C ABI and carries the reference's run-time names; the gpu test runs the miniature
solver application and checks every printed solverPerformance line against the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rapidcfd-dev_amd")


def test_mirror_builds_and_registers_reference_names(pkg):
    assert os.path.exists(os.path.join(PKG, "libmiFoam.so")) and os.path.exists(os.path.join(PKG, "pEqnFoam"))
    src = open(os.path.join(PKG, "foam", "miFoam.C")).read()
    for name in ("PCG", "PBiCG", "PBiCGStab", "smoothSolver", "GAMG", "ICCG", "BICCG"):
        assert re.search(r'MatrixConstructorToTable_\("%s"\)' % name, src), name
    syms = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, "libmiFoam.so")], capture_output=True, text=True).stdout
    assert "lduMatrix" in syms and "solver" in syms
    # the mirror computes nothing itself: it has no kernels, only calls into the C ABI
    assert "__global__" not in src and "mi_pcg_solve" in src


LINE = re.compile(r"^(\w+):  Solving for (\w+), Initial residual = (\S+), Final residual = (\S+), No Iterations (\d+)")


@pytest.mark.gpu
def test_pEqnFoam_matches_oracle(pkg, orc):
    dims = (12, 10, 8)
    out = subprocess.run([os.path.join(PKG, "pEqnFoam"), *map(str, dims)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    lines = [LINE.match(l) for l in out.stdout.splitlines()]
    got = [(m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))) for m in lines if m]
    syn = pkg.synthetic
    case = syn.box_case(*dims)           # addressing + source as in the application
    n, nf, h = case.n_cells, case.n_faces, 1.0 / dims[0]
    gam = h * h * (1.0 + 0.1 * syn.splitmix_uniform(12345, nf))
    upper, diag = orc.fvm_laplacian(n, case.lower_addr, case.upper_addr, np.full(nf, 1.0 / h), gam)
    xmin = np.nonzero(np.arange(n) % dims[0] == 0)[0]
    pdiag = orc.patch_add(xmin, np.full(xmin.shape[0], -2.0 * h), diag, 0)
    src = case.source
    P = orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, pdiag, upper, None, src)])
    z = np.zeros(n)
    exp = []
    for pre, name in (("diagonal", "diagonalPCG"), ("AINV", "AINVPCG"), ("none", "nonePCG")):
        _, p = P.pcg(z, src, pre, tolerance=1e-8); exp.append((name, "p", p))
    H = orc.GamgHierarchy(syn.LduCase(n, case.lower_addr, case.upper_addr, pdiag, upper, None, src, dims=dims),
                          orc.box_face_weights(case), 10)
    _, p = H.solve(z, src, tolerance=1e-8); exp.append(("GAMG", "p", p))
    H2 = orc.GamgHierarchy(syn.LduCase(n, case.lower_addr, case.upper_addr, pdiag, upper, None, src, dims=dims),
                           orc.box_face_weights(case), 10, merge_levels=2)
    _, p = H2.solve(z, src, tolerance=1e-8); exp.append(("GAMG", "p", p))
    _, p = P.smooth_solve(z, src, n_sweeps=2, tolerance=1e-3, maxIter=400); exp.append(("smoothSolver", "p", p))
    # UEqn = ddt + div(phi) - laplacian
    direction = np.where(case.upper_addr - case.lower_addr == 1, 0, 1)
    phi = np.where(direction == 0, 0.3 * h * h, 0.0)
    cl, cu, cd = orc.fvm_div(n, case.lower_addr, case.upper_addr, np.ones(nf), phi)
    ul, uu = cl - upper, cu - upper
    ddt_d, _ = orc.fvm_ddt_euler(1.0 / 1e-3, 1.0, np.full(n, h * h * h), z)     # UEqn = ddt; += conv; -= lap (operator order)
    ud = orc.axpby(1.0, orc.axpby(1.0, ddt_d, 1.0, cd), -1.0, diag)
    ud_solve = orc.patch_add(xmin, np.full(xmin.shape[0], 2.0 * h), ud, 0)
    U = orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, ud_solve, uu, ul, src)])
    _, p = U.pbicg(z, src, "AINV", tolerance=1e-10); exp.append(("AINVPBiCG", "Ux", p))
    _, p = U.pbicgstab(z, src, "AINV", tolerance=1e-10, replicate_quirk=True); exp.append(("AINVPBiCGStab", "Ux", p))
    # fvMatrix::A / H of the assembled UEqn (before relax), psi = src
    V = np.full(n, h * h * h)
    a_ref = ud_solve / V
    h_ref = (orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, ud, uu, ul, src)]).H(src) + src) / V
    m = re.search(r"A\(Ux\) sum max: (\S+) (\S+)  H\(Ux\) sum max: (\S+) (\S+)", out.stdout)
    assert m, out.stdout
    assert abs(float(m.group(1)) - a_ref.sum()) < 1e-9 * abs(a_ref.sum()) and float(m.group(2)) == np.max(np.abs(a_ref))
    assert abs(float(m.group(3)) - h_ref.sum()) < 1e-9 * np.abs(h_ref).sum() and float(m.group(4)) == np.max(np.abs(h_ref))
    # fvMatrix::flux of UEqn for psi = src: faceH on the internal faces, internalCoeffs*psi - boundaryCoeffs on the patches
    fh = orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, ud, uu, ul, src)]).faceH(src)
    bsum = orc.patch_flux(xmin, np.full(xmin.shape[0], 2.0 * h), np.zeros(xmin.shape[0]), src).sum()
    m = re.search(r"flux\(Ux\) sumMag max boundarySum: (\S+) (\S+) (\S+)", out.stdout)
    assert m, out.stdout
    assert abs(float(m.group(1)) - np.abs(fh).sum()) < 1e-11 * np.abs(fh).sum() and abs(float(m.group(2)) - np.max(np.abs(fh))) < 1e-14 * np.max(np.abs(fh))
    assert abs(float(m.group(3)) - bsum) < 1e-11 * np.abs(src[xmin]).sum() * 2.0 * h
    # fvMatrix::H with x-max as a coupled patch: + boundaryCoeffs * patchNeighbourField on its cells
    xmax_c = np.nonzero(np.arange(n) % dims[0] == dims[0] - 1)[0]
    idx = np.arange(xmax_c.shape[0])
    bc1 = 0.25 * h * (1.0 + 0.5 * syn.splitmix_at(4242, idx)); nbr1 = syn.splitmix_at(4343, idx) - 0.5
    hc = orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, ud, uu, ul, src)]).H(src) + src
    hc = orc.patch_add_product(xmax_c, bc1, nbr1, hc, 0) / V
    m = re.search(r"H\(Ux\) coupled sum max: (\S+) (\S+)", out.stdout)
    assert m, out.stdout
    assert abs(float(m.group(1)) - hc.sum()) < 1e-9 * np.abs(hc).sum() and abs(float(m.group(2)) - np.max(np.abs(hc))) < 1e-13 * np.max(np.abs(hc))
    # round 6: the fused assembly writes the same matrix bit for bit; setReference / setValues of the mirror against the oracle
    assert "fvm::assemble equals the operator sequence: 1" in out.stdout, out.stdout
    rdg, rsc = orc.set_reference(7, 0.5, ud, src)
    m = re.search(r"setReference cell 7: (\S+) (\S+)", out.stdout)
    assert m and abs(float(m.group(1)) - rdg[7]) < 1e-5 * abs(rdg[7]) and abs(float(m.group(2)) - rsc[7]) < 1e-5 * max(abs(rsc[7]), 1e-30), out.stdout   # (printed with 6 digits)
    sv = orc.set_values(n, case.lower_addr, case.upper_addr, np.array([3, 11, 40], np.int32), np.array([0.25, -0.5, 1.5]), src, rdg, rsc, uu, ul)
    m = re.search(r"setValues sums source upper lower psi11: (\S+) (\S+) (\S+) (\S+)", out.stdout)
    assert m, out.stdout
    for got_v, ref_v in zip(map(float, m.groups()), (sv["source"].sum(), sv["upper"].sum(), sv["lower"].sum(), -0.5)):
        assert abs(got_v - ref_v) < 2e-5 * max(abs(ref_v), 1e-12), (got_v, ref_v)
    rd, rs = orc.relax(n, case.lower_addr, case.upper_addr, 0.7, ud, ul, uu, src, z, [xmin], [np.full(xmin.shape[0], 2.0 * h)],
                       [np.zeros(xmin.shape[0])], [0])
    rd_solve = orc.patch_add(xmin, np.full(xmin.shape[0], 2.0 * h), rd, 0)
    UR = orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, rd_solve, uu, ul, rs)])
    _, p = UR.pbicg(z, rs, "diagonal", tolerance=1e-10); exp.append(("diagonalPBiCG", "Ux", p))
    # fvMatrix<vector>::solveSegregated: three component systems, per-component boundary coefficients
    xmax = np.nonzero(np.arange(n) % dims[0] == dims[0] - 1)[0]
    worst = dict(i=0.0, f=0.0, n=0)
    for d, cname in enumerate("xyz"):
        sd = (2.0 * syn.splitmix_uniform(900 + d, n) - 1.0) * h ** 3
        sd = orc.patch_add(xmax, np.full(xmax.shape[0], 0.01 * (d + 1) * h ** 3), sd, 0)          # addBoundarySource
        dd = orc.patch_add(xmin, np.full(xmin.shape[0], (2.0 + d) * h), ud, 0)                       # addBoundaryDiag(cmpt)
        _, p = orc.System([syn.LduCase(n, case.lower_addr, case.upper_addr, dd, uu, ul, sd)]).pbicg(z, sd, "AINV", tolerance=1e-10)
        exp.append(("AINVPBiCG", "U" + cname, p))
        worst = dict(i=max(worst["i"], p["initialResidual"]), f=max(worst["f"], p["finalResidual"]), n=max(worst["n"], p["nIterations"]))
    m = re.search(r"solveSegregated max: (\w+) (\S+) (\S+) (\d+)", out.stdout)
    assert m and m.group(1) == "AINVPBiCG" and int(m.group(4)) == worst["n"]
    assert abs(float(m.group(2)) - worst["i"]) < 1e-12 and abs(float(m.group(3)) - worst["f"]) < 1e-10
    assert len(got) == len(exp), out.stdout
    for (gname, gfield, gi, gf, gn), (ename, efield, p) in zip(got, exp):
        assert (gname, gfield) == (ename, efield)
        assert gn == p["nIterations"], (gname, gn, p["nIterations"])
        assert abs(gi - p["initialResidual"]) < 1e-12 and abs(gf - p["finalResidual"]) < 1e-10
    # error behaviour of the run-time selection table (lduMatrixSolver.C:84-100)
    assert "Unknown symmetric matrix solver PCGG" in out.stdout and "Valid symmetric matrix solvers are" in out.stdout
    assert re.search(r"\(GAMG ICCG PCG smoothSolver\)", out.stdout)                       # the reference's symMatrix table
    assert "Unknown symmetric matrix preconditioner DILU" in out.stdout and "4(AINV DIC diagonal none)" in out.stdout
    assert "GAMGSolver::interpolate()" in out.stdout and "Not implemented" in out.stdout    # interpolateCorrection true
    assert out.stdout.strip().endswith("End")


def _periodic_case(pkg, dims):
    """the system pEqnFoamPar assembles (global numbering), as an oracle case with two interfaces that face each other"""
    syn = pkg.synthetic
    nx, ny, nz = dims
    n, h = nx * ny * nz, 1.0 / nx
    c = np.arange(n, dtype=np.int64)
    i, j, k = c % nx, (c // nx) % ny, c // (nx * ny)
    lo, up, key = [], [], []
    for d, (mask, step) in enumerate(((i < nx - 1, 1), (j < ny - 1, nx), (k < nz - 1, nx * ny))):
        lo.append(c[mask]); up.append(c[mask] + step); key.append(c[mask] * 3 + d)
    lo, up, key = np.concatenate(lo), np.concatenate(up), np.concatenate(key)
    order = np.lexsort((up, lo))                      # owner-sorted, upper-triangular: the application's face order
    lo, up, key = lo[order], up[order], key[order]
    upper = h * (1.0 + 0.1 * syn.splitmix_uniform(12345, int(key.max()) + 1)[key])
    diag = np.zeros(n)
    np.subtract.at(diag, lo, upper); np.subtract.at(diag, up, upper)
    ymin, ymax = np.nonzero(j == 0)[0], np.nonzero(j == ny - 1)[0]
    diag[ymin] -= h; diag[ymax] -= h
    diag[i == 0] += -2.0 * h
    src = (2.0 * syn.splitmix_uniform(777, n) - 1.0) * h ** 3
    kap = np.full(ymin.shape[0], -h)
    ifs = [syn.Interface(nbr_domain=0, nbr_patch=1, face_cells=ymin.astype(np.int32), bou_coeffs=kap, int_coeffs=kap),
           syn.Interface(nbr_domain=0, nbr_patch=0, face_cells=ymax.astype(np.int32), bou_coeffs=kap, int_coeffs=kap)]
    return syn.LduCase(n, lo.astype(np.int32), up.astype(np.int32), diag, upper, None, src, dims=dims, interfaces=ifs)


def _as_ami(pkg, case, dims):
    """pEqnFoamPar's cyclicAMI mode: every face sees its opposite face (0.75) and that face's x-neighbour (0.25)"""
    syn = pkg.synthetic
    nx, ny, nz = dims
    h = 1.0 / nx
    for p, itf in enumerate(case.interfaces):
        shift = 1 if p == 0 else nx - 1
        q = np.arange(nx * nz)
        i, k = q % nx, q // nx
        addr = np.stack([i + nx * k, (i + shift) % nx + nx * k], axis=1).reshape(-1)
        itf.ami_start = (2 * np.arange(nx * nz + 1)).astype(np.int32)
        itf.ami_addr = addr.astype(np.int32)
        itf.ami_w = np.tile([0.75, 0.25], nx * nz)
        itf.ami_magsf = h * h * (1.0 + 0.05 * syn.splitmix_uniform(555 + p, nx * nz))
    return case


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["cyclic", "processor", "cyclicAMI"])
def test_pEqnFoamPar_matches_oracle(pkg, orc, mode, tmp_path):
    """lduMatrix::solver::New(...)->solve on a matrix with cyclic patches / on a (1-rank) decomposed case whose halo goes
    through RCCL (Pstream::init -> mi_matrix_attach_comm, mi_gamg_create_coupled), against the oracle's system."""
    dims = (12, 10, 8)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MI_COMM_ID_FILE=str(tmp_path / "ids"))
    out = subprocess.run([os.path.join(PKG, "pEqnFoamPar"), *map(str, dims), mode], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr
    got = [(m.group(1), m.group(2), float(m.group(3)), float(m.group(4)), int(m.group(5))) for m in map(LINE.match, out.stdout.splitlines()) if m]
    case = _periodic_case(pkg, dims)
    if mode == "cyclicAMI":
        case = _as_ami(pkg, case, dims)
    S = orc.System([case])
    z, src = np.zeros(case.n_cells), case.source
    exp = []
    _, p = S.pcg(z, src, "diagonal", tolerance=1e-8); exp.append(("diagonalPCG", p))
    _, p = S.pcg(z, src, "AINV", tolerance=1e-8); exp.append(("AINVPCG", p))
    _, p = orc.GamgSysHierarchy(S, [orc.box_face_weights(case)], 10).solve(z, src, tolerance=1e-8, directSolveCoarsest=(mode != "cyclicAMI")); exp.append(("GAMG", p))
    _, p = S.smooth_solve(z, src, n_sweeps=2, tolerance=1e-3, maxIter=400); exp.append(("smoothSolver", p))
    _, p = S.pbicgstab(z, src, "diagonal", tolerance=0.0, maxIter=12, replicate_quirk=True); exp.append(("diagonalPBiCGStab", p))
    assert len(got) == len(exp), out.stdout + out.stderr
    for (gname, gfield, gi, gf, gn), (ename, p) in zip(got, exp):
        assert (gname, gfield) == (ename, "p")
        assert gn == p["nIterations"], (gname, gn, p["nIterations"])
        assert abs(gi - p["initialResidual"]) < 1e-12 and abs(gf - p["finalResidual"]) < 1e-10
    assert out.stdout.strip().endswith("End")
