"""Writes tests/golden/full_size_v1.npz: the CPU oracle's results for the full-size GPU tests (216^3, 108^3, 640x250x250, 432^3; section
assembly216: every fvMatrix-assembly operator at 216^3 in the caller's numbering and under ordered addressing).

Run on the CPU box from the repo root (tens of minutes, ~45 GB of RAM for the 80 M-cell section):

    python tests/golden/make_full_size.py                 # every section
    python tests/golden/make_full_size.py box216_sym ...  # some sections; the others keep their records

What each record is and how the GPU tests use it: tests/full_size_ref.py.  The solves are EXACTLY the oracle calls the GPU
tests made in-process up to round 4 (same seeds, controls and case generators; test_gpu_full_size.py / test_gpu_configs.py
still make them with MI_LIVE_ORACLE=1).  The oracle follows PCG.C:133-204, PBiCG.C:67-246, PBiCGStab.C:67-300,
GAMGSolverSolve.C:59-474 (oracle/ldu_oracle.c, oracle/gamg_oracle.c cite them function by function).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import __graft_entry__ as graft  # noqa: E402
import full_size_ref as fs  # noqa: E402

N = 216


def box216_sym(pkg, orc, out):
    syn = pkg.synthetic
    case = syn.box_case(N, N, N)
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(99, n) - 0.5
    k = "box216_sym"
    fs.pack_sha(out, k + "/amul", S.amul(x)); fs.pack_sha(out, k + "/tmul", S.tmul(x)); fs.pack_sha(out, k + "/sumA", S.sumA())
    fs.pack_sha(out, k + "/residual", S.residual(x, case.source)); fs.pack_sha(out, k + "/H", S.H(x)); fs.pack_sha(out, k + "/H1", S.H1())
    for kind in ("diagonal", "AINV"):
        fs.pack_sha(out, f"{k}/precondition_{kind}", S.precondition(kind, x))
    fs.pack_sha(out, k + "/jacobi_2_sweeps", S.jacobi_smooth(x, case.source, 2, omega=0.9))
    # a few values in the clear, so that tests/test_full_size_fixture.py can say WHERE a re-derived vector differs
    out[k + "/amul_sample"] = S.amul(x)[fs.sample_idx(n)]
    z = np.zeros(n)
    for precond in ("diagonal", "AINV"):
        psi, perf = S.pcg(z, case.source, precond, tolerance=0.0, maxIter=120)
        fs.pack_perf(out, f"{k}/pcg_{precond}_120", perf); fs.pack_solution(out, f"{k}/pcg_{precond}_120", psi)
    psi, perf = S.pcg(z, case.source, "diagonal", tolerance=1e-6, maxIter=5000)
    fs.pack_perf(out, k + "/pcg_diagonal_to_1e-6", perf); fs.pack_solution(out, k + "/pcg_diagonal_to_1e-6", psi)
    psi, perf = S.pcg(z, case.source, "diagonal", tolerance=0.0, maxIter=60)      # the 2 x 2 x 2 decomposed run's reference
    fs.pack_perf(out, k + "/pcg_diagonal_60", perf); fs.pack_solution(out, k + "/pcg_diagonal_60", psi)
    w = orc.box_face_weights(case)
    H = orc.GamgHierarchy(case, w, 100)
    psi, perf = H.solve(z, case.source, tolerance=1e-6, maxIter=100)
    fs.pack_perf(out, k + "/gamg_to_1e-6", perf); fs.pack_solution(out, k + "/gamg_to_1e-6", psi)
    out[k + "/gamg_levels"] = np.array(H.n_levels)


def box216_asym(pkg, orc, out):
    syn = pkg.synthetic
    case = syn.box_case(N, N, N, symmetric=False)
    S = orc.System([case])
    n = case.n_cells
    x = syn.splitmix_uniform(98, n) - 0.5
    k = "box216_asym"
    fs.pack_sha(out, k + "/amul", S.amul(x)); fs.pack_sha(out, k + "/tmul", S.tmul(x))
    for tr in (False, True):
        fs.pack_sha(out, f"{k}/precondition_AINV_transpose{int(tr)}", S.precondition("AINV", x, transpose=tr))
    z = np.zeros(n)
    psi, perf = S.pbicg(z, case.source, "AINV", tolerance=0.0, maxIter=40)
    fs.pack_perf(out, k + "/pbicg_AINV_40", perf); fs.pack_solution(out, k + "/pbicg_AINV_40", psi)
    psi, perf = S.pbicgstab(z, case.source, "AINV", tolerance=0.0, maxIter=24)
    fs.pack_perf(out, k + "/pbicgstab_AINV_24", perf); fs.pack_solution(out, k + "/pbicgstab_AINV_24", psi)
    psi, perf = S.pbicgstab(z, case.source, "diagonal", replicate_quirk=False, tolerance=0.0, maxIter=48)
    fs.pack_perf(out, k + "/pbicgstab_diagonal_zA_48", perf); fs.pack_solution(out, k + "/pbicgstab_diagonal_zA_48", psi)
    out[k + "/pbicgstab_diagonal_zA_48/true_residual"] = np.array(np.abs(S.residual(psi, case.source)).sum() / perf["normFactor"])


def persist108(pkg, orc, out):
    syn = pkg.synthetic
    for form in ("single_rank", "distributed_self_exchange"):
        case = syn.box_case(108, 108, 108)
        if form != "single_rank":
            case = syn.add_cyclic_y(case)
        S = orc.System([case])
        z = np.zeros(case.n_cells)
        for tag, okw in (("300_iterations", dict(tolerance=0.0, maxIter=299)), ("to_1e-8", dict(tolerance=1e-8, maxIter=3000))):
            psi, perf = S.pcg(z, case.source, "diagonal", **okw)
            fs.pack_perf(out, f"persist108/{form}/{tag}", perf); fs.pack_solution(out, f"persist108/{form}/{tag}", psi)


def config4(pkg, orc, out):
    syn = pkg.synthetic
    case = syn.add_cyclic_x(syn.box_case(640, 250, 250))
    n = case.n_cells
    S = orc.System([case])
    k = "config4"
    fs.pack_sha(out, k + "/amul", S.amul(syn.splitmix_uniform(5, n) - 0.5))
    z = np.zeros(n)
    psi, perf = S.pcg(z, case.source, "diagonal", tolerance=0.0, maxIter=20)
    fs.pack_perf(out, k + "/pcg_diagonal_20", perf); fs.pack_solution(out, k + "/pcg_diagonal_20", psi)
    H = orc.GamgSysHierarchy(S, [orc.box_face_weights(case)], 100)
    psi, perf = H.solve(z, case.source, tolerance=0.0, maxIter=5)
    fs.pack_perf(out, k + "/gamg_5", perf); fs.pack_solution(out, k + "/gamg_5", psi)
    out[k + "/gamg_levels"] = np.array(H.n_levels)


def config5(pkg, orc, out):
    syn = pkg.synthetic
    dims, parts, world = (432, 432, 432), (2, 2, 2), 8
    for tag, symmetric in (("bicg", False), ("gamg", True)):
        subs = [syn.box_subdomain(dims, parts, r, symmetric=symmetric) for r in range(world)]
        S = orc.System(subs)
        n = sum(s.n_cells for s in subs)
        assert n == 432 ** 3
        src = np.concatenate([s.source for s in subs])
        offs = np.concatenate([[0], np.cumsum([s.n_cells for s in subs])])
        if tag == "bicg":
            rp, r = S.pbicg(np.zeros(n), src, "AINV", tolerance=0.0, maxIter=10)
        else:
            Hh = orc.GamgSysHierarchy(S, [orc.box_face_weights(s) for s in subs], 100)
            rp, r = Hh.solve(np.zeros(n), src, tolerance=0.0, maxIter=3)
            out["config5/gamg_levels"] = np.array(Hh.n_levels)
            del Hh
        fs.pack_perf(out, f"config5/{tag}", r)
        out[f"config5/{tag}/rank_sum"] = np.array([float(rp[offs[q]:offs[q + 1]].sum()) for q in range(world)])
        out[f"config5/{tag}/rank_abs"] = np.array([float(np.abs(rp[offs[q]:offs[q + 1]]).sum()) for q in range(world)])
        del S, subs, rp, src


def assembly216(pkg, orc, out):
    """the fvMatrix-assembly operators at 216^3 on both meshes (tests/assembly_full_size.py): sha256 of every result's bits, a sample of
    each in the clear, and the sha256 of the renumbered addressing the `ordered` records belong to"""
    import assembly_full_size as afs
    for variant in afs.VARIANTS:
        M = afs.mesh(pkg, variant)
        q = afs.inputs(pkg, M)
        fs.pack_sha(out, f"assembly216/{variant}/lowerAddr", M["lo"]); fs.pack_sha(out, f"assembly216/{variant}/upperAddr", M["up"])
        res = afs.oracle_run(pkg, orc, M, q)
        out[f"assembly216/{variant}/names"] = np.array(sorted(res))
        for name, a in res.items():
            fs.pack_sha(out, f"assembly216/{variant}/{name}", a)
            out[f"assembly216/{variant}/{name}/sample"] = a[fs.sample_idx(a.shape[0], 256)].copy()
        del res, q, M


SECTIONS = dict(assembly216=assembly216, box216_sym=box216_sym, box216_asym=box216_asym, persist108=persist108, config4=config4, config5=config5)


def main():
    graft.build()
    pkg = graft.load_package()
    from oracle import oracle as orc
    names = sys.argv[1:] or list(SECTIONS)
    out = dict(np.load(fs.FIXTURE)) if os.path.exists(fs.FIXTURE) else {}
    for name in names:
        t0 = time.perf_counter()
        for key in [q for q in out if q.startswith(name + "/")]:
            del out[key]
        SECTIONS[name](pkg, orc, out)
        out[f"seconds/{name}"] = np.array(time.perf_counter() - t0)
        for key in [q for q in out if q.startswith(f"sources/{name}/")]:
            del out[key]
        for rel, h in fs.source_hashes(name, SECTIONS[name]).items():
            out[f"sources/{name}/{rel}"] = np.array(h)
        np.savez_compressed(fs.FIXTURE, **out)       # after every section: a killed run keeps what it finished
        print(f"{name}: {time.perf_counter() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
