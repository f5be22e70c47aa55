"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- may be imported by tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py, never by the product package.
PARITY UNPINNED: see the header of ldu_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PRECOND = {"none": 0, "diagonal": 1, "AINV": 2, "DIC": 2, "DILU": 2,  # RapidCFD aliases
           "DIC_upstream": 3, "DILU_upstream": 4}


class Perf(C.Structure):
    _fields_ = [("initialResidual", C.c_double), ("finalResidual", C.c_double),
                ("normFactor", C.c_double), ("nIterations", C.c_int32),
                ("converged", C.c_int32), ("singular", C.c_int32)]


class Controls(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("relTol", C.c_double),
                ("maxIter", C.c_int32), ("minIter", C.c_int32)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith("_oracle.c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    # oracle/_ref: the reference's own host code compiled where it lies; only where /root/reference exists (the GPU box
    # uses the prebuilt file that travelled with the snapshot)
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_sys_create.restype = C.c_void_p
        L.orc_sys_size.restype = C.c_int64
        for name in ("orc_gSumProd", "orc_gSumMag", "orc_gSum", "orc_norm_factor"):
            getattr(L, name).restype = C.c_double
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class System:
    """D sub-domains solved in lock-step (D=1: an ordinary serial case)."""

    def __init__(self, cases: Sequence):
        L = lib()
        self.cases = list(cases)
        self.h = C.c_void_p(L.orc_sys_create(len(self.cases)))
        for d, cs in enumerate(self.cases):
            lo, up = _i(cs.lower_addr), _i(cs.upper_addr)
            L.orc_sys_set_domain(self.h, d, C.c_int32(cs.n_cells), C.c_int32(cs.n_faces),
                                 _p(lo, C.c_int32), _p(up, C.c_int32), _p(_d(cs.diag), C.c_double),
                                 _p(_d(cs.lower), C.c_double) if cs.lower is not None else None,
                                 _p(_d(cs.upper), C.c_double))
        for d, cs in enumerate(self.cases):
            for itf in cs.interfaces:
                fc = _i(itf.face_cells)
                L.orc_sys_add_interface(self.h, d, C.c_int32(itf.nbr_domain), C.c_int32(itf.nbr_patch),
                                        C.c_int32(fc.shape[0]), _p(fc, C.c_int32),
                                        _p(_d(itf.bou_coeffs), C.c_double), _p(_d(itf.int_coeffs), C.c_double))
        for d, cs in enumerate(self.cases):
            for p, itf in enumerate(cs.interfaces):
                if getattr(itf, "ami_start", None) is not None:
                    low = None if itf.ami_low is None else np.ascontiguousarray(itf.ami_low, dtype=np.uint8)
                    L.orc_sys_set_iface_ami(self.h, d, p, _p(_i(itf.ami_start), C.c_int32), _p(_i(itf.ami_addr), C.c_int32),
                                            _p(_d(itf.ami_w), C.c_double), _p(low, C.c_uint8))
                if getattr(itf, "ami_parts", None):      # partner side split over several domains: [(domain, interface), ...], addresses concatenated
                    pd, pp = _i([q[0] for q in itf.ami_parts]), _i([q[1] for q in itf.ami_parts])
                    L.orc_sys_set_iface_ami_parts(self.h, d, p, C.c_int(len(itf.ami_parts)), _p(pd, C.c_int32), _p(pp, C.c_int32))
                if getattr(itf, "ami_magsf", None) is not None:
                    L.orc_sys_set_iface_magsf(self.h, d, p, _p(_d(itf.ami_magsf), C.c_double))
                if getattr(itf, "transform", 1.0) != 1.0:
                    L.orc_sys_set_iface_transform(self.h, d, p, C.c_double(itf.transform))
        self.n = int(L.orc_sys_size(self.h))

    def __del__(self):
        try:
            lib().orc_sys_destroy(self.h)
        except Exception:
            pass

    def baseline_pcg(self, source, n_iter):
        """timing kernel of bench.py's cpu_baseline: one OpenMP thread per domain = one MPI rank per core (baseline_oracle.c)"""
        sec, res = C.c_double(), C.c_double()
        n = lib().orc_baseline_pcg(self.h, _p(_d(source), C.c_double), C.c_int(n_iter), C.byref(sec), C.byref(res))
        return int(n), float(sec.value), float(res.value)

    def set_accurate(self, on: bool):
        lib().orc_sys_set_accurate(self.h, int(on))

    # --- operators ------------------------------------------------------
    def _unary(self, fn, psi):
        x = _d(psi)
        y = np.empty(self.n)
        getattr(lib(), fn)(self.h, _p(x, C.c_double), _p(y, C.c_double))
        return y

    def amul(self, psi):
        return self._unary("orc_amul", psi)

    def amul_functor_literal(self, psi):
        return self._unary("orc_amul_functor_literal", psi)

    def tmul(self, psi):
        return self._unary("orc_tmul", psi)

    def amul_faceloop(self, psi):
        return self._unary("orc_amul_faceloop", psi)

    def H(self, psi):
        return self._unary("orc_H", psi)

    def sumA(self):
        y = np.empty(self.n)
        lib().orc_sumA(self.h, _p(y, C.c_double))
        return y

    def H1(self):
        y = np.empty(self.n)
        lib().orc_H1(self.h, _p(y, C.c_double))
        return y

    def faceH(self, psi, d=0):
        x = _d(psi)
        y = np.empty(self.cases[d].n_faces)
        lib().orc_faceH(self.h, d, _p(x, C.c_double), _p(y, C.c_double))
        return y

    def residual(self, psi, source):
        x, b = _d(psi), _d(source)
        y = np.empty(self.n)
        lib().orc_residual(self.h, _p(x, C.c_double), _p(b, C.c_double), _p(y, C.c_double))
        return y

    def norm_factor(self, psi, source, Apsi):
        x, b, a = _d(psi), _d(source), _d(Apsi)
        tmp = np.empty(self.n)
        return float(lib().orc_norm_factor(self.h, _p(x, C.c_double), _p(b, C.c_double),
                                           _p(a, C.c_double), _p(tmp, C.c_double)))

    def precondition(self, kind, r, transpose=False):
        x = _d(r)
        y = np.empty(self.n)
        lib().orc_precondition(self.h, PRECOND[kind], int(transpose), _p(x, C.c_double), _p(y, C.c_double))
        return y

    def jacobi_smooth(self, psi, source, n_sweeps, omega=0.9):
        x = _d(psi).copy()
        b = _d(source)
        lib().orc_jacobi_smooth(self.h, C.c_double(omega), _p(x, C.c_double), _p(b, C.c_double), n_sweeps)
        return x

    def gauss_seidel_upstream(self, psi, source, n_sweeps):
        x = _d(psi).copy()
        b = _d(source)
        lib().orc_gauss_seidel_upstream(self.h, _p(x, C.c_double), _p(b, C.c_double), n_sweeps)
        return x

    # --- solvers --------------------------------------------------------
    def _solve(self, fn, psi, source, extra, tolerance, relTol, maxIter, minIter, hist_len):
        x = _d(psi).copy()
        b = _d(source)
        ctl = Controls(tolerance, relTol, maxIter, minIter)
        perf = Perf()
        hist = np.full(hist_len, np.nan)
        getattr(lib(), fn)(self.h, _p(x, C.c_double), _p(b, C.c_double), C.byref(ctl), *extra,
                           C.byref(perf), _p(hist, C.c_double), hist_len)
        out = {k: getattr(perf, k) for k, _ in Perf._fields_}
        out["history"] = hist[~np.isnan(hist)].copy()
        return x, out

    def pcg(self, psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0):
        return self._solve("orc_pcg_solve", psi, source, (C.c_int(PRECOND[precond]),),
                           tolerance, relTol, maxIter, minIter, maxIter + 2)

    def pbicg(self, psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0):
        return self._solve("orc_pbicg_solve", psi, source, (C.c_int(PRECOND[precond]),),
                           tolerance, relTol, maxIter, minIter, maxIter + 2)

    def pbicgstab(self, psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000,
                  minIter=0, replicate_quirk=True):
        return self._solve("orc_pbicgstab_solve", psi, source,
                           (C.c_int(PRECOND[precond]), C.c_int(int(replicate_quirk))),
                           tolerance, relTol, maxIter, minIter, maxIter + 2)

    def smooth_solve(self, psi, source, n_sweeps=1, omega=0.9, tolerance=1e-6, relTol=0.0,
                     maxIter=1000, minIter=0):
        return self._solve("orc_smooth_solve", psi, source, (C.c_double(omega), C.c_int(n_sweeps)),
                           tolerance, relTol, maxIter, minIter, maxIter + 2)


# ---------------------------------------------------------------------------------------------
# GAMG (gamg_oracle.c)
# ---------------------------------------------------------------------------------------------
class GamgControls(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("relTol", C.c_double), ("maxIter", C.c_int32), ("minIter", C.c_int32),
                ("nPreSweeps", C.c_int32), ("preSweepsLevelMultiplier", C.c_int32), ("maxPreSweeps", C.c_int32),
                ("nPostSweeps", C.c_int32), ("postSweepsLevelMultiplier", C.c_int32), ("maxPostSweeps", C.c_int32),
                ("nFinestSweeps", C.c_int32), ("scaleCorrection", C.c_int32), ("omega", C.c_double),
                ("directSolveCoarsest", C.c_int32), ("reserved", C.c_int32)]


def gamg_controls(tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0, nPreSweeps=0, preSweepsLevelMultiplier=1,
                  maxPreSweeps=4, nPostSweeps=2, postSweepsLevelMultiplier=1, maxPostSweeps=4, nFinestSweeps=2,
                  scaleCorrection=-1, omega=0.9, directSolveCoarsest=True):
    """defaults of GAMGSolver.C:67-77 and lduMatrixSolver.C:167-173"""
    return GamgControls(tolerance, relTol, maxIter, minIter, nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps,
                        nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps, nFinestSweeps, scaleCorrection, omega,
                        int(bool(directSolveCoarsest)), 0)


def box_face_weights(case):
    """faceAreaPair weights |Sf/sqrt|Sf| o (1,1.01,1.02)| on the uniform hex box
    (faceAreaPairGAMGAgglomeration.C:54-81): |Sf| = h^2 along direction d."""
    nx, ny, nz = case.dims
    lo, up = case.lower_addr.astype(np.int64), case.upper_addr.astype(np.int64)
    d = up - lo
    direction = np.where(d == 1, 0, np.where(d == nx, 1, 2))
    if nx == 1 or ny == 1:  # degenerate boxes: recompute robustly
        direction = np.where(d == 1, 0, np.where((d == nx) & (ny > 1), 1, 2))
    h = 1.0 / nx
    return h * np.array([1.0, 1.01, 1.02])[direction]


class GamgHierarchy:
    def __init__(self, case, face_weights, n_cells_in_coarsest_level=10, forward=True, merge_levels=1, dummy_levels=0):
        """dummy_levels n > 0: the reference's dummyAgglomeration (n identity levels) instead of the pair agglomeration"""
        L = lib()
        L.orc_gamg_build_merged.restype = C.c_void_p
        L.orc_gamg_build_dummy.restype = C.c_void_p
        self.case = case
        lo, up = _i(case.lower_addr), _i(case.upper_addr)
        self._keep = (lo, up)
        if dummy_levels > 0:
            self.h = C.c_void_p(L.orc_gamg_build_dummy(C.c_int32(case.n_cells), C.c_int32(case.n_faces), _p(lo, C.c_int32), _p(up, C.c_int32), int(dummy_levels)))
        else:
            w = _d(face_weights)
            self.h = C.c_void_p(L.orc_gamg_build_merged(C.c_int32(case.n_cells), C.c_int32(case.n_faces), _p(lo, C.c_int32),
                                                        _p(up, C.c_int32), _p(w, C.c_double), C.c_int32(n_cells_in_coarsest_level),
                                                        int(forward), int(merge_levels)))
        self.n_levels = int(L.orc_gamg_n_levels(self.h))
        self.forward_out = bool(L.orc_gamg_forward_out(self.h))

    def __del__(self):
        try:
            lib().orc_gamg_free(self.h)
        except Exception:
            pass

    def level(self, l):
        s = (C.c_int32 * 4)()
        lib().orc_gamg_level_sizes(self.h, l, s)
        nf, nff, nc, ncf = [int(v) for v in s]
        rm, fr, ff = np.empty(nf, np.int32), np.empty(nff, np.int32), np.empty(nff, np.int32)
        cl, cu = np.empty(ncf, np.int32), np.empty(ncf, np.int32)
        lib().orc_gamg_level_maps(self.h, l, _p(rm, C.c_int32), _p(fr, C.c_int32), _p(ff, C.c_int32),
                                  _p(cl, C.c_int32), _p(cu, C.c_int32))
        return dict(n_fine=nf, n_fine_faces=nff, n_coarse=nc, n_coarse_faces=ncf, restrict=rm, face_restrict=fr,
                    face_flip=ff.astype(bool), lower=cl, upper=cu)

    def restrict(self, l, ff):
        lv = self.level(l)
        cf = np.empty(lv["n_coarse"])
        lib().orc_gamg_restrict_level(self.h, l, _p(_d(ff), C.c_double), _p(cf, C.c_double))
        return cf

    def prolong(self, l, cf):
        lv = self.level(l)
        ff = np.empty(lv["n_fine"])
        lib().orc_gamg_prolong_level(self.h, l, _p(_d(cf), C.c_double), _p(ff, C.c_double))
        return ff

    def coarse_matrix(self, up_to_level):
        cs = self.case
        lv = self.level(up_to_level)
        d, u = np.empty(lv["n_coarse"]), np.empty(lv["n_coarse_faces"])
        lo = np.empty(lv["n_coarse_faces"]) if cs.lower is not None else None
        lib().orc_gamg_coarse_matrix(self.h, up_to_level, _p(_d(cs.diag), C.c_double), _p(_d(cs.upper), C.c_double),
                                     _p(_d(cs.lower), C.c_double) if cs.lower is not None else None,
                                     _p(d, C.c_double), _p(u, C.c_double), _p(lo, C.c_double) if lo is not None else None)
        return d, u, lo

    def solve(self, psi, source, **kw):
        cs = self.case
        ctl = gamg_controls(**kw)
        x = _d(psi).copy()
        b = _d(source)
        perf = Perf()
        hist_len = ctl.maxIter + 2
        hist = np.full(hist_len, np.nan)
        lo, up = self._keep
        lib().orc_gamg_solve(self.h, _p(lo, C.c_int32), _p(up, C.c_int32), _p(_d(cs.diag), C.c_double),
                             _p(_d(cs.upper), C.c_double), _p(_d(cs.lower), C.c_double) if cs.lower is not None else None,
                             _p(x, C.c_double), _p(b, C.c_double), C.byref(ctl), C.byref(perf), _p(hist, C.c_double), hist_len)
        out = {k: getattr(perf, k) for k, _ in Perf._fields_}
        out["history"] = hist[~np.isnan(hist)].copy()
        return x, out


class GamgSysHierarchy:
    """GAMG over a System: coupled patches (cyclic) and decomposed cases (orc_gamg_build_sys / solve_sys)."""

    def __init__(self, system, face_weights_per_domain, n_cells_in_coarsest_level=10, forward=True, merge_levels=1):
        L = lib()
        L.orc_gamg_build_sys_merged.restype = C.c_void_p
        self.system = system
        w = _d(np.concatenate([np.asarray(x, dtype=np.float64) for x in face_weights_per_domain]))
        self.h = C.c_void_p(L.orc_gamg_build_sys_merged(system.h, _p(w, C.c_double), C.c_int32(n_cells_in_coarsest_level), int(forward),
                                                        int(merge_levels)))
        self.n_levels = int(L.orc_gamg_sys_n_levels(self.h))

    def __del__(self):
        try:
            lib().orc_gamg_sys_free(self.h)
        except Exception:
            pass

    def level(self, d, l):
        s = (C.c_int32 * 4)()
        lib().orc_gamg_sys_level_sizes(self.h, d, l, s)
        nf, nff, nc, ncf = [int(v) for v in s]
        rm, cl, cu = np.empty(nf, np.int32), np.empty(ncf, np.int32), np.empty(ncf, np.int32)
        lib().orc_gamg_sys_level_maps(self.h, d, l, _p(rm, C.c_int32), _p(cl, C.c_int32), _p(cu, C.c_int32))
        return dict(n_fine=nf, n_fine_faces=nff, n_coarse=nc, n_coarse_faces=ncf, restrict=rm, lower=cl, upper=cu)

    def patch(self, d, l, p, n_fine_patch_faces):
        fr = np.empty(n_fine_patch_faces, np.int32)
        fc = np.empty(max(n_fine_patch_faces, 1), np.int32)
        nc = int(lib().orc_gamg_sys_patch(self.h, d, l, p, _p(fr, C.c_int32), _p(fc, C.c_int32)))
        return dict(face_restrict=fr, face_cells=fc[:nc].copy())

    def patch_ami(self, d, l, p, n_coarse_faces):
        """agglomerated AMI of cyclicAMI patch p on the coarse side of level l (None for other patches)"""
        na = int(lib().orc_gamg_sys_patch_ami(self.h, d, l, p, None, None, None, None))
        if na < 0:
            return None
        st, ad = np.empty(n_coarse_faces + 1, np.int32), np.empty(max(na, 1), np.int32)
        w, ms = np.empty(max(na, 1)), np.empty(max(n_coarse_faces, 1))
        lib().orc_gamg_sys_patch_ami(self.h, d, l, p, _p(st, C.c_int32), _p(ad, C.c_int32), _p(w, C.c_double), _p(ms, C.c_double))
        return dict(start=st, addr=ad[:na].copy(), w=w[:na].copy(), magsf=ms[:n_coarse_faces].copy())

    def solve(self, psi, source, **kw):
        ctl = gamg_controls(**kw)
        x = _d(psi).copy()
        b = _d(source)
        perf = Perf()
        hist_len = ctl.maxIter + 2
        hist = np.full(hist_len, np.nan)
        lib().orc_gamg_solve_sys(self.h, self.system.h, _p(x, C.c_double), _p(b, C.c_double), C.byref(ctl), C.byref(perf),
                                 _p(hist, C.c_double), hist_len)
        out = {k: getattr(perf, k) for k, _ in Perf._fields_}
        out["history"] = hist[~np.isnan(hist)].copy()
        return x, out


# ---------------------------------------------------------------------------------------------
# fvMatrix assembly sweeps (fvm_oracle.c)
# ---------------------------------------------------------------------------------------------
def row_face_op(kind, n_cells, lower_addr, upper_addr, lower, upper, inout):
    lo, up = _i(lower_addr), _i(upper_addr)
    out = _d(inout).copy()
    u = _d(upper)
    l = u if lower is None else _d(lower)
    lib().orc_row_face_op(int(kind), C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32),
                          _p(l, C.c_double), _p(u, C.c_double), _p(out, C.c_double))
    return out


def fvm_laplacian(n_cells, lower_addr, upper_addr, delta_coeffs, gamma_magsf):
    lo, up = _i(lower_addr), _i(upper_addr)
    upper, diag = np.empty(lo.shape[0]), np.empty(n_cells)
    lib().orc_fvm_laplacian(C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32),
                            _p(_d(delta_coeffs), C.c_double), _p(_d(gamma_magsf), C.c_double), _p(upper, C.c_double), _p(diag, C.c_double))
    return upper, diag


def fvm_div(n_cells, lower_addr, upper_addr, weights, face_flux):
    lo, up = _i(lower_addr), _i(upper_addr)
    lower, upper, diag = np.empty(lo.shape[0]), np.empty(lo.shape[0]), np.empty(n_cells)
    lib().orc_fvm_div(C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32),
                      _p(_d(weights), C.c_double), _p(_d(face_flux), C.c_double), _p(lower, C.c_double),
                      _p(upper, C.c_double), _p(diag, C.c_double))
    return lower, upper, diag


def patch_add(face_cells, pf, intf, fn=0):
    fc = _i(face_cells)
    out = _d(intf).copy()
    lib().orc_patch_add(C.c_int32(fc.shape[0]), _p(fc, C.c_int32), _p(_d(pf), C.c_double), int(fn), C.c_int32(out.shape[0]), _p(out, C.c_double))
    return out


def patch_add_product(face_cells, pf, q, intf, fn=0):
    fc = _i(face_cells)
    out = _d(intf).copy()
    lib().orc_patch_add_product(C.c_int32(fc.shape[0]), _p(fc, C.c_int32), _p(_d(pf), C.c_double), _p(_d(q), C.c_double), int(fn), _p(out, C.c_double))
    return out


def patch_flux(face_cells, internal_coeffs, boundary_coeffs, psi, patch_neighbour_field=None):
    """boundary part of fvMatrix::flux (fvMatrix.C:1621-1653)"""
    fc = _i(face_cells)
    out = np.empty(fc.shape[0])
    nb = None if patch_neighbour_field is None else _d(patch_neighbour_field)
    lib().orc_patch_flux(C.c_int32(fc.shape[0]), _p(fc, C.c_int32), _p(_d(internal_coeffs), C.c_double), _p(_d(boundary_coeffs), C.c_double),
                         _p(_d(psi), C.c_double), _p(nb, C.c_double) if nb is not None else None, _p(out, C.c_double))
    return out


def relax(n_cells, lower_addr, upper_addr, alpha, diag, lower, upper, source, psi, face_cells=(), icoeffs=(), bcoeffs=(), coupled=()):
    lo, up = _i(lower_addr), _i(upper_addr)
    d, s = _d(diag).copy(), _d(source).copy()
    u = _d(upper)
    l = u if lower is None else _d(lower)
    n = len(face_cells)
    fcs = [_i(f) for f in face_cells]
    ics = [_d(a) for a in icoeffs]
    bcs = [_d(a) for a in bcoeffs]
    sizes = (C.c_int32 * max(n, 1))(*[f.shape[0] for f in fcs])
    fp = (C.POINTER(C.c_int32) * max(n, 1))(*[_p(f, C.c_int32) for f in fcs])
    ip = (C.POINTER(C.c_double) * max(n, 1))(*[_p(a, C.c_double) for a in ics])
    bp = (C.POINTER(C.c_double) * max(n, 1))(*[_p(a, C.c_double) for a in bcs])
    cp = (C.c_int * max(n, 1))(*[int(c) for c in coupled])
    lib().orc_relax(C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), C.c_double(alpha),
                    _p(d, C.c_double), _p(l, C.c_double), _p(u, C.c_double), _p(s, C.c_double), _p(_d(psi), C.c_double),
                    n, sizes, fp, ip, bp, cp)
    return d, s


def surface_integrate(n_cells, lower_addr, upper_addr, ssf, vol=None):
    lo, up = _i(lower_addr), _i(upper_addr)
    out = np.empty(n_cells)
    lib().orc_surface_integrate(C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32),
                                _p(_d(ssf), C.c_double), _p(_d(vol), C.c_double) if vol is not None else None, _p(out, C.c_double))
    return out


def face_interpolate(lower_addr, upper_addr, lam, phi):
    lo, up = _i(lower_addr), _i(upper_addr)
    out = np.empty(lo.shape[0])
    lib().orc_face_interpolate(C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(_d(lam), C.c_double),
                               _p(_d(phi), C.c_double), _p(out, C.c_double))
    return out


def fvm_ddt_euler(r_delta_t, rho, vol, psi_old):
    v, p0 = _d(vol), _d(psi_old)
    diag, src = np.empty_like(v), np.empty_like(v)
    lib().orc_fvm_ddt_euler(C.c_int32(v.shape[0]), C.c_double(r_delta_t), C.c_double(rho), _p(v, C.c_double), _p(p0, C.c_double),
                            _p(diag, C.c_double), _p(src, C.c_double))
    return diag, src


def fvm_ddt_euler_rho(r_delta_t, rho, rho_old, vol, psi_old):
    """fvm::ddt(rho, vf) with a density field (EulerDdtScheme.C:403-440) -> (diag, source)"""
    v, p0, r, r0 = _d(vol), _d(psi_old), _d(rho), _d(rho_old)
    diag, src = np.empty_like(v), np.empty_like(v)
    lib().orc_fvm_ddt_euler_rho(C.c_int32(v.shape[0]), C.c_double(r_delta_t), _p(r, C.c_double), _p(r0, C.c_double), _p(v, C.c_double),
                                _p(p0, C.c_double), _p(diag, C.c_double), _p(src, C.c_double))
    return diag, src


def fvm_su(vol, su, source):
    """fvm::Su (fvmSup.C:34-54): returns source - V*su"""
    out = _d(source).copy()
    lib().orc_fvm_su(C.c_int32(out.shape[0]), _p(_d(vol), C.c_double), _p(_d(su), C.c_double), _p(out, C.c_double))
    return out


def fvm_sp(vol, sp, diag):
    """fvm::Sp (fvmSup.C:100-170): returns diag + V*sp; sp a field or a number"""
    out = _d(diag).copy()
    if np.isscalar(sp):
        lib().orc_fvm_sp(C.c_int32(out.shape[0]), _p(_d(vol), C.c_double), None, C.c_double(sp), _p(out, C.c_double))
    else:
        lib().orc_fvm_sp(C.c_int32(out.shape[0]), _p(_d(vol), C.c_double), _p(_d(sp), C.c_double), C.c_double(0.0), _p(out, C.c_double))
    return out


def fvm_susp(vol, susp, vf, diag, source):
    """fvm::SuSp (fvmSup.C:190-214): returns (diag + V*max(susp,0), source - V*min(susp,0)*vf)"""
    d, s = _d(diag).copy(), _d(source).copy()
    lib().orc_fvm_susp(C.c_int32(d.shape[0]), _p(_d(vol), C.c_double), _p(_d(susp), C.c_double), _p(_d(vf), C.c_double), _p(d, C.c_double),
                       _p(s, C.c_double))
    return d, s


def flux_div(n_cells, lower_addr, upper_addr, lam, sf, v, scale=None, add_a=None, add_b=None, vol=None, want_div=True):
    """phi = Sf & interpolate([scale *] v) [+ add_a [* add_b]] on the internal faces and fvc::surfaceIntegrate(phi) [/ vol]"""
    lo, up = _i(lower_addr), _i(upper_addr)
    nf = lo.shape[0]
    phi = np.empty(nf)
    div = np.empty(n_cells) if want_div else None
    q = lambda a: _p(_d(a), C.c_double) if a is not None else None
    keep = [_d(a) for a in (lam, *sf, *v)]
    lib().orc_flux_div(C.c_int32(n_cells), C.c_int32(nf), _p(lo, C.c_int32), _p(up, C.c_int32), *[_p(a, C.c_double) for a in keep],
                       q(scale), q(add_a), q(add_b), _p(phi, C.c_double), q(vol), _p(div, C.c_double) if want_div else None)
    return (phi, div) if want_div else phi


def ddt_phi_corr(lower_addr, upper_addr, r_delta_t, lam, sf, u_old, rho_old, phi_old):
    """fvc::ddtCorr(rho, U, phi) on the internal faces (EulerDdtScheme.C:663-720; rho_old None: :523-551)"""
    lo, up = _i(lower_addr), _i(upper_addr)
    out = np.empty(lo.shape[0])
    keep = [_d(a) for a in (lam, *sf, *u_old)]
    lib().orc_ddt_phi_corr(C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), C.c_double(r_delta_t), *[_p(a, C.c_double) for a in keep],
                           _p(_d(rho_old), C.c_double) if rho_old is not None else None, _p(_d(phi_old), C.c_double), _p(out, C.c_double))
    return out


def upwind_weights(face_flux):
    f = _d(face_flux)
    w = np.empty_like(f)
    lib().orc_upwind_weights(C.c_int32(f.shape[0]), _p(f, C.c_double), _p(w, C.c_double))
    return w


def limited_linear_weights(lower_addr, upper_addr, k, cd_weights, face_flux, phi, grad, centres):
    lo, up = _i(lower_addr), _i(upper_addr)
    w, lim = np.empty(lo.shape[0]), np.empty(lo.shape[0])
    g = [_d(x) for x in grad]; c = [_d(x) for x in centres]
    lib().orc_limited_linear_weights(C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), C.c_double(k), _p(_d(cd_weights), C.c_double),
                                     _p(_d(face_flux), C.c_double), _p(_d(phi), C.c_double), _p(g[0], C.c_double), _p(g[1], C.c_double),
                                     _p(g[2], C.c_double), _p(c[0], C.c_double), _p(c[1], C.c_double), _p(c[2], C.c_double),
                                     _p(w, C.c_double), _p(lim, C.c_double))
    return w, lim


def gauss_grad(n_cells, lower_addr, upper_addr, sf, ssf, vol=None):
    lo, up = _i(lower_addr), _i(upper_addr)
    s = [_d(x) for x in sf]
    g = [np.empty(n_cells) for _ in range(3)]
    lib().orc_gauss_grad(C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(s[0], C.c_double),
                         _p(s[1], C.c_double), _p(s[2], C.c_double), _p(_d(ssf), C.c_double),
                         _p(_d(vol), C.c_double) if vol is not None else None, _p(g[0], C.c_double), _p(g[1], C.c_double), _p(g[2], C.c_double))
    return g


def axpby(a, x, b, y):
    xx, yy = _d(x), _d(y)
    out = np.empty_like(xx)
    lib().orc_axpby(C.c_int32(xx.shape[0]), C.c_double(a), _p(xx, C.c_double), C.c_double(b), _p(yy, C.c_double), _p(out, C.c_double))
    return out


# ---------------------------------------------------------------------------------------------
# oracle/_ref: the REFERENCE's own pairGAMGAgglomeration::agglomerate, compiled from /root/reference against a shim
# (oracle/ref_shim, oracle/Makefile target `ref`).  Present where the reference tree was available at build time.
# ---------------------------------------------------------------------------------------------
REF_PAIR_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_pair.so")


def ref_pair_available() -> bool:
    return os.path.exists(REF_PAIR_LIB)


def ref_pair_agglomerate(n_cells, lower_addr, upper_addr, face_weights, forward=True):
    """returns (coarseCellMap, nCoarseCells, forward_after) computed by the reference's code"""
    L = C.CDLL(REF_PAIR_LIB)
    lo, up, w = _i(lower_addr), _i(upper_addr), _d(face_weights)
    out = np.empty(n_cells, np.int32)
    fwd = C.c_int()
    nc = L.ref_pair_agglomerate(C.c_int(n_cells), C.c_int(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(w, C.c_double),
                                C.c_int(int(forward)), _p(out, C.c_int32), C.byref(fwd))
    return out, int(nc), bool(fwd.value)


REF_SOLVERS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_solvers.so")


def ref_solvers_available() -> bool:
    return os.path.exists(REF_SOLVERS_LIB)


def ref_krylov_solve(kind: str, system: "System", psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0,
                     n_sweeps=1, omega=0.9):
    """the REFERENCE's own PCG::solve / PBiCG::solve / PBiCGStab::solve (compiled from /root/reference against
    oracle/ref_shim/foam_solver_shim.H) driving this oracle's primitives; returns (psi, perf)"""
    lib()                                     # liboracle.so must be in the process before the dependent library
    L = C.CDLL(REF_SOLVERS_LIB)
    x = _d(psi).copy()
    b = _d(source)
    out = (C.c_double * 5)()
    L.ref_krylov_solve(C.c_int({"pcg": 0, "pbicg": 1, "pbicgstab": 2, "smooth": 3}[kind]), system.h, _p(x, C.c_double), _p(b, C.c_double),
                       C.c_int(PRECOND[precond]), C.c_double(tolerance), C.c_double(relTol), C.c_int(maxIter), C.c_int(minIter),
                       C.c_int(n_sweeps), C.c_double(omega), out)
    return x, dict(initialResidual=out[0], finalResidual=out[1], nIterations=int(out[2]), converged=bool(out[3]), singular=bool(out[4]))


REF_GAMG_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_gamg.so")


def ref_gamg_available() -> bool:
    return os.path.exists(REF_GAMG_LIB)


def ref_gamg_solve(hier: "GamgSysHierarchy", psi, source, **kw):
    """the REFERENCE's own GAMGSolver::solve / Vcycle / initVcycle / solveCoarsestLevel (GAMGSolverSolve.C compiled from
    /root/reference against oracle/ref_shim/foam_gamg_shim.H) on this oracle's hierarchy and primitives; returns (psi, perf)"""
    lib()
    L = C.CDLL(REF_GAMG_LIB)
    ctl = gamg_controls(**kw)
    x = _d(psi).copy()
    b = _d(source)
    out = (C.c_double * 5)()
    L.ref_gamg_solve(hier.h, hier.system.h, _p(x, C.c_double), _p(b, C.c_double), C.byref(ctl), out)
    return x, dict(initialResidual=out[0], finalResidual=out[1], nIterations=int(out[2]), converged=bool(out[3]), singular=bool(out[4]))


REF_FUNCTORS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_functors.so")


def ref_functors_available() -> bool:
    return os.path.exists(REF_FUNCTORS_LIB)


def _row_tables(case):
    n = case.n_cells
    lo, up = _i(case.lower_addr), _i(case.upper_addr)
    own_start = np.zeros(n + 1, np.int32); np.add.at(own_start, lo.astype(np.int64) + 1, 1); own_start = np.cumsum(own_start).astype(np.int32)
    los_start = np.zeros(n + 1, np.int32); np.add.at(los_start, up.astype(np.int64) + 1, 1); los_start = np.cumsum(los_start).astype(np.int32)
    losort = np.argsort(up, kind="stable").astype(np.int32)
    return lo, up, own_start, los_start, losort


def ref_jacobi_rows(case, omega, psi, b):
    """one Jacobi sweep computed row by row by the REFERENCE's JacobiSmootherFunctor<false,3> (JacobiSmootherF.H, host-compiled)"""
    L = C.CDLL(REF_FUNCTORS_LIB)
    lo, up, os_, ls_, losort = _row_tables(case)
    lower = _d(case.upper if case.lower is None else case.lower)
    out = np.empty(case.n_cells)
    L.ref_jacobi_rows(C.c_int(case.n_cells), C.c_double(omega), _p(_d(psi), C.c_double), _p(_d(case.diag), C.c_double), _p(_d(b), C.c_double),
                      _p(lower, C.c_double), _p(_d(case.upper), C.c_double), _p(lo, C.c_int32), _p(up, C.c_int32), _p(os_, C.c_int32),
                      _p(ls_, C.c_int32), _p(losort, C.c_int32), _p(out, C.c_double))
    return out


def ref_ainv_rows(case, r, transpose=False):
    """AINV preconditioner applied by the REFERENCE's AINVPreconditionerFunctor<false,3> (AINVPreconditionerF.H, host-compiled)"""
    L = C.CDLL(REF_FUNCTORS_LIB)
    lo, up, os_, ls_, losort = _row_tables(case)
    lower = _d(case.upper if case.lower is None else case.lower)
    upper = _d(case.upper)
    if transpose:       # preconditionT swaps the roles of the coefficient arrays (AINVPreconditioner.C:87-120)
        lower, upper = upper, lower
    rD = 1.0 / _d(case.diag)
    out = np.empty(case.n_cells)
    L.ref_ainv_rows(C.c_int(case.n_cells), _p(_d(r), C.c_double), _p(rD, C.c_double), _p(lower, C.c_double), _p(upper, C.c_double),
                    _p(lo, C.c_int32), _p(up, C.c_int32), _p(os_, C.c_int32), _p(ls_, C.c_int32), _p(losort, C.c_int32), _p(out, C.c_double))
    return out


REF_GAMG_FUNCTORS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_gamg_functors.so")


def ref_gamg_functors_available() -> bool:
    return os.path.exists(REF_GAMG_FUNCTORS_LIB)


def _sorted_segments(addressing):
    """createSort / createTarget (GAMGAgglomerateLduAddressing.C:37-120): stable sort of the restrict addressing, one
    segment per distinct target"""
    a = _i(addressing)
    sort = np.argsort(a, kind="stable").astype(np.int32)
    keys = a[sort]
    first = np.concatenate([[True], keys[1:] != keys[:-1]]) if a.shape[0] else np.zeros(0, bool)
    start = np.concatenate([np.nonzero(first)[0], [a.shape[0]]]).astype(np.int32)
    target = keys[first].astype(np.int32)
    return target, start, sort


def ref_gamg_restrict(restrict_map, n_coarse, ff):
    """GAMGAgglomeration::restrictField through the REFERENCE's GAMG::restrict functor (GAMGAgglomerationF.H, host-compiled)"""
    L = C.CDLL(REF_GAMG_FUNCTORS_LIB)
    target, start, sort = _sorted_segments(restrict_map)
    cf = np.zeros(n_coarse)
    L.ref_gamg_restrict(C.c_int(target.shape[0]), _p(target, C.c_int32), _p(start, C.c_int32), _p(sort, C.c_int32), _p(_d(ff), C.c_double), _p(cf, C.c_double))
    return cf


def ref_gamg_prolong(restrict_map, cf):
    """prolongField through the REFERENCE's GAMG::prolong functor"""
    L = C.CDLL(REF_GAMG_FUNCTORS_LIB)
    target, start, sort = _sorted_segments(restrict_map)
    ff = np.full(_i(restrict_map).shape[0], np.nan)
    L.ref_gamg_prolong(C.c_int(target.shape[0]), _p(target, C.c_int32), _p(start, C.c_int32), _p(sort, C.c_int32), _p(_d(cf), C.c_double), _p(ff, C.c_double))
    return ff


def ref_gamg_agglomerate_matrix(level, fine_diag, fine_upper, fine_lower=None):
    """GAMGSolver::agglomerateMatrix (deterministic path, GAMGSolverAgglomerateMatrix.C:65-72,218-317) through the
    REFERENCE's sym/asym/diag agglomerate functors (GAMGSolverAgglomerateMatrixF.H, host-compiled).  level: dict of
    GamgHierarchy.level()"""
    L = C.CDLL(REF_GAMG_FUNCTORS_LIB)
    asym = fine_lower is not None
    c_diag = ref_gamg_restrict(level["restrict"], level["n_coarse"], fine_diag)
    c_up = np.zeros(level["n_coarse_faces"]); c_lo = np.zeros(level["n_coarse_faces"])
    target, start, sort = _sorted_segments(level["face_restrict"])
    flip = np.ascontiguousarray(level["face_flip"], dtype=np.bool_)
    fl = _d(fine_lower if asym else fine_upper)
    L.ref_gamg_agglomerate_matrix(C.c_int(int(asym)), C.c_int(target.shape[0]), _p(target, C.c_int32), _p(start, C.c_int32), _p(sort, C.c_int32),
                                  _p(_d(fine_upper), C.c_double), _p(fl, C.c_double), flip.ctypes.data_as(C.c_void_p),
                                  _p(c_diag, C.c_double), _p(c_up, C.c_double), _p(c_lo, C.c_double))
    return c_diag, c_up, (c_lo if asym else None)


REF_GAMG_SCALE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_gamg_scale.so")


def ref_gamg_scale_available() -> bool:
    return os.path.exists(REF_GAMG_SCALE_LIB)


def gamg_sys_scale(system: "System", field, source):
    """the oracle's GAMGSolver::scale (gamg_oracle.c sys_scale): returns (scaled field, A*field before scaling)"""
    L = lib()
    f = np.array(_d(field), copy=True)
    acf = np.empty(system.n)
    L.orc_gamg_sys_scale.restype = None
    L.orc_gamg_sys_scale(system.h, _p(f, C.c_double), _p(acf, C.c_double), _p(_d(source), C.c_double))
    return f, acf


def ref_gamg_scale_pointwise(sf, field, source, acf, diag):
    """field = sf*field + (source - sf*Acf)/D through the REFERENCE's GAMGSolverScaleFunctor (GAMGSolverScale.C:36-55,
    compiled where it lies: oracle/ref_shim/ref_gamg_scale_tu.cpp)"""
    L = C.CDLL(REF_GAMG_SCALE_LIB)
    f = _d(field)
    out = np.empty(f.shape[0])
    L.ref_gamg_scale_pointwise(C.c_int(f.shape[0]), C.c_double(float(sf)), _p(f, C.c_double), _p(_d(source), C.c_double),
                               _p(_d(acf), C.c_double), _p(_d(diag), C.c_double), _p(out, C.c_double))
    return out


def ref_gamg_scale_terms(a, b):
    """the terms a*b of the two scaling sums through the REFERENCE's multiplyTupleFunctor (GAMGSolverScale.C:46-55)"""
    L = C.CDLL(REF_GAMG_SCALE_LIB)
    a = _d(a)
    out = np.empty(a.shape[0])
    L.ref_gamg_scale_terms(C.c_int(a.shape[0]), _p(a, C.c_double), _p(_d(b), C.c_double), _p(out, C.c_double))
    return out


REF_ATMUL_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_atmul.so")


def ref_atmul_available() -> bool:
    return os.path.exists(REF_ATMUL_LIB)


def ref_atmul(case, which, psi=None, source=None, favour_speed=0, level=0, coarsest=False):
    """The REFERENCE's lduMatrix::Amul / Tmul / residual / sumA / H1 (lduMatrixATmul.C compiled where it lies and run on the
    host, oracle/ref_shim/ref_atmul_tu.cpp) on a serial case.  which: "amul" | "tmul" | "residual" | "sumA" | "H1";
    favour_speed is lduMatrixSolutionCache::favourSpeed (0: losort-indirect path, 1-2: the pre-sorted fast paths)."""
    L = C.CDLL(REF_ATMUL_LIB)
    lo, up = _i(case.lower_addr), _i(case.upper_addr)
    n, nf = case.n_cells, lo.shape[0]
    losort = np.argsort(up, kind="stable").astype(np.int32)                       # lduAddressing::calcLosort
    owner_start = np.searchsorted(lo, np.arange(n + 1)).astype(np.int32)          # calcOwnerStart
    losort_start = np.searchsorted(up[losort], np.arange(n + 1)).astype(np.int32)  # calcLosortStart
    owner_sort = np.ascontiguousarray(lo[losort])                                 # ownerSortAddr
    upper = _d(case.upper)
    lower = upper if case.lower is None else _d(case.lower)
    lower_sort, upper_sort = np.ascontiguousarray(lower[losort]), np.ascontiguousarray(upper[losort])   # lduMatrix::lowerSort()
    x = np.zeros(n) if psi is None else _d(psi)
    b = np.zeros(n) if source is None else _d(source)
    out = np.full(n, np.nan)
    k = {"amul": 0, "tmul": 1, "residual": 2, "sumA": 3, "H1": 4, "ainv": 5, "ainvT": 6}[which]   # 5, 6: AINVPreconditioner.C, psi = r
    L.ref_atmul(C.c_int(k), C.c_int(favour_speed), C.c_int(level), C.c_int(int(coarsest)), C.c_int(n), C.c_int(nf), _p(lo, C.c_int32), _p(up, C.c_int32),
                _p(owner_sort, C.c_int32), _p(owner_start, C.c_int32), _p(losort_start, C.c_int32), _p(losort, C.c_int32), _p(_d(case.diag), C.c_double),
                _p(lower, C.c_double), _p(upper, C.c_double), _p(lower_sort, C.c_double), _p(upper_sort, C.c_double), _p(x, C.c_double), _p(b, C.c_double),
                _p(out, C.c_double))
    return out


REF_AMI_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_ami.so")


def ref_ami_available() -> bool:
    return os.path.exists(REF_AMI_LIB)


def ref_ami_interpolate(start, address, weights, fld, low_weight_correction=-1.0, weights_sum=None, default_values=None):
    """The REFERENCE's AMIInterpolationF.H functors with its own plusEqOp / multiplyWeightedOp (oracle/ref_shim/ref_ami_tu.cpp)"""
    L = C.CDLL(REF_AMI_LIB)
    st, ad = _i(start), _i(address)
    n = st.shape[0] - 1
    out = np.full(n, np.nan)
    L.ref_ami_interpolate(C.c_int(n), _p(st, C.c_int32), _p(ad, C.c_int32), _p(_d(weights), C.c_double), _p(_d(fld), C.c_double),
                          C.c_double(low_weight_correction), None if weights_sum is None else _p(_d(weights_sum), C.c_double),
                          None if default_values is None else _p(_d(default_values), C.c_double), _p(out, C.c_double))
    return out


REF_LDU_OPS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_ldu_ops.so")


def ref_ldu_ops_available() -> bool:
    return os.path.exists(REF_LDU_OPS_LIB)


def ref_ldu_ops(case, which, vec=None, favour_speed=0):
    """The REFERENCE's lduMatrixOperations.C (compiled where it lies and run on the host, oracle/ref_shim/ref_ldu_ops_tu.cpp) on a
    serial case.  which: "sumDiag" | "negSumDiag" -> new diag; "sumMagOffDiag" | "H" (vec = psi) -> field; "scale" (operator*=(vec))
    and "addNegate" (B = A; B *= 0.5; A += B; A.negate()) -> (diag, upper, lower)"""
    L = C.CDLL(REF_LDU_OPS_LIB)
    lo, up = _i(case.lower_addr), _i(case.upper_addr)
    n, nf = case.n_cells, lo.shape[0]
    losort = np.argsort(up, kind="stable").astype(np.int32)
    owner_start = np.searchsorted(lo, np.arange(n + 1)).astype(np.int32)
    losort_start = np.searchsorted(up[losort], np.arange(n + 1)).astype(np.int32)
    owner_sort = np.ascontiguousarray(lo[losort])
    x = np.zeros(n) if vec is None else _d(vec)
    out, ou, ol = np.full(n, np.nan), np.full(nf, np.nan), np.full(nf, np.nan)
    k = {"sumDiag": 0, "negSumDiag": 1, "sumMagOffDiag": 2, "H": 3, "scale": 4, "addNegate": 5}[which]
    L.ref_ldu_ops(C.c_int(k), C.c_int(favour_speed), C.c_int(n), C.c_int(nf), _p(lo, C.c_int32), _p(up, C.c_int32), _p(owner_sort, C.c_int32),
                  _p(owner_start, C.c_int32), _p(losort_start, C.c_int32), _p(losort, C.c_int32), _p(_d(case.diag), C.c_double),
                  None if case.lower is None else _p(_d(case.lower), C.c_double), _p(_d(case.upper), C.c_double), _p(x, C.c_double),
                  _p(out, C.c_double), _p(ou, C.c_double), _p(ol, C.c_double))
    return (out, ou, ol) if k >= 4 else out


def sngrad_correction_flux(lower_addr, upper_addr, corr_vecs, weights, grad, gamma_magsf=None):
    """flux of the non-orthogonal correction on the internal faces (gaussLaplacianSchemes.C:64-90, correctedSnGrad.C:45-65)"""
    lo, up = _i(lower_addr), _i(upper_addr)
    cv = [_d(x) for x in corr_vecs]; g = [_d(x) for x in grad]
    out = np.empty(lo.shape[0])
    lib().orc_sngrad_correction_flux(C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(cv[0], C.c_double), _p(cv[1], C.c_double),
                                     _p(cv[2], C.c_double), _p(_d(weights), C.c_double), _p(g[0], C.c_double), _p(g[1], C.c_double), _p(g[2], C.c_double),
                                     _p(_d(gamma_magsf), C.c_double) if gamma_magsf is not None else None, _p(out, C.c_double))
    return out


def patch_sngrad_correction_flux(face_cells, corr_vecs, weights, grad, nbr_grad, gamma_magsf=None):
    fc = _i(face_cells)
    cv = [_d(x) for x in corr_vecs]; g = [_d(x) for x in grad]; nb = [_d(x) for x in nbr_grad]
    out = np.empty(fc.shape[0])
    lib().orc_patch_sngrad_correction_flux(C.c_int32(fc.shape[0]), _p(fc, C.c_int32), _p(cv[0], C.c_double), _p(cv[1], C.c_double), _p(cv[2], C.c_double),
                                           _p(_d(weights), C.c_double), _p(g[0], C.c_double), _p(g[1], C.c_double), _p(g[2], C.c_double),
                                           _p(nb[0], C.c_double), _p(nb[1], C.c_double), _p(nb[2], C.c_double),
                                           _p(_d(gamma_magsf), C.c_double) if gamma_magsf is not None else None, _p(out, C.c_double))
    return out


def submul(x, y, inout):
    io = _d(inout).copy()
    lib().orc_submul(C.c_int32(io.shape[0]), _p(_d(x), C.c_double), _p(_d(y), C.c_double), _p(io, C.c_double))
    return io



def set_reference(celli, value, diag, source):
    """fvMatrix::setReference (fvMatrix.C:964-981) -> (diag, source)"""
    d, s = _d(diag).copy(), _d(source).copy()
    lib().orc_set_reference(C.c_int32(celli), C.c_double(value), _p(d, C.c_double), _p(s, C.c_double))
    return d, s


def set_values(n_cells, lower_addr, upper_addr, cell_labels, values, psi, diag, source, upper, lower=None, face_cells=(), icoeffs=(), bcoeffs=(),
               upstream=False):
    """fvMatrix::setValues (fvMatrix.C:454-656) -> dict(psi, source, upper, lower, icoeffs, bcoeffs)"""
    lo, up = _i(lower_addr), _i(upper_addr)
    cl, v = _i(cell_labels), _d(values)
    ps, s = _d(psi).copy(), _d(source).copy()
    U = _d(upper); Lw = None if lower is None else _d(lower)
    uo, lw = np.empty(lo.shape[0]), np.empty(lo.shape[0])
    n = len(face_cells)
    fcs = [_i(f) for f in face_cells]
    ics = [_d(a).copy() for a in icoeffs]
    bcs = [_d(a).copy() for a in bcoeffs]
    sizes = (C.c_int32 * max(n, 1))(*[f.shape[0] for f in fcs])
    fp = (C.POINTER(C.c_int32) * max(n, 1))(*[_p(f, C.c_int32) for f in fcs])
    ip = (C.POINTER(C.c_double) * max(n, 1))(*[_p(a, C.c_double) for a in ics])
    bp = (C.POINTER(C.c_double) * max(n, 1))(*[_p(a, C.c_double) for a in bcs])
    lib().orc_set_values(C.c_int32(n_cells), C.c_int32(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), C.c_int32(cl.shape[0]), _p(cl, C.c_int32),
                         _p(v, C.c_double), C.c_int(int(upstream)), _p(ps, C.c_double), _p(_d(diag), C.c_double), _p(s, C.c_double), _p(U, C.c_double),
                         None if Lw is None else _p(Lw, C.c_double), _p(uo, C.c_double), _p(lw, C.c_double), n, sizes, fp, ip, bp)
    return dict(psi=ps, source=s, upper=uo, lower=lw, icoeffs=ics, bcoeffs=bcs)


# ---------------------------------------------------------------------------------------------
# oracle/_ref/libref_fvm.so: the REFERENCE's fvMatrix-assembly functors (fvMatrix.C, fvcSurfaceIntegrate.C, gaussGrad.C,
# surfaceInterpolationScheme.C, limitedSurfaceInterpolationScheme.C, LimitedScheme.C + NVDTVD.H / limitedLinear.H,
# lduMatrixTemplates.C) on the reference's own primitives, host-compiled (oracle/ref_shim/ref_fvm_tu.cpp)
# ---------------------------------------------------------------------------------------------
REF_FVM_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_fvm.so")
REF_FVM_PATCH_KINDS = {"add": 0, "subtract": 1, "boundarySource": 2, "relaxComponentZero": 3, "relaxMagComponentZero": 4,
                       "relaxMaxComponentMag": 5, "relaxNegComponentZero": 6, "relaxNegComponentMin": 7, "surfaceIntegratePatch": 8}


def ref_fvm_available() -> bool:
    return os.path.exists(REF_FVM_LIB)


def patch_sort_tables(face_cells):
    """lduAddressing::patchSortCells / patchSortAddr / patchSortStartAddr (lduAddressing.C:373-400): the patch's unique cells ascending,
    its faces stably sorted by cell, the start of every cell's segment"""
    fc = _i(face_cells)
    sort = np.argsort(fc, kind="stable").astype(np.int32)
    cells, counts = np.unique(fc, return_counts=True)
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return np.ascontiguousarray(cells.astype(np.int32)), start, sort


def ref_fvm_patch_rows(kind, face_cells, pf, field, q=None):
    """field[cell] = F(field[cell], patch faces of the cell in ascending order) for every unique patch cell, F one of the reference's
    patch functors (REF_FVM_PATCH_KINDS)"""
    L = C.CDLL(REF_FVM_LIB)
    cells, start, sort = patch_sort_tables(face_cells)
    out = _d(field).copy()
    qq = None if q is None else _d(q)
    L.ref_fvm_patch_rows(C.c_int(REF_FVM_PATCH_KINDS[kind]), C.c_int(cells.shape[0]), _p(cells, C.c_int32), _p(start, C.c_int32), _p(sort, C.c_int32),
                         _p(_d(pf), C.c_double), None if qq is None else _p(qq, C.c_double), _p(out, C.c_double))
    return out


def ref_fvm_relax_dominance(diag, sum_off):
    L = C.CDLL(REF_FVM_LIB)
    out = _d(diag).copy()
    L.ref_fvm_relax_dominance(C.c_int(out.shape[0]), _p(_d(sum_off), C.c_double), _p(out, C.c_double))
    return out


def _ldu_row_tables(n, lower_addr, upper_addr):
    lo, up = _i(lower_addr), _i(upper_addr)
    losort = np.argsort(up, kind="stable").astype(np.int32)
    own_start = np.searchsorted(lo, np.arange(n + 1)).astype(np.int32)
    los_start = np.searchsorted(up[losort], np.arange(n + 1)).astype(np.int32)
    return lo, up, own_start, los_start, losort


def ref_surface_integrate_rows(n_cells, lower_addr, upper_addr, ssf, integrate=True):
    L = C.CDLL(REF_FVM_LIB)
    lo, up, os_, ls_, losort = _ldu_row_tables(n_cells, lower_addr, upper_addr)
    out = np.empty(n_cells)
    L.ref_surface_integrate_rows(C.c_int(int(integrate)), C.c_int(n_cells), _p(_d(ssf), C.c_double), _p(os_, C.c_int32), _p(ls_, C.c_int32),
                                 _p(lo, C.c_int32), _p(up, C.c_int32), _p(losort, C.c_int32), _p(out, C.c_double))
    return out


def ref_face_interpolate(lower_addr, upper_addr, lam, vf):
    """scalar vf[N] -> [F]; vector vf[N, 3] -> [F, 3]"""
    L = C.CDLL(REF_FVM_LIB)
    lo, up = _i(lower_addr), _i(upper_addr)
    v = _d(vf)
    out = np.empty((lo.shape[0],) + v.shape[1:])
    fn = L.ref_face_interpolate if v.ndim == 1 else L.ref_face_interpolate_vector
    fn(C.c_int(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(_d(lam), C.c_double), _p(v, C.c_double), _p(out, C.c_double))
    return out


def ref_face_dot(a3, b3):
    L = C.CDLL(REF_FVM_LIB)
    a, b = _d(a3), _d(b3)
    out = np.empty(a.shape[0])
    L.ref_face_dot(C.c_int(a.shape[0]), _p(a, C.c_double), _p(b, C.c_double), _p(out, C.c_double))
    return out


def ref_gauss_grad_rows(n_cells, lower_addr, upper_addr, sf3, ssf):
    L = C.CDLL(REF_FVM_LIB)
    lo, up, os_, ls_, losort = _ldu_row_tables(n_cells, lower_addr, upper_addr)
    out = np.empty((n_cells, 3))
    L.ref_gauss_grad_rows(C.c_int(n_cells), _p(_d(sf3), C.c_double), _p(_d(ssf), C.c_double), _p(os_, C.c_int32), _p(ls_, C.c_int32),
                          _p(lo, C.c_int32), _p(up, C.c_int32), _p(losort, C.c_int32), _p(out, C.c_double))
    return out


def ref_gauss_grad_patch_rows(face_cells, psf3, pssf, grad3):
    L = C.CDLL(REF_FVM_LIB)
    cells, start, sort = patch_sort_tables(face_cells)
    out = _d(grad3).copy()
    L.ref_gauss_grad_patch_rows(C.c_int(cells.shape[0]), _p(cells, C.c_int32), _p(start, C.c_int32), _p(sort, C.c_int32), _p(_d(psf3), C.c_double),
                                _p(_d(pssf), C.c_double), _p(out, C.c_double))
    return out


def ref_faceH(lower_addr, upper_addr, lower, upper, psi):
    L = C.CDLL(REF_FVM_LIB)
    lo, up = _i(lower_addr), _i(upper_addr)
    out = np.empty(lo.shape[0])
    L.ref_faceH(C.c_int(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(_d(lower), C.c_double), _p(_d(upper), C.c_double), _p(_d(psi), C.c_double),
                _p(out, C.c_double))
    return out


def ref_limited_linear(lower_addr, upper_addr, k, cd_weights, face_flux, phi, grad3, centres3):
    """-> (limiter, weights) of limitedLinear(k) on the internal faces"""
    L = C.CDLL(REF_FVM_LIB)
    lo, up = _i(lower_addr), _i(upper_addr)
    lim, w = np.empty(lo.shape[0]), np.empty(lo.shape[0])
    L.ref_limited_linear(C.c_int(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), C.c_double(k), _p(_d(cd_weights), C.c_double), _p(_d(face_flux), C.c_double),
                         _p(_d(phi), C.c_double), _p(_d(grad3), C.c_double), _p(_d(centres3), C.c_double), _p(lim, C.c_double), _p(w, C.c_double))
    return lim, w


def ref_set_values_source(n_cells, lower_addr, upper_addr, cell_mask, cell_values, upper, lower, source):
    """fvMatrix::setValuesFromList's row functor + clear-faces functor -> (source, upper, lower | None)"""
    L = C.CDLL(REF_FVM_LIB)
    lo, up, os_, ls_, losort = _ldu_row_tables(n_cells, lower_addr, upper_addr)
    s = _d(source).copy()
    U = _d(upper); Lw = U if lower is None else _d(lower)
    uo = np.empty(lo.shape[0]); lw = None if lower is None else np.empty(lo.shape[0])
    m = np.ascontiguousarray(cell_mask, dtype=np.uint8)
    L.ref_set_values_source(C.c_int(n_cells), C.c_int(lo.shape[0]), _p(lo, C.c_int32), _p(up, C.c_int32), _p(os_, C.c_int32), _p(ls_, C.c_int32),
                            _p(losort, C.c_int32), _p(m, C.c_uint8), _p(_d(cell_values), C.c_double), _p(U, C.c_double), _p(Lw, C.c_double),
                            _p(s, C.c_double), _p(uo, C.c_double), None if lw is None else _p(lw, C.c_double))
    return s, uo, lw
