"""ctypes binding of the C ABI (include/mi_ldu.h) plus a thin object layer.

PyTorch is only plumbing here: device memory (``torch.Tensor.data_ptr()``),
the current HIP stream and ``torch.distributed``.  All arithmetic happens in
``librapidcfd_amd.so`` (hand-written gfx950 kernels).  There is no CPU
fallback: constructing a :class:`Context` without a gfx950 device raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librapidcfd_amd.so")
if os.environ.get("MI_ENGINE_LIB"):   # A/B builds of the engine (tools/ab_dma.sh); the product path is the in-tree library above
    LIB_PATH = os.environ["MI_ENGINE_LIB"]
_lib = None

PRECOND = {"none": 0, "diagonal": 1, "AINV": 2,
           # lduMatrixPreconditioner.C:58-61, DICPreconditioner.C:42-58: DIC/DILU resolve to AINV
           "DIC": 2, "DILU": 2, "FDIC": 2}


class MiError(RuntimeError):
    pass


class SolverPerf(C.Structure):
    _fields_ = [("initialResidual", C.c_double), ("finalResidual", C.c_double),
                ("normFactor", C.c_double), ("nIterations", C.c_int32),
                ("converged", C.c_int32), ("singular", C.c_int32), ("reserved", C.c_int32)]


class SolverControls(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("relTol", C.c_double),
                ("maxIter", C.c_int32), ("minIter", C.c_int32)]


# every exported symbol of include/mi_ldu.h (tests check the library exports all of them)
SYMBOLS = [
    "mi_addr_set_ami_patch", "mi_addr_set_ami_patch_remote", "mi_addr_set_ami_face_areas", "mi_matrix_set_patch_transform",
    "mi_comm_peer_window", "mi_comm_peer_connect", "mi_comm_peer_status", "mi_gamg_create_dummy", "mi_gamg_host_build_ami",
    "mi_addr_create_adopted", "mi_layout_adopt_host", "mi_pbicg_solve_multi", "mi_comm_peer_auto", "mi_comm_peer_selftest", "mi_comm_peer_enable", "mi_matrix_peer_halo_auto", "mi_matrix_peer_halo_status",
    "mi_pcg_iterate_sampled", "mi_addr_set_ami_patch_remote_multi", "mi_fvm_assemble", "mi_fvm_set_reference", "mi_fvm_set_values", "mi_relax_multi", "mi_fvc_div",
    "mi_fvm_ddt_euler", "mi_fvm_ddt_euler_rho", "mi_fvm_su", "mi_fvm_sp", "mi_fvm_susp", "mi_flux_div", "mi_ddt_phi_corr", "mi_upwind_weights", "mi_limited_linear_weights", "mi_gauss_grad", "mi_vec_axpby", "mi_vec_div",
    "mi_comm_unique_id", "mi_comm_create", "mi_comm_destroy", "mi_comm_allreduce_sum", "mi_dpcg_comm_begin",
    "mi_dpcg_comm_iterate", "mi_gamg_create_coupled", "mi_matrix_attach_comm", "mi_matrix_detach_comm", "mi_matrix_patch_neighbour_field",
    "mi_ctx_create", "mi_ctx_destroy", "mi_ctx_synchronize", "mi_ctx_stat", "mi_ctx_set_option", "mi_last_error", "mi_device_available",
    "mi_addr_create", "mi_addr_create_coupled", "mi_addr_destroy", "mi_addr_n_cells", "mi_addr_n_faces", "mi_addr_n_tiles",
    "mi_addr_n_ext", "mi_addr_cell_perm", "mi_addr_stats", "mi_addr_patch_offsets",
    "mi_matrix_create", "mi_matrix_addr", "mi_matrix_destroy", "mi_matrix_set_coeffs", "mi_matrix_set_interface_coeffs",
    "mi_matrix_set_ext", "mi_halo_pack_engine", "mi_vec_to_engine", "mi_vec_from_engine",
    "mi_amul", "mi_tmul", "mi_sumA", "mi_residual", "mi_H", "mi_H1", "mi_faceH",
    "mi_amul_engine", "mi_tmul_engine", "mi_precondition", "mi_jacobi_smooth",
    "mi_sum", "mi_sum_prod", "mi_sum_mag", "mi_norm_factor", "mi_norm_factor_engine", "mi_precondition_engine",
    "mi_residual_engine", "mi_jacobi_smooth_engine",
    "mi_pcg_solve", "mi_pcg_begin", "mi_pcg_iterate", "mi_pcg_end",
    "mi_pbicg_solve", "mi_pbicgstab_solve", "mi_smooth_solve",
    "mi_bench_amul", "mi_bench_pcg_iters", "mi_debug_occupancy", "mi_debug_dense_invert",
    "mi_layout_build_host", "mi_layout_array", "mi_layout_free", "mi_layout_build_host_given", "mi_layout_inherit_tiles",
    "mi_dpcg_set_buffers", "mi_dpcg_phase", "mi_dpcg_status", "mi_event_record", "mi_event_elapsed_ms",
    "mi_gamg_create", "mi_gamg_update", "mi_gamg_level_matrix", "mi_gamg_scale", "mi_gamg_solve_coarsest", "mi_gamg_destroy", "mi_gamg_n_levels", "mi_gamg_forward_out", "mi_gamg_level_sizes",
    "mi_gamg_solve", "mi_gamg_restrict", "mi_gamg_prolong", "mi_gamg_level_coeffs",
    "mi_gamg_host_build", "mi_gamg_host_build_domains", "mi_gamg_host_patch_array", "mi_gamg_host_n_levels", "mi_gamg_host_array", "mi_gamg_host_free",
    "mi_row_face_op", "mi_fvm_laplacian", "mi_fvm_div", "mi_surface_integrate", "mi_face_interpolate",
    "mi_patch_create", "mi_patch_destroy", "mi_patch_add", "mi_patch_add_product", "mi_patch_flux", "mi_relax",
    "mi_sngrad_correction_flux", "mi_patch_sngrad_correction_flux", "mi_patch_internal_field", "mi_vec_submul",
    "mi_comm_create_external", "mi_addr_create_ordered", "mi_addr_tile_starts", "mi_addr_is_ordered",
]


class GamgControls(C.Structure):
    _fields_ = [("tolerance", C.c_double), ("relTol", C.c_double), ("maxIter", C.c_int32), ("minIter", C.c_int32),
                ("nPreSweeps", C.c_int32), ("preSweepsLevelMultiplier", C.c_int32), ("maxPreSweeps", C.c_int32),
                ("nPostSweeps", C.c_int32), ("postSweepsLevelMultiplier", C.c_int32), ("maxPostSweeps", C.c_int32),
                ("nFinestSweeps", C.c_int32), ("scaleCorrection", C.c_int32), ("omega", C.c_double),
                ("directSolveCoarsest", C.c_int32), ("reserved", C.c_int32)]


def gamg_controls(tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0, nPreSweeps=0, preSweepsLevelMultiplier=1,
                  maxPreSweeps=4, nPostSweeps=2, postSweepsLevelMultiplier=1, maxPostSweeps=4, nFinestSweeps=2,
                  scaleCorrection=-1, omega=0.9, directSolveCoarsest=True):
    """GAMGSolver.C:67-77 defaults"""
    return GamgControls(tolerance, relTol, maxIter, minIter, nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps,
                        nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps, nFinestSweeps, scaleCorrection, omega,
                        int(bool(directSolveCoarsest)), 0)


def adopt_host(n_cells, lower_addr, upper_addr, patch_face_cells=(), patch_nbr_cells=()):
    """host part of renumber-at-bind (mi_layout_adopt_host): dict with cell_map, face_map, face_flipped, lower, upper, n_tiles"""
    lo = np.ascontiguousarray(lower_addr, dtype=np.int32); up = np.ascontiguousarray(upper_addr, dtype=np.int32)
    patches = [np.ascontiguousarray(p, dtype=np.int32) for p in patch_face_cells]
    npatch = len(patches)
    I32, U8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    sizes = (C.c_int32 * max(npatch, 1))(*[p.shape[0] for p in patches])
    ptrs = (I32 * max(npatch, 1))(*[p.ctypes.data_as(I32) for p in patches])
    nbrs = [None if (k >= len(patch_nbr_cells) or patch_nbr_cells[k] is None) else np.ascontiguousarray(patch_nbr_cells[k], dtype=np.int32) for k in range(npatch)]
    nptrs = (I32 * max(npatch, 1))(*[q.ctypes.data_as(I32) if q is not None else I32() for q in nbrs])
    out = dict(cell_map=np.empty(n_cells, dtype=np.int32), face_map=np.empty(lo.shape[0], dtype=np.int32), face_flipped=np.empty(lo.shape[0], dtype=np.uint8),
               lower=np.empty(lo.shape[0], dtype=np.int32), upper=np.empty(lo.shape[0], dtype=np.int32))
    nt = C.c_int32(0)
    _chk(lib().mi_layout_adopt_host(C.c_int32(n_cells), C.c_int32(lo.shape[0]), lo.ctypes.data_as(I32), up.ctypes.data_as(I32), C.c_int32(npatch), sizes, ptrs, nptrs,
                                    out["cell_map"].ctypes.data_as(I32), out["face_map"].ctypes.data_as(I32), out["face_flipped"].ctypes.data_as(U8),
                                    out["lower"].ctypes.data_as(I32), out["upper"].ctypes.data_as(I32), C.byref(nt)))
    out["n_tiles"] = int(nt.value)
    return out


def host_layout(n_cells, lower_addr, upper_addr, patch_face_cells=(), tile_cells=0, slot_cap=0, patch_nbr_cells=()) -> dict:
    """Build the tiled layout on the host only and return its tables as numpy arrays (tests)."""
    lo = np.ascontiguousarray(lower_addr, dtype=np.int32)
    up = np.ascontiguousarray(upper_addr, dtype=np.int32)
    patches = [np.ascontiguousarray(p, dtype=np.int32) for p in patch_face_cells]
    npatch = len(patches)
    sizes = (C.c_int32 * max(npatch, 1))(*[p.shape[0] for p in patches])
    ptrs = (C.POINTER(C.c_int32) * max(npatch, 1))(*[p.ctypes.data_as(C.POINTER(C.c_int32)) for p in patches])
    h = C.c_void_p()
    nbrs = [None if (k >= len(patch_nbr_cells) or patch_nbr_cells[k] is None)
            else np.ascontiguousarray(patch_nbr_cells[k], dtype=np.int32) for k in range(npatch)]
    nptrs = (C.POINTER(C.c_int32) * max(npatch, 1))(*[q.ctypes.data_as(C.POINTER(C.c_int32)) if q is not None
                                                       else C.POINTER(C.c_int32)() for q in nbrs])
    _chk(lib().mi_layout_build_host(C.c_int32(n_cells), C.c_int32(lo.shape[0]),
                                    lo.ctypes.data_as(C.POINTER(C.c_int32)), up.ctypes.data_as(C.POINTER(C.c_int32)),
                                    C.c_int32(npatch), sizes, ptrs, nptrs, C.c_int32(tile_cells), C.c_int32(slot_cap), C.byref(h)))
    out = {}
    try:
        for name in ("e2c", "c2e", "tileCellStart", "tileSlotStart", "tileIfaceSlot0", "tileHaloStart", "haloCell",
                     "tileSliceStart", "sliceEntryStart", "entries", "slotFace", "extSlot", "interiorTiles", "boundaryTiles",
                     "patchOffset", "patchFaceCellsE", "faceSlot", "sliceEntryStart16", "entries16", "slotBase", "tileSbStart"):
            data, ln = C.c_void_p(), C.c_int64()
            _chk(lib().mi_layout_array(h, name.encode(), C.byref(data), C.byref(ln)))
            dt = np.uint32 if name in ("entries", "entries16") else (np.uint16 if name == "slotBase" else np.int32)
            if ln.value:
                buf = (C.c_char * (ln.value * np.dtype(dt).itemsize)).from_address(data.value)
                out[name] = np.frombuffer(buf, dtype=dt).copy()
            else:
                out[name] = np.zeros(0, dtype=dt)
    finally:
        lib().mi_layout_free(h)
    return out


_LAYOUT_ARRAYS = ("e2c", "c2e", "tileCellStart", "tileSlotStart", "tileIfaceSlot0", "tileHaloStart", "haloCell", "tileSliceStart", "sliceEntryStart",
                  "entries", "slotFace", "extSlot", "interiorTiles", "boundaryTiles", "patchOffset", "patchFaceCellsE", "faceSlot")


def host_layout_given(n_cells, lower_addr, upper_addr, part, n_parts) -> dict:
    """Host-only layout of a GIVEN partition (mi_layout_build_host_given; tests)."""
    I32 = C.POINTER(C.c_int32)
    lo = np.ascontiguousarray(lower_addr, dtype=np.int32); up = np.ascontiguousarray(upper_addr, dtype=np.int32)
    pt = np.ascontiguousarray(part, dtype=np.int32)
    h = C.c_void_p()
    _chk(lib().mi_layout_build_host_given(C.c_int32(n_cells), C.c_int32(lo.shape[0]), lo.ctypes.data_as(I32), up.ctypes.data_as(I32), C.c_int32(n_parts), pt.ctypes.data_as(I32), C.byref(h)))
    out = {}
    try:
        for name in _LAYOUT_ARRAYS:
            data, ln = C.c_void_p(), C.c_int64()
            _chk(lib().mi_layout_array(h, name.encode(), C.byref(data), C.byref(ln)))
            dt = np.uint32 if name == "entries" else np.int32
            out[name] = np.frombuffer((C.c_char * (ln.value * 4)).from_address(data.value), dtype=dt).copy() if ln.value else np.zeros(0, dtype=dt)
    finally:
        lib().mi_layout_free(h)
    return out


def inherit_tiles(restrict_map, fine_tile_of_cell, n_fine_tiles, n_coarse, c_lower, c_upper, cell_cap=0, slot_cap=0):
    """Tiles of a coarse GAMG level inherited from its fine level's tiles (mi_layout_inherit_tiles; tests): (part, n_parts)."""
    I32 = C.POINTER(C.c_int32)
    rm = np.ascontiguousarray(restrict_map, dtype=np.int32); ft = np.ascontiguousarray(fine_tile_of_cell, dtype=np.int32)
    cl = np.ascontiguousarray(c_lower, dtype=np.int32); cu = np.ascontiguousarray(c_upper, dtype=np.int32)
    part = np.empty(n_coarse, dtype=np.int32); n_parts = C.c_int32()
    _chk(lib().mi_layout_inherit_tiles(C.c_int32(rm.shape[0]), rm.ctypes.data_as(I32), ft.ctypes.data_as(I32), C.c_int32(n_fine_tiles), C.c_int32(n_coarse),
                                       C.c_int32(cl.shape[0]), cl.ctypes.data_as(I32), cu.ctypes.data_as(I32), C.c_int32(cell_cap), C.c_int32(slot_cap),
                                       part.ctypes.data_as(I32), C.byref(n_parts)))
    return part, int(n_parts.value)


def lib():
    """Load the native library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch (the plumbing for device memory / streams here) preloads its bundled HIP runtime by path.  If the engine
        # were loaded first it would bind /opt/rocm's copy, torch would then map a second runtime, and the second one
        # cannot open the device ("no ROCm-capable device").  Loading torch first makes both share one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:  # pure C-ABI use without PyTorch
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.mi_last_error.restype = C.c_char_p
    return _lib


def _chk(rc: int):
    if rc != 0:
        raise MiError(f"mi error {rc}: {lib().mi_last_error().decode()}")


def device_available() -> bool:
    return bool(lib().mi_device_available())


def _ptr(t) -> C.c_void_p:
    """Device pointer of a contiguous float64 torch tensor (or None)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous() and t.dtype.is_floating_point and t.element_size() == 8, "need contiguous float64"
    return C.c_void_p(t.data_ptr())


class Context:
    def __init__(self, device: int = 0, stream_handle: Optional[int] = None):
        self.h = C.c_void_p()
        _chk(lib().mi_ctx_create(int(device), C.c_void_p(stream_handle or 0), C.byref(self.h)))
        self.device = device

    def synchronize(self):
        _chk(lib().mi_ctx_synchronize(self.h))

    def set_option(self, name: str, value: int):
        """mi_ctx_set_option: "pcg_persist", "pcg_fuse_rp", "win_direct", "gamg_graph_attached" 0 / 1"""
        _chk(lib().mi_ctx_set_option(self.h, name.encode(), C.c_int32(int(value))))

    def dense_invert(self, a_dev, inv_dev, n: int, which: int) -> bool:
        """diagnostic: invert the row-major n x n matrix a_dev into inv_dev with path `which` (mi_debug_dense_invert); False: singular"""
        sing = C.c_int32()
        _chk(lib().mi_debug_dense_invert(self.h, _ptr(a_dev), C.c_int32(n), _ptr(inv_dev), C.c_int32(which), C.byref(sing)))
        return sing.value == 0

    def stat(self, which: int) -> int:
        """mi_ctx_stat: 0 = launches of the persistent PCG kernel on plain matrices, 1 = on communicator-attached ones,
        2 = grid-barrier litmus runs, 3 = V-cycles of a decomposed case replayed as a hipGraph, 4 = launches of the fused residual / direction kernel of PCG"""
        v = C.c_int64(0)
        _chk(lib().mi_ctx_stat(self.h, C.c_int32(which), C.byref(v)))
        return int(v.value)

    def close(self):
        if self.h:
            lib().mi_ctx_destroy(self.h)
            self.h = C.c_void_p()

    # field reductions -------------------------------------------------------
    def _red(self, fn, a, b=None):
        out = C.c_double()
        if b is None:
            _chk(getattr(lib(), fn)(self.h, _ptr(a), C.c_int64(a.numel()), C.byref(out)))
        else:
            _chk(getattr(lib(), fn)(self.h, _ptr(a), _ptr(b), C.c_int64(a.numel()), C.byref(out)))
        return out.value

    def sum(self, a):
        return self._red("mi_sum", a)

    def sum_mag(self, a):
        return self._red("mi_sum_mag", a)

    def sum_prod(self, a, b):
        return self._red("mi_sum_prod", a, b)


class Addressing:
    """lduAddressing: host lowerAddr/upperAddr + coupled-patch faceCells -> tiled engine layout."""

    def __init__(self, ctx: Context, n_cells: int, lower_addr, upper_addr, patch_face_cells: Sequence = (),
                 patch_nbr_cells: Sequence = (), ordered: bool = False, tile_cell_start=None, adopt: bool = False):
        """patch_nbr_cells[p] (optional): local cells across patch p => cyclic (local) coupling; None => processor patch.
        ordered: keep the caller's numbering (mi_addr_create_ordered); tile_cell_start: the tiles as cell ranges, or None.
        adopt: renumber-at-bind (mi_addr_create_adopted) -- the addressing is that of the mesh RENUMBERED into the engine order;
        self.cell_map / face_map / face_flipped / lower_addr / upper_addr describe the renumbering (new -> old)"""
        self.ctx = ctx
        lo = np.ascontiguousarray(lower_addr, dtype=np.int32)
        up = np.ascontiguousarray(upper_addr, dtype=np.int32)
        self._patches = [np.ascontiguousarray(p, dtype=np.int32) for p in patch_face_cells]
        npatch = len(self._patches)
        sizes = (C.c_int32 * max(npatch, 1))(*[p.shape[0] for p in self._patches])
        ptrs = (C.POINTER(C.c_int32) * max(npatch, 1))(*[p.ctypes.data_as(C.POINTER(C.c_int32)) for p in self._patches])
        self.h = C.c_void_p()
        self._nbrs = [None if (k >= len(patch_nbr_cells) or patch_nbr_cells[k] is None)
                      else np.ascontiguousarray(patch_nbr_cells[k], dtype=np.int32) for k in range(npatch)]
        nptrs = (C.POINTER(C.c_int32) * max(npatch, 1))(*[q.ctypes.data_as(C.POINTER(C.c_int32)) if q is not None
                                                           else C.POINTER(C.c_int32)() for q in self._nbrs])
        if adopt:
            I32, U8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
            self.cell_map = np.empty(n_cells, dtype=np.int32); self.face_map = np.empty(lo.shape[0], dtype=np.int32)
            self.face_flipped = np.empty(lo.shape[0], dtype=np.uint8)
            self.lower_addr = np.empty(lo.shape[0], dtype=np.int32); self.upper_addr = np.empty(lo.shape[0], dtype=np.int32)
            _chk(lib().mi_addr_create_adopted(ctx.h, C.c_int32(n_cells), C.c_int32(lo.shape[0]), lo.ctypes.data_as(I32), up.ctypes.data_as(I32),
                                              C.c_int32(npatch), sizes, ptrs, nptrs, self.cell_map.ctypes.data_as(I32), self.face_map.ctypes.data_as(I32),
                                              self.face_flipped.ctypes.data_as(U8), self.lower_addr.ctypes.data_as(I32), self.upper_addr.ctypes.data_as(I32),
                                              C.byref(self.h)))
        elif ordered:
            ts = None if tile_cell_start is None else np.ascontiguousarray(tile_cell_start, dtype=np.int32)
            _chk(lib().mi_addr_create_ordered(ctx.h, C.c_int32(n_cells), C.c_int32(lo.shape[0]),
                                              lo.ctypes.data_as(C.POINTER(C.c_int32)), up.ctypes.data_as(C.POINTER(C.c_int32)),
                                              C.c_int32(npatch), sizes, ptrs, nptrs, C.c_int32(0 if ts is None else ts.shape[0] - 1),
                                              ts.ctypes.data_as(C.POINTER(C.c_int32)) if ts is not None else C.POINTER(C.c_int32)(), C.byref(self.h)))
        else:
            _chk(lib().mi_addr_create_coupled(ctx.h, C.c_int32(n_cells), C.c_int32(lo.shape[0]),
                                              lo.ctypes.data_as(C.POINTER(C.c_int32)), up.ctypes.data_as(C.POINTER(C.c_int32)),
                                              C.c_int32(npatch), sizes, ptrs, nptrs, C.byref(self.h)))
        self.n_cells = n_cells
        self.n_faces = int(lo.shape[0])
        self.n_ext = int(lib().mi_addr_n_ext(self.h))
        self.n_tiles = int(lib().mi_addr_n_tiles(self.h))

    def cell_perm(self) -> np.ndarray:
        out = np.empty(self.n_cells, dtype=np.int32)
        _chk(lib().mi_addr_cell_perm(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def tile_starts(self) -> np.ndarray:
        out = np.empty(self.n_tiles + 1, dtype=np.int32)
        _chk(lib().mi_addr_tile_starts(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    @property
    def is_ordered(self) -> bool:
        return bool(lib().mi_addr_is_ordered(self.h))

    def set_ami_patch(self, patch: int, nbr_patch: int, start=None, address=None, weights=None, low_weight=None):
        """patch becomes a cyclicAMI patch coupled to nbr_patch (mi_addr_set_ami_patch); start None: one face to one face with
        unit weights (a cyclic patch that carries a transformation factor)"""
        ip = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.int32)
        st, ad = ip(start), ip(address)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        lw = None if low_weight is None else np.ascontiguousarray(low_weight, dtype=np.uint8)
        cp = lambda a, t: a.ctypes.data_as(C.POINTER(t)) if a is not None else None
        _chk(lib().mi_addr_set_ami_patch(self.h, C.c_int32(patch), C.c_int32(nbr_patch), cp(st, C.c_int32), cp(ad, C.c_int32),
                                         cp(w, C.c_double), cp(lw, C.c_uint8)))

    def set_ami_patch_remote(self, patch: int, transport_patch: int, n_partner_faces: int, start, address, weights, low_weight=None):
        """cyclicAMI patch whose partner lives on another rank: its neighbour values are interpolated from what the processor
        patch `transport_patch` receives (mi_addr_set_ami_patch_remote)"""
        st = np.ascontiguousarray(start, dtype=np.int32); ad = np.ascontiguousarray(address, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        lw = None if low_weight is None else np.ascontiguousarray(low_weight, dtype=np.uint8)
        _chk(lib().mi_addr_set_ami_patch_remote(self.h, C.c_int32(patch), C.c_int32(transport_patch), C.c_int32(n_partner_faces),
                                                st.ctypes.data_as(C.POINTER(C.c_int32)), ad.ctypes.data_as(C.POINTER(C.c_int32)),
                                                w.ctypes.data_as(C.POINTER(C.c_double)),
                                                lw.ctypes.data_as(C.POINTER(C.c_uint8)) if lw is not None else C.POINTER(C.c_uint8)()))

    def set_ami_patch_remote_multi(self, patch: int, transport_patches, n_partner_faces, start, address, weights, low_weight=None):
        """cyclicAMI patch whose partner SIDE is split over several ranks: one transport patch per partner piece, addresses numbering the
        pieces' faces concatenated (mi_addr_set_ami_patch_remote_multi)"""
        tp = np.ascontiguousarray(transport_patches, dtype=np.int32); nf = np.ascontiguousarray(n_partner_faces, dtype=np.int32)
        st = np.ascontiguousarray(start, dtype=np.int32); ad = np.ascontiguousarray(address, dtype=np.int32)
        w = np.ascontiguousarray(weights, dtype=np.float64)
        lw = None if low_weight is None else np.ascontiguousarray(low_weight, dtype=np.uint8)
        I32 = C.POINTER(C.c_int32)
        _chk(lib().mi_addr_set_ami_patch_remote_multi(self.h, C.c_int32(patch), C.c_int32(tp.shape[0]), tp.ctypes.data_as(I32), nf.ctypes.data_as(I32),
                                                      st.ctypes.data_as(I32), ad.ctypes.data_as(I32), w.ctypes.data_as(C.POINTER(C.c_double)),
                                                      lw.ctypes.data_as(C.POINTER(C.c_uint8)) if lw is not None else C.POINTER(C.c_uint8)()))

    def set_ami_face_areas(self, patch: int, mag_sf):
        """face areas of a cyclicAMI patch (srcMagSf / tgtMagSf): the GAMG hierarchy agglomerates the AMI with them"""
        a = np.ascontiguousarray(mag_sf, dtype=np.float64)
        _chk(lib().mi_addr_set_ami_face_areas(self.h, C.c_int32(patch), a.ctypes.data_as(C.POINTER(C.c_double))))

    def patch_offsets(self) -> np.ndarray:
        out = np.empty(len(self._patches) + 1, dtype=np.int32)
        _chk(lib().mi_addr_patch_offsets(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def stats(self) -> dict:
        st = (C.c_int64 * 8)()
        _chk(lib().mi_addr_stats(self.h, st))
        keys = ["tiles", "slots", "entries", "halo", "max_cells", "max_slots", "max_halo", "lds_bytes_sym"]
        return dict(zip(keys, [int(v) for v in st]))

    def to_engine(self, x, out):
        _chk(lib().mi_vec_to_engine(self.h, _ptr(x), _ptr(out)))

    def from_engine(self, xe, out):
        _chk(lib().mi_vec_from_engine(self.h, _ptr(xe), _ptr(out)))

    def halo_pack(self, xe, send):
        _chk(lib().mi_halo_pack_engine(self.h, _ptr(xe), _ptr(send)))

    def close(self):
        if self.h:
            lib().mi_addr_destroy(self.h)
            self.h = C.c_void_p()


class Comm:
    """RCCL communicator of one rank (mi_comm_*).  ``unique_id()`` on rank 0, ship the 128 bytes to every rank."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * 128)()
        _chk(lib().mi_comm_unique_id(buf, C.c_int32(128)))
        return bytes(buf)

    def __init__(self, ctx: "Context", n_ranks: int, rank: int, uid: bytes):
        self.ctx = ctx
        self.h = C.c_void_p()
        self.n_ranks, self.rank = n_ranks, rank
        _chk(lib().mi_comm_create(ctx.h, C.c_int32(n_ranks), C.c_int32(rank), C.c_char_p(uid), C.byref(self.h)))

    def allreduce_sum(self, t):
        _chk(lib().mi_comm_allreduce_sum(self.h, _ptr(t), C.c_int64(t.numel())))

    # one-shot peer all-reduce (opt-in): window() on every rank, ship the 64 bytes to all ranks, peer_connect(all handles)
    def peer_window(self) -> bytes:
        buf = (C.c_char * 64)()
        _chk(lib().mi_comm_peer_window(self.h, buf, C.c_int32(64)))
        return bytes(buf)

    def peer_connect(self, handles):
        blob = b"".join(handles)
        assert len(blob) == 64 * self.n_ranks
        _chk(lib().mi_comm_peer_connect(self.h, C.c_char_p(blob), C.c_int32(self.n_ranks)))

    def peer_status(self):
        st, fg = C.c_int32(0), C.c_int32(0)
        _chk(lib().mi_comm_peer_status(self.h, C.byref(st), C.byref(fg)))
        return int(st.value), bool(fg.value)

    def peer_auto(self) -> bool:
        """collective: windows + handle exchange over the communicator's own transport + self-test + agreement (mi_comm_peer_auto);
        True on every rank when the scalars (and the halo of matrices attached afterwards) travel through peer windows"""
        on = C.c_int32(0)
        _chk(lib().mi_comm_peer_auto(self.h, C.byref(on)))
        return bool(on.value)

    def peer_selftest(self, rounds: int = 16) -> bool:
        ok = C.c_int32(0)
        _chk(lib().mi_comm_peer_selftest(self.h, C.c_int32(rounds), C.byref(ok)))
        return bool(ok.value)

    def peer_enable(self, on: bool):
        _chk(lib().mi_comm_peer_enable(self.h, C.c_int32(1 if on else 0)))

    def close(self):
        if self.h:
            lib().mi_comm_destroy(self.h)
            self.h = C.c_void_p()


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                          C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


class ExternalComm(Comm):
    """A communicator over the CALLER's transport (mi_comm_create_external) -- MPI through Pstream in an OpenFOAM shim, gloo
    in the tests.  ``allreduce(ptr, n)`` sums n doubles at device pointer ptr over the ranks in place; ``exchange(sends, recvs)``
    gets lists of (peer, tag, device pointer, count) and returns when the received data is in device memory."""

    def __init__(self, ctx: "Context", n_ranks: int, rank: int, allreduce, exchange):
        self.ctx = ctx
        self.h = C.c_void_p()
        self.n_ranks, self.rank = n_ranks, rank
        self.errors = []

        def _ar(_user, buf, n):
            try:
                allreduce(int(buf), int(n)); return 0
            except Exception as e:  # an exception must not unwind through the C frames
                self.errors.append(e); return 1

        def _ex(_user, ns, sp, st, sb, sc, nr, rp, rt, rb, rc):
            try:
                exchange([(sp[i], st[i], int(sb[i]), int(sc[i])) for i in range(ns)], [(rp[i], rt[i], int(rb[i]), int(rc[i])) for i in range(nr)]); return 0
            except Exception as e:
                self.errors.append(e); return 1

        self._cb = (ALLREDUCE_FN(_ar), EXCHANGE_FN(_ex))      # keep the trampolines alive as long as the communicator
        _chk(lib().mi_comm_create_external(ctx.h, C.c_int32(n_ranks), C.c_int32(rank), self._cb[0], self._cb[1], None, C.byref(self.h)))


class Matrix:
    """lduMatrix coefficients bound to an :class:`Addressing`."""

    def __init__(self, addr: Addressing):
        self.addr = addr
        self.h = C.c_void_p()
        _chk(lib().mi_matrix_create(addr.h, C.byref(self.h)))

    def set_coeffs(self, diag, upper, lower=None):
        _chk(lib().mi_matrix_set_coeffs(self.h, _ptr(diag), _ptr(upper), _ptr(lower)))

    def set_interface_coeffs(self, patch: int, bou, inte=None):
        _chk(lib().mi_matrix_set_interface_coeffs(self.h, C.c_int32(patch), _ptr(bou), _ptr(inte)))

    def set_patch_transform(self, patch: int, factor: float):
        """transformCoupleField factor of a coupled patch (mi_matrix_set_patch_transform)"""
        _chk(lib().mi_matrix_set_patch_transform(self.h, C.c_int32(patch), C.c_double(factor)))

    def set_ext(self, ext):
        _chk(lib().mi_matrix_set_ext(self.h, _ptr(ext)))

    # SpMV family (caller order) ---------------------------------------------
    def amul(self, psi, out):
        _chk(lib().mi_amul(self.h, _ptr(psi), _ptr(out)))

    def tmul(self, psi, out):
        _chk(lib().mi_tmul(self.h, _ptr(psi), _ptr(out)))

    def sumA(self, out):
        _chk(lib().mi_sumA(self.h, _ptr(out)))

    def residual(self, psi, source, out):
        _chk(lib().mi_residual(self.h, _ptr(psi), _ptr(source), _ptr(out)))

    def H(self, psi, out):
        _chk(lib().mi_H(self.h, _ptr(psi), _ptr(out)))

    def H1(self, out):
        _chk(lib().mi_H1(self.h, _ptr(out)))

    def patch_neighbour_field(self, psi, out):
        """psi across every interface face (n_ext values, caller patch order); exchanges when a communicator is attached"""
        _chk(lib().mi_matrix_patch_neighbour_field(self.h, _ptr(psi), _ptr(out)))

    def norm_factor(self, psi, source, Apsi):
        out = C.c_double(0.0)
        _chk(lib().mi_norm_factor(self.h, _ptr(psi), _ptr(source), _ptr(Apsi), C.byref(out)))
        return out.value

    def faceH(self, psi, out):
        _chk(lib().mi_faceH(self.h, _ptr(psi), _ptr(out)))

    # engine order -------------------------------------------------------------
    def amul_engine(self, psi_e, out_e, which: int = 0):
        _chk(lib().mi_amul_engine(self.h, _ptr(psi_e), _ptr(out_e), int(which)))

    def tmul_engine(self, psi_e, out_e, which: int = 0):
        _chk(lib().mi_tmul_engine(self.h, _ptr(psi_e), _ptr(out_e), int(which)))

    def precondition(self, kind: str, rA, wA, transpose: bool = False):
        _chk(lib().mi_precondition(self.h, PRECOND[kind], int(transpose), _ptr(rA), _ptr(wA)))

    def jacobi_smooth(self, psi, source, n_sweeps: int, omega: float = 0.9):
        _chk(lib().mi_jacobi_smooth(self.h, C.c_double(omega), _ptr(psi), _ptr(source), C.c_int32(n_sweeps)))

    # solvers -------------------------------------------------------------------
    def _solve(self, fn, psi, source, extra, tolerance, relTol, maxIter, minIter):
        ctl = SolverControls(tolerance, relTol, maxIter, minIter)
        perf = SolverPerf()
        hist_len = maxIter + 2
        hist = np.full(hist_len, np.nan)
        _chk(getattr(lib(), fn)(self.h, _ptr(psi), _ptr(source), C.byref(ctl), *extra, C.byref(perf),
                                hist.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(hist_len)))
        out = {k: getattr(perf, k) for k, _ in SolverPerf._fields_ if k != "reserved"}
        out["history"] = hist[~np.isnan(hist)].copy()
        return out

    def pcg(self, psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0):
        return self._solve("mi_pcg_solve", psi, source, (C.c_int(PRECOND[precond]),), tolerance, relTol, maxIter, minIter)

    def pbicg(self, psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0):
        return self._solve("mi_pbicg_solve", psi, source, (C.c_int(PRECOND[precond]),), tolerance, relTol, maxIter, minIter)

    def pbicg_multi(self, psis, sources, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0, diags=None):
        """the components of a vector equation in one solve (mi_pbicg_solve_multi): list of per-component results;
        diags: per-component diagonals (addBoundaryDiag(diag, cmpt)) or None = the bound diagonal for all"""
        n = len(psis)
        assert 1 <= n <= 3 and len(sources) == n
        ctl = SolverControls(tolerance, relTol, maxIter, minIter)
        perf = (SolverPerf * n)()
        hist_len = maxIter + 2
        hist = np.full((n, hist_len), np.nan)
        P = (C.c_void_p * n)(*[_ptr(t) for t in psis])
        S = (C.c_void_p * n)(*[_ptr(t) for t in sources])
        D = None if diags is None else (C.c_void_p * n)(*[_ptr(t) for t in diags])
        _chk(lib().mi_pbicg_solve_multi(self.h, C.c_int32(n), D, P, S, C.byref(ctl), C.c_int(PRECOND[precond]), perf,
                                        hist.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(hist_len)))
        out = []
        for k in range(n):
            d = {f: getattr(perf[k], f) for f, _ in SolverPerf._fields_ if f != "reserved"}
            d["history"] = hist[k][~np.isnan(hist[k])].copy()
            out.append(d)
        return out

    def pbicgstab(self, psi, source, precond="diagonal", tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0,
                  replicate_quirk=True):
        return self._solve("mi_pbicgstab_solve", psi, source, (C.c_int(PRECOND[precond]), C.c_int(int(replicate_quirk))),
                           tolerance, relTol, maxIter, minIter)

    def smooth_solve(self, psi, source, n_sweeps=1, omega=0.9, tolerance=1e-6, relTol=0.0, maxIter=1000, minIter=0):
        return self._solve("mi_smooth_solve", psi, source, (C.c_double(omega), C.c_int32(n_sweeps)),
                           tolerance, relTol, maxIter, minIter)

    # PCG session (bench) -------------------------------------------------------
    def pcg_begin(self, psi0, source, precond="diagonal", tolerance=0.0, relTol=0.0, maxIter=1000, minIter=0,
                  history_len=0):
        ctl = SolverControls(tolerance, relTol, maxIter, minIter)
        _chk(lib().mi_pcg_begin(self.h, _ptr(psi0), _ptr(source), C.byref(ctl), C.c_int(PRECOND[precond]),
                                C.c_int32(history_len)))

    def pcg_iterate(self, n_iters: int, time_amul: bool = False, event_stride: int = 1) -> Optional[float]:
        if time_amul:
            ms = C.c_float()
            _chk(lib().mi_pcg_iterate_sampled(self.h, C.c_int32(n_iters), C.c_int32(event_stride), C.byref(ms)))
            return ms.value
        _chk(lib().mi_pcg_iterate(self.h, C.c_int32(n_iters), None))
        return None

    def pcg_end(self, psi_out=None, history_len=0):
        perf = SolverPerf()
        hist = np.full(max(history_len, 1), np.nan)
        _chk(lib().mi_pcg_end(self.h, _ptr(psi_out), C.byref(perf), hist.ctypes.data_as(C.POINTER(C.c_double)),
                              C.c_int32(history_len)))
        out = {k: getattr(perf, k) for k, _ in SolverPerf._fields_ if k != "reserved"}
        out["history"] = hist[: min(history_len, perf.nIterations + 1)].copy()
        return out

    # distributed PCG phases (parallel.py) -----------------------------------------
    def dpcg_set_buffers(self, psi_e, src_e, pA_e, wA_e, rA_e, scal8, send_buf, precond="diagonal",
                         tolerance=0.0, relTol=0.0, maxIter=1000, minIter=0, history_len=0):
        ctl = SolverControls(tolerance, relTol, maxIter, minIter)
        _chk(lib().mi_dpcg_set_buffers(self.h, _ptr(psi_e), _ptr(src_e), _ptr(pA_e), _ptr(wA_e), _ptr(rA_e),
                                       _ptr(scal8), _ptr(send_buf), C.byref(ctl), C.c_int(PRECOND[precond]),
                                       C.c_int32(history_len)))

    def dpcg_phase(self, phase: int, it: int = 0, arg: float = 0.0):
        _chk(lib().mi_dpcg_phase(self.h, int(phase), C.c_int32(it), C.c_double(arg)))

    def dpcg_status(self, history_len=0):
        perf = SolverPerf()
        done = C.c_int32()
        hist = np.full(max(history_len, 1), np.nan)
        _chk(lib().mi_dpcg_status(self.h, C.byref(perf), C.byref(done), hist.ctypes.data_as(C.POINTER(C.c_double)),
                                  C.c_int32(history_len)))
        out = {k: getattr(perf, k) for k, _ in SolverPerf._fields_ if k != "reserved"}
        out["done"] = int(done.value)
        out["history"] = hist[~np.isnan(hist)].copy()
        return out

    def attach_comm(self, reduce: "Comm", halo: "Comm", patch_rank, patch_nbr_patch=None, n_global=0):
        """decomposed-case behaviour for every operator and solver of this matrix (mi_matrix_attach_comm)"""
        pr = np.ascontiguousarray(patch_rank, dtype=np.int32)
        pn = None if patch_nbr_patch is None else np.ascontiguousarray(patch_nbr_patch, dtype=np.int32)
        I32 = C.POINTER(C.c_int32)
        _chk(lib().mi_matrix_attach_comm(self.h, reduce.h, halo.h, pr.ctypes.data_as(I32) if pr.size else I32(),
                                         pn.ctypes.data_as(I32) if pn is not None and pn.size else I32(), C.c_int64(n_global)))
        self._comms = (reduce, halo)

    def detach_comm(self):
        _chk(lib().mi_matrix_detach_comm(self.h))
        self._comms = None

    def peer_halo_auto(self) -> bool:
        """collective: halo windows of this attached matrix (mi_matrix_peer_halo_auto); True when every rank has them"""
        on = C.c_int32(0)
        _chk(lib().mi_matrix_peer_halo_auto(self.h, C.byref(on)))
        return bool(on.value)

    def peer_halo_status(self):
        """(windows in use, a wait ran out of polls)"""
        on, st = C.c_int32(0), C.c_int32(0)
        _chk(lib().mi_matrix_peer_halo_status(self.h, C.byref(on), C.byref(st)))
        return bool(on.value), int(st.value)

    def dpcg_comm_begin(self, reduce: "Comm", halo: "Comm", patch_rank, patch_nbr_patch=None, n_global=0):
        pr = np.ascontiguousarray(patch_rank, dtype=np.int32)
        pn = None if patch_nbr_patch is None else np.ascontiguousarray(patch_nbr_patch, dtype=np.int32)
        I32 = C.POINTER(C.c_int32)
        _chk(lib().mi_dpcg_comm_begin(self.h, reduce.h, halo.h, pr.ctypes.data_as(I32) if pr.size else I32(),
                                      pn.ctypes.data_as(I32) if pn is not None and pn.size else I32(), C.c_int64(n_global)))

    def dpcg_comm_iterate(self, n_iters: int, event_stride: int = 0):
        _chk(lib().mi_dpcg_comm_iterate(self.h, C.c_int32(n_iters), C.c_int32(event_stride)))

    def event_record(self, idx: int):
        _chk(lib().mi_event_record(self.h, C.c_int32(idx)))

    def event_elapsed_ms(self, i0: int, i1: int) -> float:
        ms = C.c_float()
        _chk(lib().mi_event_elapsed_ms(self.h, C.c_int32(i0), C.c_int32(i1), C.byref(ms)))
        return ms.value

    def occupancy(self):
        b, l, s = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(lib().mi_debug_occupancy(self.h, C.byref(b), C.byref(l), C.byref(s)))
        return dict(blocks_per_cu=b.value, lds_bytes=l.value, block_size=s.value)

    def bench_amul(self, reps: int) -> float:
        ms = C.c_float()
        _chk(lib().mi_bench_amul(self.h, C.c_int32(reps), C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            lib().mi_matrix_destroy(self.h)
            self.h = C.c_void_p()


class Gamg:
    """GAMG hierarchy + solver (lduMatrix::solver 'GAMG', agglomerator faceAreaPair/algebraicPair)."""

    def __init__(self, addr: Addressing, face_weights, n_cells_in_coarsest_level: int = 10, forward: bool = True,
                 comms=None, patch_rank=None, patch_nbr_patch=None, merge_levels: int = 1, dummy_levels: int = 0):
        """comms = (reduce, halo) Comm pair of a decomposed case (the matrix must be attached to the same pair);
        dummy_levels n > 0: the reference's dummyAgglomeration (n identity levels), face_weights unused"""
        self.addr = addr
        self.h = C.c_void_p()
        if dummy_levels > 0:
            _chk(lib().mi_gamg_create_dummy(addr.h, C.c_int32(dummy_levels), C.byref(self.h)))
            self._keep = None
            self.n_levels = int(lib().mi_gamg_n_levels(self.h))
            self.forward_out = bool(lib().mi_gamg_forward_out(self.h))
            return
        w = np.ascontiguousarray(face_weights, dtype=np.float64)
        I32 = C.POINTER(C.c_int32)
        pr = None if patch_rank is None else np.ascontiguousarray(patch_rank, dtype=np.int32)
        pn = None if patch_nbr_patch is None else np.ascontiguousarray(patch_nbr_patch, dtype=np.int32)
        _chk(lib().mi_gamg_create_coupled(addr.h, w.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(n_cells_in_coarsest_level),
                                          C.c_int32(merge_levels), int(forward), comms[0].h if comms else C.c_void_p(), comms[1].h if comms else C.c_void_p(),
                                          pr.ctypes.data_as(I32) if pr is not None and pr.size else I32(),
                                          pn.ctypes.data_as(I32) if pn is not None and pn.size else I32(), C.byref(self.h)))
        self._keep = (comms, pr, pn)
        self.n_levels = int(lib().mi_gamg_n_levels(self.h))
        self.forward_out = bool(lib().mi_gamg_forward_out(self.h))

    def level_sizes(self, level: int):
        out = (C.c_int32 * 4)()
        _chk(lib().mi_gamg_level_sizes(self.h, C.c_int32(level), out))
        return dict(zip(("n_fine", "n_fine_faces", "n_coarse", "n_coarse_faces"), [int(v) for v in out]))

    def solve(self, mat: Matrix, psi, source, **kw):
        ctl = gamg_controls(**kw)
        perf = SolverPerf()
        hist_len = ctl.maxIter + 2
        hist = np.full(hist_len, np.nan)
        _chk(lib().mi_gamg_solve(self.h, mat.h, _ptr(psi), _ptr(source), C.byref(ctl), C.byref(perf),
                                 hist.ctypes.data_as(C.POINTER(C.c_double)), C.c_int32(hist_len)))
        out = {k: getattr(perf, k) for k, _ in SolverPerf._fields_ if k != "reserved"}
        out["history"] = hist[~np.isnan(hist)].copy()
        return out

    def restrict(self, level, fine, coarse):
        _chk(lib().mi_gamg_restrict(self.h, C.c_int32(level), _ptr(fine), _ptr(coarse)))

    def prolong(self, level, coarse, fine):
        _chk(lib().mi_gamg_prolong(self.h, C.c_int32(level), _ptr(coarse), _ptr(fine)))

    def level_coeffs(self, mat: Matrix, level, diag, upper, lower=None):
        _chk(lib().mi_gamg_level_coeffs(self.h, mat.h, C.c_int32(level), _ptr(diag), _ptr(upper), _ptr(lower)))

    def close(self):
        if self.h:
            lib().mi_gamg_destroy(self.h)
            self.h = C.c_void_p()


def gamg_host_hierarchy_domains(domains, face_weights_per_domain, n_cells_in_coarsest_level=10, forward=True, merge_levels=1):
    """Host-only build of the per-rank GAMG hierarchies of a decomposed case (list of LduCase sub-domains with interfaces), all
    domains in this process; returns [domain][level] dicts incl. "patches": [{faceRestrict, faceCells, nbrCells}] (tests)."""
    I32P, F64P = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    D = len(domains)
    keep = []
    def i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return a.ctypes.data_as(I32P)
    n_cells = (C.c_int32 * D)(*[d.n_cells for d in domains]); n_faces = (C.c_int32 * D)(*[d.n_faces for d in domains])
    lower = (I32P * D)(*[i32(d.lower_addr) for d in domains]); upper = (I32P * D)(*[i32(d.upper_addr) for d in domains])
    ws = [np.ascontiguousarray(w, dtype=np.float64) for w in face_weights_per_domain]; keep.append(ws)
    weights = (F64P * D)(*[w.ctypes.data_as(F64P) for w in ws])
    n_patches = (C.c_int32 * D)(*[len(d.interfaces) for d in domains])
    sizes = (I32P * D)(*[i32([len(i.face_cells) for i in d.interfaces] or [0]) for d in domains])
    nbr_d = (I32P * D)(*[i32([i.nbr_domain for i in d.interfaces] or [0]) for d in domains])
    nbr_p = (I32P * D)(*[i32([i.nbr_patch for i in d.interfaces] or [0]) for d in domains])
    fcs = []
    for d in domains:
        arr = (I32P * max(len(d.interfaces), 1))(*[i32(i.face_cells) for i in d.interfaces]); keep.append(arr); fcs.append(arr)
    PP = C.POINTER(I32P)
    fc = (PP * D)(*[C.cast(a, PP) for a in fcs])
    out = (C.c_void_p * D)()
    _chk(lib().mi_gamg_host_build_domains(C.c_int32(D), n_cells, n_faces, lower, upper, weights, n_patches, sizes, fc, nbr_d, nbr_p,
                                          C.c_int32(n_cells_in_coarsest_level), C.c_int32(merge_levels), int(forward), out))
    res = []
    try:
        for d in range(D):
            h = C.c_void_p(out[d]); levels = []
            for lvl in range(int(lib().mi_gamg_host_n_levels(h))):
                lv = {}
                for name in ("restrictMap", "faceRestrict", "cLower", "cUpper"):
                    data, ln, es = C.c_void_p(), C.c_int64(), C.c_int32()
                    _chk(lib().mi_gamg_host_array(h, C.c_int32(lvl), name.encode(), C.byref(data), C.byref(ln), C.byref(es)))
                    lv[name] = np.frombuffer((C.c_char * (ln.value * 4)).from_address(data.value), dtype=np.int32).copy() if ln.value else np.zeros(0, np.int32)
                lv["patches"] = []
                for p in range(len(domains[d].interfaces)):
                    pd = {}
                    for name in ("faceRestrict", "faceCells", "nbrCells"):
                        data, ln = C.c_void_p(), C.c_int64()
                        _chk(lib().mi_gamg_host_patch_array(h, C.c_int32(lvl), C.c_int32(p), name.encode(), C.byref(data), C.byref(ln)))
                        pd[name] = np.frombuffer((C.c_char * (ln.value * 4)).from_address(data.value), dtype=np.int32).copy() if ln.value else np.zeros(0, np.int32)
                    lv["patches"].append(pd)
                levels.append(lv)
            res.append(levels)
    finally:
        for d in range(D):
            lib().mi_gamg_host_free(C.c_void_p(out[d]))
    return res


def gamg_host_hierarchy_ami(case, face_weights, n_cells_in_coarsest_level=10):
    """Host-only build of the GAMG hierarchy of ONE domain whose interfaces are all cyclicAMI (LduCase with ami_* fields);
    returns [level] dicts incl. "patches": [{faceRestrict, faceCells, amiStart, amiAddr, amiW, amiMagSf}] (CPU tests)."""
    I32P, F64P = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    keep = []
    def i32(a):
        a = np.ascontiguousarray(a, dtype=np.int32); keep.append(a); return a.ctypes.data_as(I32P)
    def f64(a):
        a = np.ascontiguousarray(a, dtype=np.float64); keep.append(a); return a.ctypes.data_as(F64P)
    P = len(case.interfaces)
    sizes = i32([len(i.face_cells) for i in case.interfaces]); nbr = i32([i.nbr_patch for i in case.interfaces])
    fc = (I32P * P)(*[i32(i.face_cells) for i in case.interfaces])
    st = (I32P * P)(*[i32(i.ami_start) for i in case.interfaces]); ad = (I32P * P)(*[i32(i.ami_addr) for i in case.interfaces])
    ww = (F64P * P)(*[f64(i.ami_w) for i in case.interfaces]); ms = (F64P * P)(*[f64(i.ami_magsf) for i in case.interfaces])
    h = C.c_void_p()
    _chk(lib().mi_gamg_host_build_ami(C.c_int32(case.n_cells), C.c_int32(case.n_faces), i32(case.lower_addr), i32(case.upper_addr), f64(face_weights),
                                      C.c_int32(n_cells_in_coarsest_level), C.c_int32(P), sizes, fc, nbr, st, ad, ww, ms, C.byref(h)))
    levels = []
    try:
        for lvl in range(int(lib().mi_gamg_host_n_levels(h))):
            lv = {}
            for name in ("restrictMap",):
                data, ln, es = C.c_void_p(), C.c_int64(), C.c_int32()
                _chk(lib().mi_gamg_host_array(h, C.c_int32(lvl), name.encode(), C.byref(data), C.byref(ln), C.byref(es)))
                lv[name] = np.frombuffer((C.c_char * (ln.value * 4)).from_address(data.value), dtype=np.int32).copy()
            lv["patches"] = []
            for p in range(P):
                pd = {}
                for name, dt in (("faceRestrict", np.int32), ("faceCells", np.int32), ("amiStart", np.int32), ("amiAddr", np.int32), ("amiW", np.float64), ("amiMagSf", np.float64)):
                    data, ln = C.c_void_p(), C.c_int64()
                    _chk(lib().mi_gamg_host_patch_array(h, C.c_int32(lvl), C.c_int32(p), name.encode(), C.byref(data), C.byref(ln)))
                    nb = ln.value * np.dtype(dt).itemsize
                    pd[name] = np.frombuffer((C.c_char * nb).from_address(data.value), dtype=dt).copy() if ln.value else np.zeros(0, dt)
                lv["patches"].append(pd)
            levels.append(lv)
    finally:
        lib().mi_gamg_host_free(h)
    return levels


def gamg_host_hierarchy(n_cells, lower_addr, upper_addr, face_weights, n_cells_in_coarsest_level=10, forward=True, merge_levels=1):
    """Host-only build of the GAMG hierarchy; returns a list of per-level dicts of numpy arrays (tests)."""
    lo = np.ascontiguousarray(lower_addr, dtype=np.int32)
    up = np.ascontiguousarray(upper_addr, dtype=np.int32)
    w = np.ascontiguousarray(face_weights, dtype=np.float64)
    h = C.c_void_p()
    _chk(lib().mi_gamg_host_build(C.c_int32(n_cells), C.c_int32(lo.shape[0]), lo.ctypes.data_as(C.POINTER(C.c_int32)),
                                  up.ctypes.data_as(C.POINTER(C.c_int32)), w.ctypes.data_as(C.POINTER(C.c_double)),
                                  C.c_int32(n_cells_in_coarsest_level), C.c_int32(merge_levels), int(forward), C.byref(h)))
    out = []
    try:
        for lvl in range(int(lib().mi_gamg_host_n_levels(h))):
            d = {}
            for name in ("restrictMap", "faceRestrict", "faceFlip", "cLower", "cUpper", "cellChildStart", "cellChild",
                         "faceChildStart", "faceChild", "diagChildStart", "diagChild"):
                data, ln, es = C.c_void_p(), C.c_int64(), C.c_int32()
                _chk(lib().mi_gamg_host_array(h, C.c_int32(lvl), name.encode(), C.byref(data), C.byref(ln), C.byref(es)))
                dt = np.uint8 if es.value == 1 else np.int32
                if ln.value:
                    buf = (C.c_char * (ln.value * es.value)).from_address(data.value)
                    d[name] = np.frombuffer(buf, dtype=dt).copy()
                else:
                    d[name] = np.zeros(0, dtype=dt)
            out.append(d)
    finally:
        lib().mi_gamg_host_free(h)
    return out


class Patch:
    """A boundary patch (its faceCells) for the assembly sweeps."""

    def __init__(self, ctx: Context, n_cells: int, face_cells):
        fc = np.ascontiguousarray(face_cells, dtype=np.int32)
        self.h = C.c_void_p()
        _chk(lib().mi_patch_create(ctx.h, C.c_int32(n_cells), C.c_int32(fc.shape[0]),
                                   fc.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(self.h)))

    def add(self, pf, intf, fn: int = 0):
        _chk(lib().mi_patch_add(self.h, _ptr(pf), _ptr(intf), int(fn)))

    def add_product(self, pf, q, intf, fn: int = 0):
        """intf[faceCells] += pf*q (coupled part of addBoundarySource)"""
        _chk(lib().mi_patch_add_product(self.h, _ptr(pf), _ptr(q), _ptr(intf), int(fn)))

    def flux(self, internal_coeffs, boundary_coeffs, psi, out, patch_neighbour_field=None):
        """boundary part of fvMatrix::flux; patch_neighbour_field only for coupled patches"""
        _chk(lib().mi_patch_flux(self.h, _ptr(internal_coeffs), _ptr(boundary_coeffs), _ptr(psi), _ptr(patch_neighbour_field), _ptr(out)))

    def internal_field(self, psi, out):
        """fvPatchField::patchInternalField: out[i] = psi[faceCells[i]]"""
        _chk(lib().mi_patch_internal_field(self.h, _ptr(psi), _ptr(out)))

    def sngrad_correction_flux(self, corr_vecs, weights, grad, nbr_grad, gamma_magsf, out):
        """non-orthogonal correction flux on a COUPLED patch (gaussLaplacianSchemes.C:64-90 + surfaceInterpolationScheme.C:360-365)"""
        _chk(lib().mi_patch_sngrad_correction_flux(self.h, _ptr(corr_vecs[0]), _ptr(corr_vecs[1]), _ptr(corr_vecs[2]), _ptr(weights), _ptr(grad[0]),
                                                   _ptr(grad[1]), _ptr(grad[2]), _ptr(nbr_grad[0]), _ptr(nbr_grad[1]), _ptr(nbr_grad[2]),
                                                   _ptr(gamma_magsf), _ptr(out)))

    def close(self):
        if self.h:
            lib().mi_patch_destroy(self.h)
            self.h = C.c_void_p()


class FvmTerms(C.Structure):
    """mi_fvm_terms (include/mi_ldu.h)"""
    _fields_ = [("ddt", C.c_int32), ("r_delta_t", C.c_double), ("rho_value", C.c_double), ("rho_dev", C.c_void_p), ("rho_old_dev", C.c_void_p),
                ("vol_dev", C.c_void_p), ("div_flux_dev", C.c_void_p), ("div_weights_dev", C.c_void_p), ("lap_delta_coeffs_dev", C.c_void_p),
                ("lap_gamma_magsf_dev", C.c_void_p), ("sp_dev", C.c_void_p), ("sp_sign", C.c_double), ("n_rhs", C.c_int32),
                ("psi_old_dev", C.POINTER(C.c_void_p)), ("n_su", C.c_int32), ("su_dev", C.POINTER(C.c_void_p)), ("su_sign", C.POINTER(C.c_double))]


class Assembly:
    """fvm::div / fvm::laplacian / negSumDiag / relax ... on caller-order arrays of an Addressing."""

    def __init__(self, addr: Addressing):
        self.addr = addr

    def row_face_op(self, kind: int, lower, upper, inout):
        _chk(lib().mi_row_face_op(self.addr.h, int(kind), _ptr(lower), _ptr(upper), _ptr(inout)))

    def fvm_laplacian(self, delta_coeffs, gamma_magsf, upper_out, diag_out):
        _chk(lib().mi_fvm_laplacian(self.addr.h, _ptr(delta_coeffs), _ptr(gamma_magsf), _ptr(upper_out), _ptr(diag_out)))

    def fvm_div(self, weights, face_flux, lower_out, upper_out, diag_out):
        _chk(lib().mi_fvm_div(self.addr.h, _ptr(weights), _ptr(face_flux), _ptr(lower_out), _ptr(upper_out), _ptr(diag_out)))

    def surface_integrate(self, ssf, vol, ivf):
        _chk(lib().mi_surface_integrate(self.addr.h, _ptr(ssf), _ptr(vol), _ptr(ivf)))

    def face_interpolate(self, lam, phi, sf):
        _chk(lib().mi_face_interpolate(self.addr.h, _ptr(lam), _ptr(phi), _ptr(sf)))

    def fvm_ddt_euler(self, r_delta_t, rho, vol, psi_old, diag_out, source_out):
        _chk(lib().mi_fvm_ddt_euler(self.addr.ctx.h, C.c_int64(vol.numel()), C.c_double(r_delta_t), C.c_double(rho), _ptr(vol), _ptr(psi_old),
                                    _ptr(diag_out), _ptr(source_out)))

    def fvm_ddt_euler_rho(self, r_delta_t, rho, rho_old, vol, psi_old, diag_out, source_out):
        """fvm::ddt(rho, vf) with a density field (EulerDdtScheme.C:403-440)"""
        _chk(lib().mi_fvm_ddt_euler_rho(self.addr.ctx.h, C.c_int64(vol.numel()), C.c_double(r_delta_t), _ptr(rho), _ptr(rho_old), _ptr(vol),
                                        _ptr(psi_old), _ptr(diag_out), _ptr(source_out)))

    def fvm_su(self, vol, su, source_inout):
        _chk(lib().mi_fvm_su(self.addr.ctx.h, C.c_int64(vol.numel()), _ptr(vol), _ptr(su), _ptr(source_inout)))

    def fvm_sp(self, vol, sp, diag_inout):
        """sp: a cell field or a number (fvmSup.C:100-170)"""
        if isinstance(sp, (int, float)):
            _chk(lib().mi_fvm_sp(self.addr.ctx.h, C.c_int64(vol.numel()), _ptr(vol), None, C.c_double(sp), _ptr(diag_inout)))
        else:
            _chk(lib().mi_fvm_sp(self.addr.ctx.h, C.c_int64(vol.numel()), _ptr(vol), _ptr(sp), C.c_double(0.0), _ptr(diag_inout)))

    def fvm_susp(self, vol, susp, vf, diag_inout, source_inout):
        _chk(lib().mi_fvm_susp(self.addr.ctx.h, C.c_int64(vol.numel()), _ptr(vol), _ptr(susp), _ptr(vf), _ptr(diag_inout), _ptr(source_inout)))

    def flux_div(self, lam, sf, v, phi_out, div_out, cell_scale=None, add_a=None, add_b=None, vol=None):
        """phi = Sf & interpolate([cell_scale *] v) [+ add_a [* add_b]] and div = surfaceIntegrate(phi) [/ vol] in one row pass"""
        _chk(lib().mi_flux_div(self.addr.h, _ptr(lam), _ptr(sf[0]), _ptr(sf[1]), _ptr(sf[2]), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(cell_scale),
                               _ptr(add_a), _ptr(add_b), _ptr(phi_out), _ptr(vol), _ptr(div_out)))

    def ddt_phi_corr(self, r_delta_t, lam, sf, u_old, rho_old, phi_old, out):
        """fvc::ddtCorr(rho, U, phi) on the internal faces (EulerDdtScheme.C:663-720; rho_old None: :523-551)"""
        _chk(lib().mi_ddt_phi_corr(self.addr.h, C.c_double(r_delta_t), _ptr(lam), _ptr(sf[0]), _ptr(sf[1]), _ptr(sf[2]), _ptr(u_old[0]), _ptr(u_old[1]),
                                   _ptr(u_old[2]), _ptr(rho_old), _ptr(phi_old), _ptr(out)))

    def upwind_weights(self, face_flux, w_out):
        _chk(lib().mi_upwind_weights(self.addr.ctx.h, C.c_int64(face_flux.numel()), _ptr(face_flux), _ptr(w_out)))

    def limited_linear_weights(self, k, cd_weights, face_flux, phi, grad, centres, w_out, limiter_out=None):
        _chk(lib().mi_limited_linear_weights(self.addr.h, C.c_double(k), _ptr(cd_weights), _ptr(face_flux), _ptr(phi), _ptr(grad[0]), _ptr(grad[1]),
                                             _ptr(grad[2]), _ptr(centres[0]), _ptr(centres[1]), _ptr(centres[2]), _ptr(w_out), _ptr(limiter_out)))

    def gauss_grad(self, sf, ssf, vol, grad_out):
        _chk(lib().mi_gauss_grad(self.addr.h, _ptr(sf[0]), _ptr(sf[1]), _ptr(sf[2]), _ptr(ssf), _ptr(vol), _ptr(grad_out[0]), _ptr(grad_out[1]),
                                 _ptr(grad_out[2])))

    def sngrad_correction_flux(self, corr_vecs, weights, grad, gamma_magsf, out):
        """gammaMagSf * (nonOrthCorrectionVectors & interpolate(grad)) on the internal faces (gaussLaplacianSchemes.C:64-90)"""
        _chk(lib().mi_sngrad_correction_flux(self.addr.h, _ptr(corr_vecs[0]), _ptr(corr_vecs[1]), _ptr(corr_vecs[2]), _ptr(weights), _ptr(grad[0]),
                                             _ptr(grad[1]), _ptr(grad[2]), _ptr(gamma_magsf), _ptr(out)))

    def submul(self, x, y, inout):
        """inout -= x*y (source -= V*div(...))"""
        _chk(lib().mi_vec_submul(self.addr.ctx.h, C.c_int64(inout.numel()), _ptr(x), _ptr(y), _ptr(inout)))

    def axpby(self, a, x, b, y, out):
        _chk(lib().mi_vec_axpby(self.addr.ctx.h, C.c_int64(x.numel()), C.c_double(a), _ptr(x), C.c_double(b), _ptr(y), _ptr(out)))

    def assemble(self, upper_out, diag_out, lower_out=None, sources_out=(), ddt=None, div=None, laplacian=None, sp=None, su=(), sum_mag_out=None):
        """[fvm::ddt] + [fvm::div] - [fvm::laplacian] [+- fvm::Sp] [+- su] in one row pass (mi_fvm_assemble).
        ddt = dict(r_delta_t=, vol=, psi_old=[...], rho=None | tensor, rho_old=None | tensor, rho_value=1.0); div = dict(flux=, weights=None (upwind) | tensor);
        laplacian = dict(delta_coeffs=, gamma_magsf=); sp = (field, sign); su = [(sign, [field per rhs]), ...]; vol is taken from ddt or the `vol` key of sp / su
        through ddt["vol"] (pass ddt=dict(vol=...) with r_delta_t omitted for no time derivative)."""
        t = FvmTerms()
        n_rhs = len(sources_out)
        keep = []
        if ddt is not None and "r_delta_t" in ddt:
            t.ddt = 1; t.r_delta_t = float(ddt["r_delta_t"]); t.rho_value = float(ddt.get("rho_value", 1.0))
            t.rho_dev = _ptr(ddt.get("rho")).value; t.rho_old_dev = _ptr(ddt.get("rho_old")).value
            po = (C.c_void_p * max(n_rhs, 1))(*[_ptr(x) for x in ddt["psi_old"]]); keep.append(po)
            t.psi_old_dev = C.cast(po, C.POINTER(C.c_void_p))
        if ddt is not None:
            t.vol_dev = _ptr(ddt.get("vol")).value
        if div is not None:
            t.div_flux_dev = _ptr(div["flux"]).value; t.div_weights_dev = _ptr(div.get("weights")).value
        if laplacian is not None:
            t.lap_delta_coeffs_dev = _ptr(laplacian["delta_coeffs"]).value; t.lap_gamma_magsf_dev = _ptr(laplacian["gamma_magsf"]).value
        if sp is not None:
            t.sp_dev = _ptr(sp[0]).value; t.sp_sign = float(sp[1])
        t.n_rhs = n_rhs; t.n_su = len(su)
        if su:
            sd = (C.c_void_p * (len(su) * n_rhs))(*[_ptr(f) for _, fields in su for f in fields]); keep.append(sd)
            sg = (C.c_double * len(su))(*[float(sign) for sign, _ in su]); keep.append(sg)
            t.su_dev = C.cast(sd, C.POINTER(C.c_void_p)); t.su_sign = C.cast(sg, C.POINTER(C.c_double))
        so = (C.c_void_p * max(n_rhs, 1))(*[_ptr(x) for x in sources_out])
        _chk(lib().mi_fvm_assemble(self.addr.h, C.byref(t), _ptr(lower_out), _ptr(upper_out), _ptr(diag_out), so, _ptr(sum_mag_out)))

    def fvc_div(self, face_flux, weights, vf, vol, face_out, div_out):
        """fvc::div(faceFlux, vf) (gaussConvectionScheme.C:117-140); weights None: upwind"""
        _chk(lib().mi_fvc_div(self.addr.h, _ptr(face_flux), _ptr(weights), _ptr(vf), _ptr(vol), _ptr(face_out), _ptr(div_out)))

    def set_reference(self, celli, value, diag, source):
        """fvMatrix::setReference (fvMatrix.C:964-981)"""
        _chk(lib().mi_fvm_set_reference(self.addr.h, C.c_int32(celli), C.c_double(value), _ptr(diag), _ptr(source)))

    def set_values(self, cell_labels, values, psi, diag, source, upper_in, lower_in, upper_out, lower_out, patches=(), internal_coeffs=(), boundary_coeffs=(),
                   upstream=False):
        """fvMatrix::setValues (fvMatrix.C:454-656); cell_labels: int32 device tensor"""
        n = len(patches)
        ph = (C.c_void_p * max(n, 1))(*[p.h for p in patches])
        ic = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in internal_coeffs])
        bc = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in boundary_coeffs])
        assert cell_labels.is_contiguous() and cell_labels.element_size() == 4
        _chk(lib().mi_fvm_set_values(self.addr.h, C.c_int32(cell_labels.numel()), C.c_void_p(cell_labels.data_ptr()), _ptr(values), C.c_int32(int(upstream)),
                                     _ptr(psi), _ptr(diag), _ptr(source), _ptr(upper_in), _ptr(lower_in), _ptr(upper_out), _ptr(lower_out),
                                     C.c_int32(n), ph, ic, bc))

    def relax_multi(self, alpha, diag, lower, upper, sources, psis, sum_mag=None, patches=(), internal_coeffs=(), boundary_coeffs=(), coupled=()):
        """fvMatrix<Type>::relax for n_rhs components sharing the diagonal; sum_mag: mi_fvm_assemble's sumMagOffDiag by-product (completed in place)"""
        n = len(patches)
        ph = (C.c_void_p * max(n, 1))(*[p.h for p in patches])
        ic = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in internal_coeffs])
        bc = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in boundary_coeffs])
        cp = (C.c_int32 * max(n, 1))(*[int(v) for v in coupled])
        m = len(sources)
        sp = (C.c_void_p * max(m, 1))(*[_ptr(t) for t in sources])
        pp = (C.c_void_p * max(m, 1))(*[_ptr(t) for t in psis])
        _chk(lib().mi_relax_multi(self.addr.h, C.c_double(alpha), _ptr(diag), _ptr(lower), _ptr(upper), _ptr(sum_mag), C.c_int32(m), sp, pp,
                                  C.c_int32(n), ph, ic, bc, cp))

    def relax(self, alpha, diag, lower, upper, source, psi, patches=(), internal_coeffs=(), boundary_coeffs=(), coupled=()):
        n = len(patches)
        ph = (C.c_void_p * max(n, 1))(*[p.h for p in patches])
        ic = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in internal_coeffs])
        bc = (C.c_void_p * max(n, 1))(*[_ptr(t) for t in boundary_coeffs])
        cp = (C.c_int32 * max(n, 1))(*[int(v) for v in coupled])
        _chk(lib().mi_relax(self.addr.h, C.c_double(alpha), _ptr(diag), _ptr(lower), _ptr(upper), _ptr(source), _ptr(psi),
                            C.c_int32(n), ph, ic, bc, cp))
