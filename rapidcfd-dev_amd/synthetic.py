"""Synthetic hex-box lduMatrix inputs (SURVEY.md section 8d "synthetic inputs").

The reference ships no meshes or tutorials, so tests and bench.py build their
inputs here: a structured ``nx x ny x nz`` box with cells numbered
lexicographically (``c = i + nx*(j + ny*k)``) and internal faces in OpenFOAM's
upper-triangular order (for every cell ascending: its +x, +y, +z neighbours),
so that ``lowerAddr`` is sorted and ``lowerAddr[f] < upperAddr[f]`` exactly as
``fvMeshLduAddressing`` hands them to ``lduMatrix``
(src/finiteVolume/fvMesh/fvMeshLduAddressing.H:91-121).

Coefficients follow OpenFOAM's sign convention: ``fvm::laplacian`` gives
``upper = +gamma*|Sf|*deltaCoeffs`` and ``diag = -sum(offdiag)``
(gaussLaplacianScheme.C:63-64 + lduMatrix::negSumDiag), plus the fixedValue
``internalCoeffs = -2h`` on the x-min patch, which is what
``fvMatrix::addBoundaryDiag`` (fvMatrix.C:208-226) folds into the diagonal
before the solver is called.

All random numbers come from a splitmix64 counter hash so that every consumer
(numpy here, C in the oracle, tests on any box) sees identical bits.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_U64 = np.uint64


def splitmix_uniform(seed: int, n: int, start: int = 0) -> np.ndarray:
    """``n`` doubles in [0,1): u[i] = splitmix64(seed + (start+i)*0x9E37..15) >> 11 * 2^-53."""
    with np.errstate(over="ignore"):
        z = (np.arange(start, start + n, dtype=np.uint64) + _U64(1)) * _U64(0x9E3779B97F4A7C15)
        z = z + _U64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        z = z ^ (z >> _U64(31))
    return (z >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def splitmix_at(seed: int, idx: np.ndarray) -> np.ndarray:
    """splitmix_uniform(seed, .)[idx] without generating the whole sequence"""
    with np.errstate(over="ignore"):
        z = (np.asarray(idx).astype(np.uint64) + _U64(1)) * _U64(0x9E3779B97F4A7C15)
        z = z + _U64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U64(27))) * _U64(0x94D049BB133111EB)
        z = z ^ (z >> _U64(31))
    return (z >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class Interface:
    """One processor patch of a sub-domain (processorLduInterface + coeffs)."""
    nbr_domain: int
    nbr_patch: int
    face_cells: np.ndarray  # int32 [Pf]
    bou_coeffs: np.ndarray  # float64 [Pf]  interfaceBouCoeffs
    int_coeffs: np.ndarray  # float64 [Pf]  interfaceIntCoeffs
    # cyclicAMI (cyclicAMIFvPatchField): weights into the faces of the neighbour interface; None = one face to one face
    ami_start: Optional[np.ndarray] = None   # int32 [Pf+1]
    ami_addr: Optional[np.ndarray] = None    # int32 [ami_start[-1]] face index in the neighbour interface
    ami_w: Optional[np.ndarray] = None       # float64
    ami_low: Optional[np.ndarray] = None     # uint8 [Pf]: weight sum under lowWeightCorrection -> the face's own cell value
    ami_magsf: Optional[np.ndarray] = None   # float64 [Pf] face areas of this side (GAMG agglomerates the AMI with them)
    transform: float = 1.0                   # transformCoupleField factor (rotational cyclic, component solves)
    # cyclicAMI whose partner interface lives in ANOTHER domain (decompose_cyclic_ami_y): the partner's face count and, for the
    # engine's transport patch, the index that patch has in the partner rank's engine patch list
    ami_partner_size: Optional[int] = None
    ami_transport_nbr_patch: Optional[int] = None
    # cyclicAMI whose partner SIDE is split over several domains (decompose_cyclic_ami_split; the reference's distributed AMI,
    # AMIInterpolation.C:940-1091): ami_addr numbers the faces of the interfaces ami_parts[q] = (domain, interface) concatenated
    # in this order, piece q holding ami_part_sizes[q] faces; the engine gets one transport patch per piece, whose index in the
    # PARTNER rank's engine patch list is ami_transport_nbr_patches[q]
    ami_parts: Optional[list] = None
    ami_part_sizes: Optional[list] = None
    ami_transport_nbr_patches: Optional[list] = None


@dataclass
class LduCase:
    """Host-side LDU matrix + addressing in the caller's (OpenFOAM) order."""
    n_cells: int
    lower_addr: np.ndarray  # int32 [F] owner
    upper_addr: np.ndarray  # int32 [F] neighbour
    diag: np.ndarray
    upper: np.ndarray
    lower: Optional[np.ndarray]  # None => symmetric
    source: np.ndarray
    dims: tuple = ()
    interfaces: List[Interface] = field(default_factory=list)
    global_cells: Optional[np.ndarray] = None  # for decomposed cases: global cell id of each local cell
    global_faces: Optional[np.ndarray] = None  # ... and global face id of each local internal face

    @property
    def n_faces(self) -> int:
        return int(self.lower_addr.shape[0])

    @property
    def symmetric(self) -> bool:
        return self.lower is None


def box_addressing(nx: int, ny: int, nz: int):
    """(lowerAddr, upperAddr, direction) of the internal faces in OpenFOAM order: owner-major, every cell's faces in the order
    +x, +y, +z.  (Positions by a prefix sum over the cells' face counts and three strided fills: 1 s at 10 M cells where the
    masked (n, 3) int64 table of rounds 1-4 took 11 s -- the same arrays, tests/test_layout.py pins them against that form.)"""
    n = nx * ny * nz
    c = np.arange(n, dtype=np.int32)
    i = c % np.int32(nx)
    jk = c // np.int32(nx)
    j = jk % np.int32(ny)
    k = jk // np.int32(ny)
    hx, hy, hz = i < nx - 1, j < ny - 1, k < nz - 1
    del i, j, k, jk
    start = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(hx.astype(np.int8) + hy.astype(np.int8) + hz.astype(np.int8), dtype=np.int64, out=start[1:])
    nf = int(start[-1])
    own, nei, direction = np.empty(nf, dtype=np.int32), np.empty(nf, dtype=np.int32), np.empty(nf, dtype=np.int8)
    pos = start[:-1]
    for d, (has, step) in enumerate(((hx, 1), (hy, nx), (hz, nx * ny))):
        q, cells = pos[has], c[has]
        own[q] = cells
        nei[q] = cells + np.int32(step)
        direction[q] = d
        if d < 2:
            pos = pos + has
    return own, nei, direction


def box_addressing_reference(nx: int, ny: int, nz: int):
    """the rounds 1-4 form of box_addressing (a masked (n, 3) neighbour table), kept as the statement the fast form is tested against"""
    n = nx * ny * nz
    c = np.arange(n, dtype=np.int64)
    i = c % nx
    j = (c // nx) % ny
    k = c // (nx * ny)
    nbr = np.full((n, 3), -1, dtype=np.int64)
    nbr[i < nx - 1, 0] = c[i < nx - 1] + 1
    nbr[j < ny - 1, 1] = c[j < ny - 1] + nx
    nbr[k < nz - 1, 2] = c[k < nz - 1] + nx * ny
    valid = nbr >= 0
    own = np.broadcast_to(c[:, None], (n, 3))[valid]
    nei = nbr[valid]
    direction = np.broadcast_to(np.arange(3)[None, :], (n, 3))[valid]
    return own.astype(np.int32), nei.astype(np.int32), direction.astype(np.int8)


def box_case(nx: int, ny: int, nz: int, *, symmetric: bool = True, vary: float = 0.1,
             seed: int = 12345, rhs_seed: int = 777, dirichlet_all: bool = False) -> LduCase:
    """Pressure-like (symmetric) or momentum-like (asymmetric) matrix on a hex box."""
    n = nx * ny * nz
    lo, up, direction = box_addressing(nx, ny, nz)
    nf = lo.shape[0]
    h = 1.0 / nx
    u = splitmix_uniform(seed, nf)
    c = np.arange(n, dtype=np.int64)
    i = c % nx
    j = (c // nx) % ny
    k = c // (nx * ny)
    if symmetric:
        upper = h * (1.0 + vary * u)
        lower = None
        lo_c = upper
    else:
        # -nu*laplacian + upwind convection (flux 0.3 h^2 in +x) + ddt; positive diagonal
        nu_h = h * (1.0 + vary * u)
        phi = np.where(direction == 0, 0.3 * h * h, 0.0)
        upper = -nu_h
        lower = -nu_h - phi
        lo_c = lower
    # negSumDiag (lduMatrixOperations.C:62-83): diag[l] -= lower ; diag[u] -= upper
    diag = -(np.bincount(lo, weights=lo_c, minlength=n) + np.bincount(up, weights=upper, minlength=n)).astype(np.float64)
    sign = -1.0 if symmetric else 1.0
    bnd = (i == 0)
    if dirichlet_all:
        nb = ((i == 0).astype(np.int64) + (i == nx - 1) + (j == 0) + (j == ny - 1)
              + (k == 0) + (k == nz - 1))
        diag += sign * 2.0 * h * nb
    else:
        diag[bnd] += sign * 2.0 * h
    if not symmetric:
        diag += h ** 3 / 1e-3  # V/deltaT
    source = (2.0 * splitmix_uniform(rhs_seed, n) - 1.0) * h ** 3
    return LduCase(n, lo, up, diag, upper, lower, source, dims=(nx, ny, nz))


def decompose(case: LduCase, dom, n_domains: Optional[int] = None) -> List[LduCase]:
    """Split ANY case into sub-domains with processor interfaces, given the domain of every cell (what a
    ``decomposePar`` cellDecomposition file holds).

    Mirrors what ``decomposePar`` produces for the solver: every cut face
    becomes a face of a processor patch whose ``boundaryCoeffs`` hold minus
    the off-diagonal coefficient seen from this side and whose
    ``internalCoeffs`` hold minus the one seen from the other side, so that
    ``Apsi[faceCells] -= bouCoeffs*psiNbr`` (coupledFvPatchField.C:236-257)
    reproduces the undecomposed product.  Local cells and faces keep their
    relative global order (what decomposePar does), so local addressing is
    again upper-triangular and owner-sorted; one patch per neighbouring domain, ordered by domain number, its faces in
    global face order on both sides.  Domains may be empty of cut faces but not of cells.
    """
    n = case.n_cells
    dom = np.asarray(dom, dtype=np.int64)
    assert dom.shape[0] == n and dom.min() >= 0
    nd = int(dom.max()) + 1 if n_domains is None else int(n_domains)
    lo = case.lower_addr.astype(np.int64)
    up = case.upper_addr.astype(np.int64)
    lower_c = case.upper if case.lower is None else case.lower
    local_id = np.empty(n, dtype=np.int64)
    cells_of = []
    for d in range(nd):
        ids = np.nonzero(dom == d)[0]
        assert ids.shape[0] > 0, f"domain {d} has no cells"
        cells_of.append(ids)
        local_id[ids] = np.arange(ids.shape[0])
    dl, du = dom[lo], dom[up]
    out: List[LduCase] = []
    # cut faces grouped per ordered domain pair
    cut = np.nonzero(dl != du)[0]
    pair_faces = {}
    for d in range(nd):
        pair_faces[d] = {}
    for a, b in sorted(set(zip(dl[cut].tolist(), du[cut].tolist()))):
        f = cut[(dl[cut] == a) & (du[cut] == b)]
        pair_faces[a].setdefault(b, []).append(("own", f))
        pair_faces[b].setdefault(a, []).append(("nei", f))
    patch_index = {d: {nb: idx for idx, nb in enumerate(sorted(pair_faces[d]))} for d in range(nd)}
    for d in range(nd):
        ids = cells_of[d]
        fint = np.nonzero((dl == d) & (du == d))[0]
        sub = LduCase(
            n_cells=int(ids.shape[0]),
            lower_addr=local_id[lo[fint]].astype(np.int32),
            upper_addr=local_id[up[fint]].astype(np.int32),
            diag=case.diag[ids].copy(),
            upper=case.upper[fint].copy(),
            lower=None if case.lower is None else case.lower[fint].copy(),
            source=case.source[ids].copy(),
            global_cells=ids.astype(np.int64),
        )
        sub.global_faces = fint.astype(np.int64)       # internal faces of the sub-domain in global numbering
        for nb in sorted(pair_faces[d]):
            fcs, bou, inte = [], [], []
            # faces ordered by global face id on both sides so the two patches match 1:1
            allf = np.concatenate([e[1] for e in pair_faces[d][nb]])
            side = np.concatenate([np.full(len(e[1]), e[0] == "own") for e in pair_faces[d][nb]])
            order = np.argsort(allf, kind="stable")
            allf, side = allf[order], side[order]
            for f, is_own in zip(allf.tolist(), side.tolist()):
                if is_own:   # this domain holds the owner: row lo[f], coefficient upper[f]
                    fcs.append(local_id[lo[f]]); bou.append(-case.upper[f]); inte.append(-lower_c[f])
                else:        # this domain holds the neighbour: row up[f], coefficient lower[f]
                    fcs.append(local_id[up[f]]); bou.append(-lower_c[f]); inte.append(-case.upper[f])
            sub.interfaces.append(Interface(
                nbr_domain=nb, nbr_patch=patch_index[nb][d],
                face_cells=np.asarray(fcs, dtype=np.int32),
                bou_coeffs=np.asarray(bou, dtype=np.float64),
                int_coeffs=np.asarray(inte, dtype=np.float64)))
        out.append(sub)
    return out


def decompose_box(case: LduCase, parts) -> List[LduCase]:
    """``decompose`` with decomposePar's `simple` method on a box: px*py*pz blocks."""
    nx, ny, nz = case.dims
    px, py, pz = parts
    c = np.arange(case.n_cells, dtype=np.int64)
    i = c % nx
    j = (c // nx) % ny
    k = c // (nx * ny)

    def chunk(idx, nd, p):
        b = (np.arange(p + 1) * nd) // p
        return np.searchsorted(b, idx, side="right") - 1

    dom = chunk(i, nx, px) + px * (chunk(j, ny, py) + py * chunk(k, nz, pz))
    subs = decompose(case, dom, px * py * pz)
    for sub in subs:
        ids = sub.global_cells
        sub.dims = (int(i[ids].max() - i[ids].min() + 1), int(j[ids].max() - j[ids].min() + 1), int(k[ids].max() - k[ids].min() + 1))
    return subs


def add_cyclic_x(case: LduCase, kappa_scale: float = 1.0) -> LduCase:
    """The box made periodic in x (a channel's stream-wise cyclic pair, BASELINE config 4): the x-min and x-max cells become
    a cyclic patch pair, face q of one side coupled with face q of the other (both enumerate (j, k) in the same order);
    symmetric coefficients as add_cyclic_y.  The x-min Dirichlet contribution of box_case stays in the diagonal."""
    import copy
    nx, ny, nz = case.dims
    c = np.arange(case.n_cells, dtype=np.int64)
    i = c % nx
    xmin = np.nonzero(i == 0)[0].astype(np.int32)
    xmax = np.nonzero(i == nx - 1)[0].astype(np.int32)
    h = 1.0 / nx
    sign = 1.0 if case.lower is None else -1.0
    kappa = sign * h * kappa_scale * np.ones(xmin.shape[0])
    out = copy.copy(case)
    out.diag = case.diag.copy()
    np.subtract.at(out.diag, xmin, kappa)
    np.subtract.at(out.diag, xmax, kappa)
    out.interfaces = [Interface(nbr_domain=0, nbr_patch=1, face_cells=xmin, bou_coeffs=-kappa, int_coeffs=-kappa),
                      Interface(nbr_domain=0, nbr_patch=0, face_cells=xmax, bou_coeffs=-kappa, int_coeffs=-kappa)]
    return out


def add_cyclic_y(case: LduCase, kappa_scale: float = 1.0, asym_shift: float = 0.0) -> LduCase:
    """Make the box periodic in y: the y-min and y-max boundary patches become a cyclic pair
    (cyclicFvPatchField / cyclicLduInterfaceField: the neighbour values are local cells).  Face i of the
    y-min patch couples with face i of the y-max patch; coupling kappa = h*kappa_scale enters the diagonal
    (internalCoeffs, what addBoundaryDiag folds in) and the interface coefficients (boundaryCoeffs = -offdiag).
    Returned case: same addressing, two interfaces that reference each other inside domain 0."""
    import copy
    nx, ny, nz = case.dims
    c = np.arange(case.n_cells, dtype=np.int64)
    j = (c // nx) % ny
    ymin = np.nonzero(j == 0)[0].astype(np.int32)
    ymax = np.nonzero(j == ny - 1)[0].astype(np.int32)
    h = 1.0 / nx
    sign = 1.0 if case.lower is None else -1.0      # offdiag sign of the base matrix (sym: +h, asym: -nu*h)
    kappa = sign * h * kappa_scale * np.ones(ymin.shape[0])
    out = copy.copy(case)
    out.diag = case.diag.copy()
    np.subtract.at(out.diag, ymin, kappa)
    np.subtract.at(out.diag, ymax, kappa)
    out.interfaces = [
        Interface(nbr_domain=0, nbr_patch=1, face_cells=ymin, bou_coeffs=-kappa, int_coeffs=-(kappa - sign * asym_shift * h)),
        Interface(nbr_domain=0, nbr_patch=0, face_cells=ymax, bou_coeffs=-(kappa - sign * asym_shift * h), int_coeffs=-kappa),
    ]
    return out


def add_cyclic_ami_y(case: LduCase, shift: float = 0.37, low_weight_every: int = 0, transform: float = 1.0, seed: int = 71) -> LduCase:
    """Couple the y-min and y-max boundary patches of the box through a NON-CONFORMAL interface (cyclicAMI): the y-max patch
    is the y-min patch refined 1:2 in x and shifted by `shift` cells, so a y-min face overlaps up to three y-max faces and the
    two sides have different sizes (nx*nz against 2*nx*nz: the y-max cells carry two patch faces each).  Weights are the
    overlap fractions (area-normalised, AMIInterpolation::normaliseWeights), with a little seeded noise so that they are not
    exactly representable; low_weight_every > 0 marks every n-th face of each side as a low-weight face."""
    import copy
    nx, ny, nz = case.dims
    c = np.arange(case.n_cells, dtype=np.int64)
    j = (c // nx) % ny
    ymin = np.nonzero(j == 0)[0].astype(np.int32)                      # face f -> cell (i, 0, k), f = k*nx + i
    ymax_cells = np.nonzero(j == ny - 1)[0].astype(np.int32)
    ymax = np.repeat(ymax_cells, 2).astype(np.int32)                   # two half-width faces per cell: f = 2*(k*nx + i) + half
    h = 1.0 / nx
    sign = 1.0 if case.lower is None else -1.0

    def overlaps(x0, x1, width, count):
        """cells [m*width, (m+1)*width) of a periodic row of `count` cells overlapped by [x0, x1): (index, length) list"""
        out = []
        m = int(np.floor(x0 / width))
        while m * width < x1 - 1e-14:
            lo, hi = max(x0, m * width), min(x1, (m + 1) * width)
            if hi - lo > 1e-14:
                out.append((m % count, hi - lo))
            m += 1
        return out

    # source side (y-min, width 1) sees the shifted target row (width 0.5); target side sees the source row
    s_start, s_addr, s_w = [0], [], []
    for k in range(nz):
        for i in range(nx):
            ov = overlaps(i + shift, i + 1 + shift, 0.5, 2 * nx)
            for (m, ln) in ov:
                s_addr.append(2 * k * nx + m); s_w.append(ln / 1.0)
            s_start.append(len(s_addr))
    t_start, t_addr, t_w = [0], [], []
    for k in range(nz):
        for m in range(2 * nx):
            ov = overlaps(m * 0.5 - shift, (m + 1) * 0.5 - shift, 1.0, nx)
            for (i, ln) in ov:
                t_addr.append(k * nx + i); t_w.append(ln / 0.5)
            t_start.append(len(t_addr))
    s_w = np.array(s_w) * (1.0 + 1e-3 * (splitmix_uniform(seed, len(s_w)) - 0.5))
    t_w = np.array(t_w) * (1.0 + 1e-3 * (splitmix_uniform(seed + 1, len(t_w)) - 0.5))
    kap_s = sign * h * (0.8 + 0.4 * splitmix_uniform(seed + 2, ymin.shape[0]))
    kap_t = sign * 0.5 * h * (0.8 + 0.4 * splitmix_uniform(seed + 3, ymax.shape[0]))
    out = copy.copy(case)
    out.diag = case.diag.copy()
    np.subtract.at(out.diag, ymin, kap_s)
    np.subtract.at(out.diag, ymax, kap_t)
    low = lambda n: None if low_weight_every <= 0 else (np.arange(n) % low_weight_every == low_weight_every - 1).astype(np.uint8)
    ic = 1.0 if case.lower is None else 0.9               # interfaceIntCoeffs (Tmul) differ from interfaceBouCoeffs only for asymmetric matrices
    out.interfaces = [
        Interface(0, 1, ymin, -kap_s, -kap_s * ic, np.array(s_start, np.int32), np.array(s_addr, np.int32), s_w, low(ymin.shape[0]), transform),
        Interface(0, 0, ymax, -kap_t, -kap_t * ic, np.array(t_start, np.int32), np.array(t_addr, np.int32), t_w, low(ymax.shape[0]), transform),
    ]
    out.interfaces[0].ami_magsf = h * h * (1.0 + 0.05 * splitmix_uniform(seed + 4, ymin.shape[0]))
    out.interfaces[1].ami_magsf = 0.5 * h * h * (1.0 + 0.05 * splitmix_uniform(seed + 5, ymax.shape[0]))
    return out


def decompose_cyclic_ami_y(base: LduCase, py: int, **ami_kw) -> List[LduCase]:
    """The box with the non-conformal y-min / y-max interface of add_cyclic_ami_y, cut into `py` slabs in y: the two sides of
    the cyclicAMI pair end up on DIFFERENT ranks (slab 0 holds y-min, slab py-1 holds y-max; the reference's distributed AMI,
    AMIInterpolation.C:940-1091), with ordinary processor patches between neighbouring slabs.  Every sub-domain: processor
    interfaces first (as decompose_box orders them), the AMI interface last; its ami_addr numbers the faces of the partner
    interface, which lives in domain nbr_domain.  The multi-domain oracle takes the list as it is; the engine adds one
    transport patch per remote AMI interface (parallel.DistributedMatrix)."""
    import copy
    full = add_cyclic_ami_y(base, **ami_kw)
    bare = copy.copy(full); bare.interfaces = []
    subs = decompose_box(bare, (1, py, 1))
    a, b = full.interfaces
    first, last = subs[0], subs[-1]
    loc = lambda sub, cells: np.searchsorted(sub.global_cells, cells).astype(np.int32)
    na, nb = len(first.interfaces), len(last.interfaces)
    if py == 1:
        ia, ib = copy.copy(a), copy.copy(b)
        ia.nbr_patch, ib.nbr_patch = na + 1, na
        first.interfaces += [ia, ib]
        return subs
    ia, ib = copy.copy(a), copy.copy(b)
    ia.face_cells, ib.face_cells = loc(first, a.face_cells), loc(last, b.face_cells)
    ia.nbr_domain, ia.nbr_patch, ia.ami_partner_size = py - 1, nb, len(b.face_cells)
    ib.nbr_domain, ib.nbr_patch, ib.ami_partner_size = 0, na, len(a.face_cells)
    # engine patch lists: the interfaces in order, then one transport patch per remote AMI interface
    ia.ami_transport_nbr_patch = nb + 1
    ib.ami_transport_nbr_patch = na + 1
    first.interfaces.append(ia); last.interfaces.append(ib)
    return subs


def decompose_cyclic_ami_split(base: LduCase, px: int, **ami_kw) -> List[LduCase]:
    """The box with the non-conformal y-min / y-max interface of add_cyclic_ami_y, cut into px x 2 blocks in x and y: BOTH sides of
    the cyclicAMI pair are split over px ranks each (ranks 0 .. px-1 hold the pieces of the y-min side, px .. 2 px - 1 those of the
    y-max side), and -- the y-max side being shifted and periodic in x -- every piece overlaps faces of SEVERAL partner pieces:
    the reference's distributed AMI in full (singlePatchProc_ == -1, AMIInterpolation.C:940-1091 calcProcMap: a rank's source
    faces interpolate from target faces that arrive from all ranks, numbered rank by rank).  Every sub-domain: processor interfaces
    first (as decompose_box orders them), its piece of the AMI patch last, with
        ami_parts       the partner pieces it addresses, as (domain, interface index), ascending in domain,
        ami_addr        numbering those pieces' faces concatenated in that order (a piece's faces keep the global face order),
        ami_part_sizes  their face counts.
    The multi-domain oracle takes the list as it is; the engine adds one transport patch per partner piece
    (parallel.DistributedMatrix, mi_addr_set_ami_patch_remote_multi).  Partner lists are symmetric: piece P lists piece Q when P
    addresses Q or Q addresses P (a transport patch carries both directions)."""
    import copy
    full = add_cyclic_ami_y(base, **ami_kw)
    bare = copy.copy(full); bare.interfaces = []
    subs = decompose_box(bare, (px, 2, 1))
    sides = full.interfaces                                   # [y-min side, y-max side]
    n_dom = len(subs)
    owner = np.empty(base.n_cells, dtype=np.int64); local = np.empty(base.n_cells, dtype=np.int64)
    for d, sub in enumerate(subs):
        owner[sub.global_cells] = d; local[sub.global_cells] = np.arange(sub.n_cells)
    # pieces: faces of a side held by each domain, in global face order
    piece_faces = [dict(), dict()]                            # side -> {domain: global face ids}
    face_dom, face_pos = [None, None], [None, None]
    for sd, itf in enumerate(sides):
        fd = owner[itf.face_cells]
        face_dom[sd] = fd
        pos = np.empty(len(fd), dtype=np.int64)
        for d in sorted(set(fd.tolist())):
            ids = np.nonzero(fd == d)[0]
            piece_faces[sd][d] = ids; pos[ids] = np.arange(len(ids))
        face_pos[sd] = pos
    n_proc = [len(sub.interfaces) for sub in subs]            # the AMI piece of domain d is its interface number n_proc[d]
    # who talks to whom (symmetric)
    talks = {d: set() for d in range(n_dom)}
    for sd, itf in enumerate(sides):
        other = 1 - sd
        for d, ids in piece_faces[sd].items():
            for i in ids:
                for k in range(itf.ami_start[i], itf.ami_start[i + 1]):
                    q = int(face_dom[other][itf.ami_addr[k]])
                    talks[d].add(q); talks[q].add(d)
    partners = {d: sorted(talks[d]) for d in range(n_dom)}
    for sd, itf in enumerate(sides):
        other = 1 - sd
        for d, ids in piece_faces[sd].items():
            parts = partners[d]
            sizes = [len(piece_faces[other][q]) for q in parts]
            start_of = dict(zip(parts, np.concatenate([[0], np.cumsum(sizes)])[:-1].tolist()))
            st, ad, ww = [0], [], []
            for i in ids:
                for k in range(itf.ami_start[i], itf.ami_start[i + 1]):
                    j = int(itf.ami_addr[k])
                    ad.append(start_of[int(face_dom[other][j])] + int(face_pos[other][j])); ww.append(itf.ami_w[k])
                st.append(len(ad))
            piece = Interface(nbr_domain=parts[0], nbr_patch=n_proc[parts[0]], face_cells=local[itf.face_cells[ids]].astype(np.int32),
                              bou_coeffs=itf.bou_coeffs[ids].copy(), int_coeffs=itf.int_coeffs[ids].copy(), ami_start=np.array(st, np.int32),
                              ami_addr=np.array(ad, np.int32), ami_w=np.array(ww, np.float64), ami_low=None if itf.ami_low is None else itf.ami_low[ids].copy(),
                              ami_magsf=None if itf.ami_magsf is None else itf.ami_magsf[ids].copy(), transform=itf.transform)
            piece.ami_parts = [(q, n_proc[q]) for q in parts]
            piece.ami_part_sizes = sizes
            # engine patch list of a rank: processor interfaces, the AMI piece, then one transport per partner (ascending rank)
            piece.ami_transport_nbr_patches = [n_proc[q] + 1 + partners[q].index(d) for q in parts]
            subs[d].interfaces.append(piece)
    return subs


def _chunks(nd: int, p: int) -> np.ndarray:
    return (np.arange(p + 1) * nd) // p


def _global_face_index(c, i, j, k, direction, dims):
    """index of the internal face (owner c, +direction) in box_addressing's order, in closed form"""
    nx, ny, nz = dims
    before = 3 * c - c // nx - (k * nx + np.where(j == ny - 1, i, 0)) - np.where(k == nz - 1, c - (nz - 1) * nx * ny, 0)
    has_x, has_y = (i < nx - 1).astype(np.int64), (j < ny - 1).astype(np.int64)
    return before + np.where(direction == 0, 0, np.where(direction == 1, has_x, has_x + has_y))


def box_subdomain(global_dims, parts, rank: int, *, symmetric: bool = True, vary: float = 0.1, seed: int = 12345, rhs_seed: int = 777) -> LduCase:
    """Sub-domain ``rank`` of decompose_box(box_case(*global_dims, symmetric=symmetric), parts) built DIRECTLY: the global
    case is never formed, so an 8 x 216^3 weak-scaling run costs every rank only its own 10 M cells.  Identical, array
    for array, to the decomposed global case (tests/test_distributed.py), pressure-like and momentum-like."""
    nx, ny, nz = global_dims
    px, py, pz = parts
    bx, by, bz = _chunks(nx, px), _chunks(ny, py), _chunks(nz, pz)
    rx, ry, rz = rank % px, (rank // px) % py, rank // (px * py)
    i0, i1, j0, j1, k0, k1 = bx[rx], bx[rx + 1], by[ry], by[ry + 1], bz[rz], bz[rz + 1]
    lx, ly, lz = int(i1 - i0), int(j1 - j0), int(k1 - k0)
    n = lx * ly * lz
    h = 1.0 / nx
    lo, up, direction = box_addressing(lx, ly, lz)
    c = np.arange(n, dtype=np.int64)
    gi, gj, gk = c % lx + i0, (c // lx) % ly + j0, c // (lx * ly) + k0
    gc = gi + nx * (gj + ny * gk)

    def coef(owner_local, d):   # coefficient of the global face (owner, +d)
        o = owner_local
        f = _global_face_index(gc[o], gi[o], gj[o], gk[o], d, (nx, ny, nz))
        return h * (1.0 + vary * splitmix_at(seed, f))

    upper = coef(lo.astype(np.int64), direction.astype(np.int64))
    lower = None
    if symmetric:
        diag = -(np.bincount(lo, weights=upper, minlength=n) + np.bincount(up, weights=upper, minlength=n)).astype(np.float64)
        diag[gi == 0] += -2.0 * h
    else:   # box_case's momentum-like matrix: -nu laplacian + upwind convection (flux 0.3 h^2 in +x) + ddt
        nu_h = upper
        upper = -nu_h
        lower = -nu_h - np.where(direction == 0, 0.3 * h * h, 0.0)
        diag = -(np.bincount(lo, weights=lower, minlength=n) + np.bincount(up, weights=upper, minlength=n)).astype(np.float64)
    source = (2.0 * splitmix_at(rhs_seed, gc) - 1.0) * h ** 3
    sub = LduCase(n, lo, up, diag, upper, lower, source, dims=(lx, ly, lz), global_cells=gc)
    # processor patches, ordered by neighbour rank (as decompose_box orders them); faces by global face id
    def nbr_list(qx, qy, qz):
        out = []
        for dz, dy, dx in ((-1, 0, 0), (0, -1, 0), (0, 0, -1), (0, 0, 1), (0, 1, 0), (1, 0, 0)):
            ax, ay, az = qx + dx, qy + dy, qz + dz
            if 0 <= ax < px and 0 <= ay < py and 0 <= az < pz:
                out.append(ax + px * (ay + py * az))
        return sorted(out)
    mine = nbr_list(rx, ry, rz)
    for nb in mine:
        ax, ay, az = nb % px, (nb // px) % py, nb // (px * py)
        d = 0 if ax != rx else (1 if ay != ry else 2)
        plus = (ax > rx) if d == 0 else ((ay > ry) if d == 1 else (az > rz))   # neighbour on the + side of this block
        if d == 0:
            cells = c[(gi == (i1 - 1 if plus else i0))]
        elif d == 1:
            cells = c[(gj == (j1 - 1 if plus else j0))]
        else:
            cells = c[(gk == (k1 - 1 if plus else k0))]
        # the owner of a cut face is the cell on the lower side; its global face index orders the patch
        if plus:
            fidx = _global_face_index(gc[cells], gi[cells], gj[cells], gk[cells], d, (nx, ny, nz))
        else:
            step = (1, nx, nx * ny)[d]
            oc = gc[cells] - step
            oi, oj, ok = gi[cells] - (d == 0), gj[cells] - (d == 1), gk[cells] - (d == 2)
            fidx = _global_face_index(oc, oi, oj, ok, d, (nx, ny, nz))
        order = np.argsort(fidx, kind="stable")
        cells, fidx = cells[order], fidx[order]
        cf = h * (1.0 + vary * splitmix_at(seed, fidx))
        theirs = nbr_list(ax, ay, az)
        if symmetric:
            np.subtract.at(sub.diag, cells, cf)
            sub.interfaces.append(Interface(nbr_domain=nb, nbr_patch=theirs.index(rank), face_cells=cells.astype(np.int32),
                                            bou_coeffs=-cf, int_coeffs=-cf))
        else:
            up_f = -cf
            lo_f = -cf - (0.3 * h * h if d == 0 else 0.0)
            # this block holds the owner of the cut face (neighbour on the + side): row coefficient upper, diag -= lower;
            # else it holds the neighbour: row coefficient lower, diag -= upper (negSumDiag on the undivided mesh)
            mine_c, other_c = (up_f, lo_f) if plus else (lo_f, up_f)
            np.subtract.at(sub.diag, cells, other_c)
            sub.interfaces.append(Interface(nbr_domain=nb, nbr_patch=theirs.index(rank), face_cells=cells.astype(np.int32),
                                            bou_coeffs=-mine_c, int_coeffs=-other_c))
    if not symmetric:
        sub.diag[gi == 0] += 2.0 * h
        sub.diag += h ** 3 / 1e-3   # V/deltaT
    return sub


def renumber(case: LduCase, new_to_old) -> LduCase:
    """The case after ``renumberMesh`` with cell map ``new_to_old`` (new cell i = old cell new_to_old[i], e.g. the engine
    order ``mi_addr_cell_perm`` proposes): cells permuted, every face keeps owner < neighbour (a face whose cells swap
    order is flipped: its lower and upper coefficients swap roles), faces re-sorted into OpenFOAM's upper-triangular order
    (polyMesh::renumber / renumberMesh.C do exactly this to the mesh; the matrix follows).  Interfaces keep their face order."""
    new_to_old = np.asarray(new_to_old, dtype=np.int64)
    n = case.n_cells
    old_to_new = np.empty(n, dtype=np.int64)
    old_to_new[new_to_old] = np.arange(n)
    lo, up = old_to_new[case.lower_addr], old_to_new[case.upper_addr]
    flip = lo > up
    nlo, nup = np.where(flip, up, lo), np.where(flip, lo, up)
    lower_c = case.upper if case.lower is None else case.lower
    nupper = np.where(flip, lower_c, case.upper)
    nlower = np.where(flip, case.upper, lower_c)
    order = np.lexsort((nup, nlo))                      # owner-sorted, then by neighbour: upper-triangular order
    out = LduCase(n_cells=n, lower_addr=nlo[order].astype(np.int32), upper_addr=nup[order].astype(np.int32), diag=case.diag[new_to_old].copy(),
                  upper=nupper[order].copy(), lower=None if case.lower is None else nlower[order].copy(), source=case.source[new_to_old].copy(),
                  dims=case.dims)
    out.global_cells = new_to_old.copy()                # old cell of every new cell
    out.global_faces = order.astype(np.int64)           # old face of every new face
    out.face_flipped = flip[order]
    for itf in case.interfaces:
        import copy
        j = copy.copy(itf)
        j.face_cells = old_to_new[itf.face_cells].astype(np.int32)
        out.interfaces.append(j)
    return out
