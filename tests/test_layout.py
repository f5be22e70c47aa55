"""CPU: the host-side tiling (rapidcfd-dev_amd/csrc/tiling.cpp) is verified by interpreting
its tables in numpy -- exactly what tile_kernel does on the GPU -- and comparing with the oracle."""
import numpy as np
import pytest

from conftest import random_graph_case


def _interpret(L, case, x, ext=None, bou=None, transpose=False, compact=False):
    """numpy re-enactment of tile_kernel<OP_AMUL> over the layout tables (explicit 32-bit entries, or the
    compact 16-bit entries whose slots are implied by the slot order)."""
    if compact:
        assert L["entries16"].shape[0] > 0 or case.n_faces == 0, "layout has no compact form"
    n = case.n_cells
    e2c = L["e2c"]
    xe = np.concatenate([x[e2c], ext if ext is not None else np.zeros(0)])
    slot_face = L["slotFace"]
    up = np.zeros(slot_face.shape[0]); lo = np.zeros(slot_face.shape[0])
    internal = slot_face >= 0
    lower_c = case.upper if case.lower is None else case.lower
    up[internal] = case.upper[slot_face[internal]]
    lo[internal] = lower_c[slot_face[internal]]
    if bou is not None:
        up[L["extSlot"]] = -bou
        lo[L["extSlot"]] = -bou
    ye = np.zeros(n)
    nT = L["tileCellStart"].shape[0] - 1
    for t in range(nT):
        c0, c1 = L["tileCellStart"][t], L["tileCellStart"][t + 1]
        s0 = L["tileSlotStart"][t]
        h0, h1 = L["tileHaloStart"][t], L["tileHaloStart"][t + 1]
        xs = np.concatenate([xe[c0:c1], xe[L["haloCell"][h0:h1]]])
        nc = c1 - c0
        nh = h1 - h0
        if compact:  # slot bases of cells, halo cells and the pad cell; the pad cell reads x = 0
            w0 = 2 * L["tileSbStart"][t]
            sb = L["slotBase"][w0: w0 + nc + nh + 1].astype(np.int64)
            xs = np.concatenate([xs, np.zeros(1)])
            ifs0 = L["tileIfaceSlot0"][t]
        for s in range(L["tileSliceStart"][t], L["tileSliceStart"][t + 1]):
            r0 = (s - L["tileSliceStart"][t]) * 64
            rows = np.arange(r0, min(r0 + 64, nc))
            acc = case.diag[e2c[c0 + rows]] * xs[rows]
            if compact:
                e0, e1 = L["sliceEntryStart16"][s], L["sliceEntryStart16"][s + 1]
                words = L["entries16"][e0:e1].reshape(-1, 64)
                for j in range(2 * words.shape[0]):
                    w = words[j // 2, : rows.shape[0]]
                    en = ((w >> 16) if (j & 1) else (w & 0xFFFF)).astype(np.int64)
                    o = en & 0xFFF
                    rule = (en >> 15).astype(bool)
                    sl = np.where(rule, sb[o] + ((en >> 12) & 7), sb[rows] + j)
                    is_low = rule & (sl < ifs0)
                    coef = np.where(is_low != transpose, lo[s0 + sl], up[s0 + sl])
                    acc = acc + coef * xs[o]
            else:
                e0, e1 = L["sliceEntryStart"][s], L["sliceEntryStart"][s + 1]
                ent = L["entries"][e0:e1].reshape(-1, 64)
                for j in range(ent.shape[0]):
                    en = ent[j, : rows.shape[0]]
                    o = (en & 0xFFFF).astype(np.int64)
                    sl = ((en >> 16) & 0x7FFF).astype(np.int64)
                    is_low = (en >> 31).astype(bool)
                    coef = np.where(is_low != transpose, lo[s0 + sl], up[s0 + sl])
                    acc = acc + coef * xs[o]
            ye[c0 + rows] = acc
    y = np.empty(n)
    y[e2c] = ye
    return y


def interpret_amul(L, case, x, **kw):
    """explicit form; when the layout also carries the compact form, that one must give the same bits"""
    y = _interpret(L, case, x, compact=False, **kw)
    if L["entries16"].shape[0] > 0:
        COMPACT_SEEN.append(1)
        assert np.array_equal(_interpret(L, case, x, compact=True, **kw), y)
    return y


COMPACT_SEEN = []


@pytest.mark.parametrize("tile_cells", [64, 1024])
@pytest.mark.parametrize("kind", ["box_sym", "box_asym", "graph_sym", "graph_asym", "tiny"])
def test_layout_reproduces_amul(pkg, orc, kind, tile_cells):
    syn, eng = pkg.synthetic, pkg.engine
    if kind == "tiny":
        case = syn.box_case(2, 1, 1)
    elif kind.startswith("box"):
        case = syn.box_case(13, 9, 7, symmetric=kind.endswith("_sym"))
    else:
        case = random_graph_case(pkg, 500, symmetric=kind.endswith("_sym"))
    L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, tile_cells=tile_cells)
    n = case.n_cells
    # permutation is a bijection, tiles partition the cells, every tile respects the cap
    assert np.array_equal(np.sort(L["e2c"]), np.arange(n))
    assert np.array_equal(L["c2e"][L["e2c"]], np.arange(n))
    sizes = np.diff(L["tileCellStart"])
    assert sizes.min() >= 1 and sizes.max() <= tile_cells and sizes.sum() == n
    assert np.all(L["tileSlotStart"] % 2 == 0)
    # every face owns at least one slot; internal faces exactly one or two (cut)
    cnt = np.bincount(L["slotFace"][L["slotFace"] >= 0], minlength=case.n_faces)
    assert cnt.min() >= 1 and cnt.max() <= 2
    assert np.array_equal(L["slotFace"][L["faceSlot"]], np.arange(case.n_faces))
    S = orc.System([case])
    x = syn.splitmix_uniform(21, n) - 0.5
    ref = S.amul(x)
    got = interpret_amul(L, case, x)
    assert np.max(np.abs(got - ref)) <= 4e-16 * np.max(np.abs(ref)) * 8
    gott = interpret_amul(L, case, x, transpose=True)
    assert np.max(np.abs(gott - S.tmul(x))) <= 4e-16 * np.max(np.abs(ref)) * 8


def test_single_cell(pkg):
    L = pkg.engine.host_layout(1, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert L["tileCellStart"].tolist() == [0, 1]
    assert L["entries"].shape[0] == 0


def test_tiles_are_compact_bricks_on_a_box(pkg):
    # multilevel heavy-edge matching must recover brick-shaped tiles on a lexicographic box:
    # that is what makes each symmetric coefficient be read once
    case = pkg.synthetic.box_case(32, 32, 32)
    L = pkg.engine.host_layout(case.n_cells, case.lower_addr, case.upper_addr)
    sizes = np.diff(L["tileCellStart"])
    assert sizes.min() == sizes.max() == 1024
    slots_per_row = np.diff(L["tileSlotStart"]).sum() / case.n_cells
    halo_per_row = np.diff(L["tileHaloStart"]).sum() / case.n_cells
    assert slots_per_row < 3.2 and halo_per_row < 0.5


def test_interfaces_become_boundary_tiles(pkg, orc):
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.box_case(12, 10, 8)
    parts = syn.decompose_box(case, (2, 1, 1))
    x = syn.splitmix_uniform(4, case.n_cells) - 0.5
    ref = orc.System([case]).amul(x)
    for d, sub in enumerate(parts):
        fcs = [itf.face_cells for itf in sub.interfaces]
        L = eng.host_layout(sub.n_cells, sub.lower_addr, sub.upper_addr, fcs, tile_cells=64)
        n_ext = sum(len(f) for f in fcs)
        assert L["patchOffset"][-1] == n_ext and L["extSlot"].shape[0] == n_ext
        assert len(L["interiorTiles"]) + len(L["boundaryTiles"]) == len(L["tileCellStart"]) - 1
        assert len(L["boundaryTiles"]) > 0 and len(L["interiorTiles"]) > 0
        # interior tiles never reference an ext cell
        for t in L["interiorTiles"]:
            h = L["haloCell"][L["tileHaloStart"][t]:L["tileHaloStart"][t + 1]]
            assert np.all(h < sub.n_cells)
        # halo values: the neighbour's psi at its matching patch cells
        ext = np.concatenate([x[parts[itf.nbr_domain].global_cells][parts[itf.nbr_domain].interfaces[itf.nbr_patch].face_cells]
                              for itf in sub.interfaces])
        bou = np.concatenate([itf.bou_coeffs for itf in sub.interfaces])
        got = interpret_amul(L, sub, x[sub.global_cells], ext=ext, bou=bou)
        assert np.max(np.abs(got - ref[sub.global_cells])) < 1e-15


def test_bad_addressing_is_rejected(pkg):
    with pytest.raises(pkg.engine.MiError):
        pkg.engine.host_layout(4, np.array([2], np.int32), np.array([1], np.int32))  # lower >= upper
    with pytest.raises(pkg.engine.MiError):
        pkg.engine.host_layout(4, np.array([0], np.int32), np.array([7], np.int32))  # out of range


def test_layout_property_random_graphs(pkg, orc):
    """property test: any LDU addressing (ragged rows, multi-degree hubs, odd tile caps) -> the interpreted
    tile tables reproduce A*x and A^T*x of the oracle."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=25, deadline=None)
    @given(n=st.integers(2, 400), extra=st.floats(0.0, 4.0), seed=st.integers(0, 10_000),
           tile=st.sampled_from([3, 17, 64, 200, 1024]), sym=st.booleans())
    def run(n, extra, seed, tile, sym):
        case = random_graph_case(pkg, n, extra=extra, seed=seed, symmetric=sym)
        L = pkg.engine.host_layout(case.n_cells, case.lower_addr, case.upper_addr, tile_cells=tile)
        assert np.diff(L["tileCellStart"]).max() <= tile
        S = orc.System([case])
        x = pkg.synthetic.splitmix_uniform(seed + 7, n) - 0.5
        ref = S.amul(x)
        tol = 1e-14 * max(np.max(np.abs(ref)), 1e-300)
        assert np.max(np.abs(interpret_amul(L, case, x) - ref)) <= tol
        assert np.max(np.abs(interpret_amul(L, case, x, transpose=True) - S.tmul(x))) <= tol

    run()


def test_hub_cell_with_many_faces(pkg, orc):
    # a star: one cell connected to 300 others (polyhedral "hub"): rows longer than the register-prefetch depth
    n = 301
    lo = np.zeros(n - 1, np.int32); up = np.arange(1, n, dtype=np.int32)
    syn = pkg.synthetic
    upper = -(0.1 + syn.splitmix_uniform(1, n - 1))
    diag = np.zeros(n); np.subtract.at(diag, lo, upper); np.subtract.at(diag, up, upper); diag += 0.5
    case = syn.LduCase(n, lo, up, diag, upper, None, syn.splitmix_uniform(2, n))
    L = pkg.engine.host_layout(n, lo, up, tile_cells=64)
    x = syn.splitmix_uniform(3, n)
    ref = orc.System([case]).amul(x)
    assert np.max(np.abs(interpret_amul(L, case, x) - ref)) < 1e-12


def test_cyclic_patches_are_local_couplings(pkg, orc):
    # cyclic pair (y-periodic box): interface slots point at local cells, no tile depends on the ext region
    syn, eng = pkg.synthetic, pkg.engine
    for symmetric in (True, False):
        case = syn.add_cyclic_y(syn.box_case(9, 6, 5, symmetric=symmetric), asym_shift=0.0 if symmetric else 0.3)
        fcs = [i.face_cells for i in case.interfaces]
        nbrs = [case.interfaces[i.nbr_patch].face_cells for i in case.interfaces]
        L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, fcs, tile_cells=64, patch_nbr_cells=nbrs)
        assert len(L["boundaryTiles"]) == 0
        assert np.all(L["haloCell"] < case.n_cells)
        # interface slots close every tile's slot segment
        for t in range(len(L["tileIfaceSlot0"])):
            s0, s1 = L["tileSlotStart"][t], L["tileSlotStart"][t + 1]
            sf = L["slotFace"][s0:s1]
            k = L["tileIfaceSlot0"][t]
            assert np.all(sf[:k] >= 0) and np.all(sf[k:] < 0)
        x = syn.splitmix_uniform(8, case.n_cells) - 0.5
        S = orc.System([case])
        bou = np.concatenate([i.bou_coeffs for i in case.interfaces])
        got = interpret_amul(L, case, x, ext=np.zeros(len(bou)), bou=bou)
        assert np.max(np.abs(got - S.amul(x))) < 1e-15


def test_layout_property_random_coupled_patches(pkg, orc):
    """property test: random graphs with a random cyclic patch pair AND a random processor patch (ext values supplied) ->
    the interpreted tables reproduce the oracle's Amul with interfaces; partners may fall in the same or another tile."""
    from hypothesis import given, settings, strategies as st
    syn = pkg.synthetic

    @settings(max_examples=20, deadline=None)
    @given(n=st.integers(8, 300), extra=st.floats(0.5, 3.0), seed=st.integers(0, 10_000), tile=st.sampled_from([5, 33, 128, 1024]),
           npair=st.integers(1, 12), nproc=st.integers(0, 9))
    def run(n, extra, seed, tile, npair, nproc):
        case = random_graph_case(pkg, n, extra=extra, seed=seed)
        u = syn.splitmix_uniform(seed + 11, 2 * npair + nproc)
        a = (u[:npair] * n).astype(np.int32); b = (u[npair:2 * npair] * n).astype(np.int32)
        pc = (u[2 * npair:] * n).astype(np.int32)
        kap = -(0.1 + syn.splitmix_uniform(seed + 12, npair))
        pb = -(0.1 + syn.splitmix_uniform(seed + 13, nproc))
        ext_vals = syn.splitmix_uniform(seed + 14, nproc) - 0.5
        import copy
        cs = copy.copy(case)
        cs.interfaces = [syn.Interface(0, 1, a, kap, kap), syn.Interface(0, 0, b, kap, kap)]
        # the oracle sees the processor patch as a second domain of `nproc` cells that holds the ext values
        S = None
        L = pkg.engine.host_layout(n, case.lower_addr, case.upper_addr, [a, b, pc], tile_cells=tile, patch_nbr_cells=[b, a, None])
        assert np.all(L["haloCell"] < n + 2 * npair + nproc)
        x = syn.splitmix_uniform(seed + 7, n) - 0.5
        ref = orc.System([cs]).amul(x)
        np.subtract.at(ref, pc, pb * ext_vals)           # result[faceCells] -= bouCoeffs * psi_neighbour
        ext = np.concatenate([np.zeros(2 * npair), ext_vals])
        got = interpret_amul(L, case, x, ext=ext, bou=np.concatenate([kap, kap, pb]))
        assert np.max(np.abs(got - ref)) <= 1e-13 * max(np.max(np.abs(ref)), 1e-300)
        # tiles that read the ext region are exactly the boundary tiles
        bt = set(L["boundaryTiles"].tolist())
        for t in range(len(L["tileCellStart"]) - 1):
            h = L["haloCell"][L["tileHaloStart"][t]:L["tileHaloStart"][t + 1]]
            assert (np.any(h >= n)) == (t in bt)

    run()


def test_compact_entries_are_built_for_boxes_and_not_for_hubs(pkg):
    """the 16-bit form exists for mesh-like graphs; a hub cell (more than 8 owned faces towards one tile) or a
    tile with more than 4095 cells+halo falls back to the explicit form"""
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.box_case(13, 9, 7)
    L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr)
    assert L["entries16"].shape[0] > 0 and L["slotBase"].shape[0] >= case.n_cells + L["haloCell"].shape[0] + (L["tileCellStart"].shape[0] - 1)
    assert 2 * L["entries16"].shape[0] <= L["entries"].shape[0] + 64 * (L["tileSliceStart"][-1])
    n = 40
    lower = np.zeros(n - 1, dtype=np.int32); upper = np.arange(1, n, dtype=np.int32)   # cell 0 owns 39 faces
    L = eng.host_layout(n, lower, upper)
    assert L["entries16"].shape[0] == 0 and L["entries"].shape[0] > 0


def test_compact_form_was_exercised_with_patches(pkg, orc):
    """cyclic + processor-like patches on a box: the compact form must exist and reproduce Amul (private halo
    entries name the interface slots)"""
    syn, eng = pkg.synthetic, pkg.engine
    case = syn.add_cyclic_y(syn.box_case(9, 8, 7, symmetric=False), asym_shift=0.05)
    fcs = [i.face_cells for i in case.interfaces]
    nbrs = [case.interfaces[i.nbr_patch].face_cells for i in case.interfaces]
    L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, fcs, tile_cells=128, patch_nbr_cells=nbrs)
    assert L["entries16"].shape[0] > 0
    x = syn.splitmix_uniform(4, case.n_cells) - 0.5
    bou = np.concatenate([i.bou_coeffs for i in case.interfaces])
    before = len(COMPACT_SEEN)
    got = interpret_amul(L, case, x, ext=np.zeros(len(bou)), bou=bou)
    assert len(COMPACT_SEEN) == before + 1
    ref = orc.System([case]).amul(x)
    assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref))


@pytest.mark.parametrize("kind", ["box", "box_cyclic", "graph"])
def test_layout_does_not_depend_on_the_callers_numbering(pkg, orc, monkeypatch, kind):
    """The clustering visits cells in index order, so a mesh numbered without locality used to get half-filled tiles (545
    cells on average for a randomly numbered 64^3 box, Amul 2.4x slower on the GPU).  When the numbering has no locality
    the clustering now runs on a Cuthill-McKee ordering of the cell graph: the shuffled box gets the bricks of the
    lexicographic one, and the tables still reproduce the oracle's Amul / Tmul on the caller's (shuffled) numbering."""
    syn, eng = pkg.synthetic, pkg.engine
    rng = np.random.default_rng(11)
    if kind == "graph":
        base = random_graph_case(pkg, 6000, extra=1.5, seed=4, symmetric=False)
    else:
        base = syn.box_case(32, 32, 32 if kind == "box" else 8, symmetric=(kind == "box"))
        if kind == "box_cyclic":
            base = syn.add_cyclic_y(base, asym_shift=0.25)
    case = syn.renumber(base, rng.permutation(base.n_cells).astype(np.int32))
    kw = {}
    if case.interfaces:
        kw = dict(patch_face_cells=[i.face_cells for i in case.interfaces], patch_nbr_cells=[case.interfaces[i.nbr_patch].face_cells for i in case.interfaces])
    kw0 = {}
    if base.interfaces:
        kw0 = dict(patch_face_cells=[i.face_cells for i in base.interfaces], patch_nbr_cells=[base.interfaces[i.nbr_patch].face_cells for i in base.interfaces])
    monkeypatch.setenv("MI_TILE_REORDER", "0")
    L_off = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, **kw)
    monkeypatch.setenv("MI_TILE_REORDER", "-1")
    L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, **kw)
    L_ref = eng.host_layout(base.n_cells, base.lower_addr, base.upper_addr, **kw0)
    tiles = lambda T: np.diff(T["tileCellStart"]).shape[0]
    slots = lambda T: int(np.diff(T["tileSlotStart"]).sum())
    if kind == "graph":   # a random graph has no locality to recover: the ordering must at least not hurt
        assert slots(L) <= 1.02 * slots(L_off)
    else:
        assert tiles(L_off) > 1.3 * tiles(L_ref)                       # what the numbering used to cost
        assert tiles(L) == tiles(L_ref) and slots(L) == slots(L_ref)   # the bricks of the well-numbered mesh
        assert int(np.diff(L["tileHaloStart"]).sum()) == int(np.diff(L_ref["tileHaloStart"]).sum())
    n = case.n_cells
    assert np.array_equal(np.sort(L["e2c"]), np.arange(n)) and np.array_equal(L["c2e"][L["e2c"]], np.arange(n))
    S = orc.System([case])
    x = syn.splitmix_uniform(21, n) - 0.5
    ext = bou = None
    if case.interfaces:
        ext = np.concatenate([x[case.interfaces[i.nbr_patch].face_cells] for i in case.interfaces])
    ref = S.amul(x)
    if case.interfaces:
        bou = np.concatenate([i.bou_coeffs for i in case.interfaces])
        got = interpret_amul(L, case, x, ext=np.zeros(len(bou)), bou=bou)     # cyclic patches are local couplings: no ext values
    else:
        got = interpret_amul(L, case, x)
    assert np.max(np.abs(got - ref)) <= 4e-16 * np.max(np.abs(ref)) * 8


def test_forced_reordering_on_disconnected_and_isolated_cells(pkg, orc, monkeypatch):
    """the Cuthill-McKee pre-ordering on the awkward inputs: several components, cells without any face, a single cell"""
    syn, eng = pkg.synthetic, pkg.engine
    monkeypatch.setenv("MI_TILE_REORDER", "1")
    # two boxes that do not touch + three isolated cells at the end
    a, b = syn.box_case(7, 5, 3), syn.box_case(4, 4, 4, symmetric=False)
    n = a.n_cells + b.n_cells + 3
    lo = np.concatenate([a.lower_addr, b.lower_addr + a.n_cells]).astype(np.int32)
    up = np.concatenate([a.upper_addr, b.upper_addr + a.n_cells]).astype(np.int32)
    upper = np.concatenate([a.upper, b.upper]); lower = np.concatenate([a.upper, b.lower])
    diag = np.concatenate([a.diag, b.diag, [2.0, 3.0, 4.0]])
    case = syn.LduCase(n, lo, up, diag, upper, lower, np.zeros(n))
    rng = np.random.default_rng(5)
    case = syn.renumber(case, rng.permutation(n).astype(np.int32))
    for tile_cells in (16, 1024):
        L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, tile_cells=tile_cells)
        assert np.array_equal(np.sort(L["e2c"]), np.arange(n))
        x = syn.splitmix_uniform(3, n) - 0.5
        ref = orc.System([case]).amul(x)
        assert np.max(np.abs(interpret_amul(L, case, x) - ref)) <= 4e-16 * np.max(np.abs(ref)) * 8
    L = eng.host_layout(1, np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert L["tileCellStart"].tolist() == [0, 1]


def test_renumber_at_bind_is_what_renumberMesh_does_with_the_engine_order(pkg):
    """mi_layout_adopt_host (the host part of mi_addr_create_adopted, round 3): the mesh renumbered into the engine's cell order in
    ONE call -- cells by the clustered layout's new-to-old map, faces re-pointed, flipped where owner > neighbour and sorted
    upper-triangular -- equals synthetic.renumber (polyMesh::renumber with that cell map) array for array; the maps invert
    consistently; and the renumbered mesh is tile-contiguous, so its ORDERED layout has the same tiles and the identity permutation."""
    syn, eng = pkg.synthetic, pkg.engine
    for case in (syn.box_case(20, 16, 12), syn.add_cyclic_y(syn.box_case(12, 10, 8)), random_graph_case(pkg, 1500, symmetric=False)):
        fcs = [i.face_cells for i in case.interfaces]
        nbs = [case.interfaces[i.nbr_patch].face_cells for i in case.interfaces]
        A = eng.adopt_host(case.n_cells, case.lower_addr, case.upper_addr, fcs, nbs)
        L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr, fcs, patch_nbr_cells=nbs)
        assert np.array_equal(A["cell_map"], L["e2c"]) and A["n_tiles"] == len(L["tileCellStart"]) - 1
        ref = syn.renumber(case, A["cell_map"])
        assert np.array_equal(A["lower"], ref.lower_addr) and np.array_equal(A["upper"], ref.upper_addr)
        assert np.array_equal(A["face_map"], ref.global_faces) and np.array_equal(A["face_flipped"].astype(bool), ref.face_flipped)
        assert np.all(A["lower"] < A["upper"]) and np.all(np.diff(A["lower"].astype(np.int64) * case.n_cells + A["upper"]) > 0)   # upper-triangular order
        # the ordered layout of the renumbered mesh: identity permutation, the same tile boundaries
        rfcs = [i.face_cells for i in ref.interfaces]
        rnbs = [ref.interfaces[i.nbr_patch].face_cells for i in ref.interfaces]
        Lo = eng.host_layout(ref.n_cells, ref.lower_addr, ref.upper_addr, rfcs, patch_nbr_cells=rnbs)
        assert np.array_equal(np.sort(Lo["e2c"]), np.arange(case.n_cells))


def test_box_addressing_equals_its_reference_form(pkg):
    """synthetic.box_addressing (prefix sum + strided fills, round 5) against the masked-neighbour-table form of rounds 1-4 that
    every committed record was generated with: the same arrays and dtypes, degenerate boxes included"""
    syn = pkg.synthetic
    for dims in [(1, 1, 1), (2, 1, 1), (1, 3, 1), (1, 1, 4), (5, 4, 3), (7, 1, 3), (1, 6, 5), (13, 11, 9), (40, 3, 2), (33, 32, 31)]:
        for a, b in zip(syn.box_addressing(*dims), syn.box_addressing_reference(*dims)):
            assert a.dtype == b.dtype and np.array_equal(a, b), dims
