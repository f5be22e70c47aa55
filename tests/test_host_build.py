"""CPU: the threaded one-time host builds (tile layout, GAMG hierarchy) give the tables of their sequential forms.

Round 4 moved the sequential passes of `build_tile_layout` / `build_gamg_hierarchy` onto a shared pool of host threads
(csrc/host_parallel.hpp: counting sorts through atomic cursors + per-bucket sort, prefix scans, per-owner passes), narrowed the
sequential pair matching to the neighbours that can still be free (csrc/gamg.cpp: `laterOnly`, the forward sweep over owned face
ranges) and added an opt-in matching by the host threads (csrc/host_match.hpp).  None of it may move a slot or a coarse cell:
the reference's visiting order decides the agglomeration (pairGAMGAgglomerate.C:204-313, pinned by
tests/test_gamg.py::test_pair_agglomeration_equals_the_reference_code), and the committed PMC traffic figures are tied to the
layout's output (tools/source_fingerprint.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Env:
    def __init__(self, **kv): self.kv = kv
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items(): os.environ[k] = str(v)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _same_levels(a, b, what):
    assert len(a) == len(b), (what, len(a), len(b))
    for l, (x, y) in enumerate(zip(a, b)):
        for k in x:
            if isinstance(x[k], np.ndarray):
                assert np.array_equal(x[k], y[k]), (what, l, k)


def _weight_sets(pkg, case):
    syn = pkg.synthetic
    nf = case.n_faces
    u = syn.splitmix_uniform(77, nf)
    sets = {
        "uniform": np.ones(nf),                                   # every choice is a tie: list order decides
        "random": 0.5 + u,
        "few_values": np.floor(u * 3.0),                          # ties and zeros
        "negative": u - 0.5,                                       # still choosable (> -1e20)
    }
    odd = (0.5 + u).copy(); odd[::97] = -1e30                      # faces that can never be chosen: the plain sequential loop
    sets["unchoosable"] = odd
    nan = (0.5 + u).copy(); nan[5::211] = np.nan
    sets["nan"] = nan
    return sets


@pytest.mark.parametrize("shape", ["box", "graph", "shuffled"])
def test_pair_matching_variants_give_the_same_hierarchy(pkg, shape):
    """later-neighbours-only sequential matching (default), the plain loop (MI_MATCH_LATER_ONLY=0) and the matching by the host
    threads (MI_MATCH_PARALLEL=1) on a level large enough to take them (>= 32768 cells), both sweep directions, with ties,
    zero / negative weights, and weights the narrowed forms must refuse (<= -1e20, NaN: they fall back to the plain loop)"""
    eng, syn = pkg.engine, pkg.synthetic
    if shape == "box":
        case = syn.box_case(40, 36, 44)
    else:
        from conftest import random_graph_case
        case = random_graph_case(pkg, 40000, extra=2.5, seed=9)
        if shape == "shuffled":   # faces in no order at all (owners not ascending: face lists instead of face ranges)
            order = syn.splitmix_uniform(31, case.n_faces).argsort()
            case = syn.LduCase(case.n_cells, case.lower_addr[order], case.upper_addr[order], case.diag, case.upper[order], None, case.source)
    for name, w in _weight_sets(pkg, case).items():
        for fwd in (True, False):
            with _Env(MI_MATCH_LATER_ONLY=0, MI_MATCH_PARALLEL=0):
                plain = eng.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, w, 30, fwd)
            with _Env(MI_MATCH_LATER_ONLY=1, MI_MATCH_PARALLEL=0):
                later = eng.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, w, 30, fwd)
            with _Env(MI_MATCH_LATER_ONLY=1, MI_MATCH_PARALLEL=1):
                par = eng.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, w, 30, fwd)
            assert len(plain) >= 3
            _same_levels(plain, later, (shape, name, fwd, "later-only"))
            _same_levels(plain, par, (shape, name, fwd, "host threads"))


def test_layout_clustering_by_the_host_threads_gives_the_same_layout(pkg):
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.box_case(40, 36, 44)
    perm = syn.splitmix_uniform(9, case.n_cells).argsort().astype(np.int32)      # a numbering without locality as well
    inv = np.empty_like(perm); inv[perm] = np.arange(case.n_cells, dtype=np.int32)
    lo, up = inv[case.lower_addr], inv[case.upper_addr]
    lo, up = np.minimum(lo, up), np.maximum(lo, up)
    order = np.lexsort((up, lo))
    for lower, upper in ((case.lower_addr, case.upper_addr), (lo[order], up[order])):
        with _Env(MI_MATCH_PARALLEL=0):
            a = eng.host_layout(case.n_cells, lower, upper)
        with _Env(MI_MATCH_PARALLEL=1):
            b = eng.host_layout(case.n_cells, lower, upper)
        for k in a:
            assert np.array_equal(a[k], b[k]), k


_BIG = """
import sys, hashlib, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import __graft_entry__ as g, workloads
pkg = g.load_package(); eng, syn = pkg.engine, pkg.synthetic
case = syn.box_case(60, 56, 48)          # 161 280 cells: the two-pass prefix scans and every threaded pass take their parallel form
h = hashlib.sha256()
L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr)
for k in sorted(L): h.update(k.encode()); h.update(np.ascontiguousarray(L[k]).tobytes())
for fwd in (True, False):
    for lvl in eng.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, 0.5 + syn.splitmix_uniform(5, case.n_faces), 50, fwd):
        for k in sorted(lvl):
            if isinstance(lvl[k], np.ndarray): h.update(k.encode()); h.update(np.ascontiguousarray(lvl[k]).tobytes())
print(h.hexdigest())
"""


def test_larger_case_one_host_thread_against_all(pkg):
    outs = []
    for threads in ("1", "0"):
        env = dict(os.environ)
        env.pop("MI_HOST_THREADS", None)
        if threads != "0": env["MI_HOST_THREADS"] = threads
        out = subprocess.run([sys.executable, "-c", _BIG], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(out.stdout.split()[-1])
    assert outs[0] == outs[1]


def test_tables_do_not_depend_on_the_number_of_host_threads(pkg):
    """the fingerprints of tools/source_fingerprint.py (every table of the layout / hierarchy of its reference cases) computed
    with ONE host thread in a fresh process equal those of this process (all cores)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_fingerprint as sf
    here = (sf.layout_fingerprint(), sf.hierarchy_fingerprint())
    env = dict(os.environ, MI_HOST_THREADS="1")
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, 'tools'); import source_fingerprint as s; print(s.layout_fingerprint(), s.hierarchy_fingerprint())"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert tuple(out.stdout.split()[-2:]) == here


def _tile_of_cell(L, n_cells):
    t = np.empty(n_cells, dtype=np.int32)
    starts = L["tileCellStart"]
    for k in range(len(starts) - 1):
        t[L["e2c"][starts[k]:starts[k + 1]]] = k
    return t


def test_the_partition_of_a_clustered_layout_gives_that_layout_back(pkg):
    """TileParams::givenPart (mi_layout_build_host_given): no clustering, tiles ordered by their smallest cell, cells ascending
    -- fed with the tiles the clustering made, every table of the layout comes out the same"""
    eng, syn = pkg.engine, pkg.synthetic
    case = syn.box_case(40, 36, 44)
    L = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr)
    part = _tile_of_cell(L, case.n_cells)
    G = eng.host_layout_given(case.n_cells, case.lower_addr, case.upper_addr, part, len(L["tileCellStart"]) - 1)
    for k in G:
        assert np.array_equal(G[k], L[k]), k
    with pytest.raises(eng.MiError):      # a tile beyond the caps is refused (the caller then clusters)
        eng.host_layout_given(case.n_cells, case.lower_addr, case.upper_addr, np.zeros(case.n_cells, dtype=np.int32), 1)


def test_tiles_inherited_from_the_finer_gamg_level(pkg):
    """inherit_tiles (csrc/tiling.hpp): a coarse cell goes where its first child is, tiles merged pairwise under the caps.  Every
    inherited partition must be a valid layout input, and on the levels of a box it must be as good as clustering that level
    from scratch (tools/exp/inherit_tiles.cpp: identical tiles on the first two levels of the 216^3 box)."""
    eng, syn = pkg.engine, pkg.synthetic
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    case = syn.box_case(64, 48, 48)
    fineL = eng.host_layout(case.n_cells, case.lower_addr, case.upper_addr)
    tile_of, n_tiles = _tile_of_cell(fineL, case.n_cells), len(fineL["tileCellStart"]) - 1
    H = eng.gamg_host_hierarchy(case.n_cells, case.lower_addr, case.upper_addr, workloads.box_pair_weights(case), 50, True)
    checked = 0
    for lvl in H[:3]:
        n_coarse = int(lvl["restrictMap"].max()) + 1
        if n_coarse < 4096:
            break
        part, n_parts = eng.inherit_tiles(lvl["restrictMap"], tile_of, n_tiles, n_coarse, lvl["cLower"], lvl["cUpper"])
        assert part.min() == 0 and part.max() == n_parts - 1 and np.bincount(part).max() <= 1024
        inherited = eng.host_layout_given(n_coarse, lvl["cLower"], lvl["cUpper"], part, n_parts)
        clustered = eng.host_layout(n_coarse, lvl["cLower"], lvl["cUpper"])
        assert np.array_equal(np.sort(inherited["e2c"]), np.arange(n_coarse))
        assert len(inherited["haloCell"]) <= 1.05 * len(clustered["haloCell"]), (len(inherited["haloCell"]), len(clustered["haloCell"]))
        assert len(inherited["slotFace"]) <= 1.02 * len(clustered["slotFace"])
        tile_of, n_tiles = part, n_parts          # the next level inherits from this one
        checked += 1
    assert checked >= 2
