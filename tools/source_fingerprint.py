"""What the committed PMC traffic figures (profiles/traffic_latest.json) are tied to.

A figure measured with rocprofv3 --pmc on the GPU box stays valid while (a) the KERNELS that move the bytes are the ones it was
measured with -- their source text is hashed -- and (b) the host builders hand those kernels the same tables: the tile layout
and the GAMG hierarchy are hashed by their OUTPUT on fixed reference cases (every array mi_layout_build_host /
mi_gamg_host_build return), not by the text of tiling.cpp / gamg.cpp.  A host-side change that leaves every table bit-identical
(threads, allocation, timing hooks) therefore keeps the figures; one that moves a single slot voids them.  Needs the built
library (python -c "import __graft_entry__ as g; g.build()"); no GPU."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rapidcfd-dev_amd", "csrc")
_cache = {}


def _pkg():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as graft
    return graft.load_package()


def _product_text(path):
    """the file without its `#ifdef MI_TIMING ... [#else ...] #endif` hooks (tools/timing_build.sh builds them in; the product
    never does; the #else branch, if any, is kept) and without the regions marked `// [start-up begin]` ... `// [start-up end]`
    (hierarchy creation in gamg_engine.inc: once-per-mesh host code whose RESULT the output fingerprints cover)"""
    out, depth, keep, startup = [], 0, True, False
    for line in open(path, "r").read().split("\n"):
        t = line.strip()
        if t.startswith("// [start-up begin]"): startup = True
        if startup:
            if t.startswith("// [start-up end]"): startup = False
            continue
        if depth == 0 and t.startswith("#ifdef MI_TIMING"): depth, keep = 1, False; continue
        if depth > 0:
            if t.startswith("#if"): depth += 1
            elif t.startswith("#endif"):
                depth -= 1
                if depth == 0: keep = True; continue
            elif depth == 1 and t.startswith("#else"): keep = True; continue
        if keep: out.append(line)
    return "\n".join(out).encode()


def text_hash(files):
    h = hashlib.sha256()
    for f in files:
        h.update(_product_text(os.path.join(CSRC, f)))
    return h.hexdigest()


def _hash_arrays(h, d):
    for k in sorted(d):
        a = np.ascontiguousarray(d[k])
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())


def _cases(pkg):
    syn = pkg.synthetic
    plain = syn.box_case(40, 36, 44)                       # 63 360 cells: 62+ tiles, interior and cut faces in all directions
    cyc = syn.add_cyclic_y(syn.box_case(24, 20, 28))       # a local coupled patch pair: ext slots, private halo entries
    return plain, cyc


def layout_fingerprint():
    """sha256 over every table of the tile layout of the reference cases (default tile caps)"""
    if "layout" not in _cache:
        pkg = _pkg(); eng = pkg.engine
        h = hashlib.sha256()
        plain, cyc = _cases(pkg)
        _hash_arrays(h, eng.host_layout(plain.n_cells, plain.lower_addr, plain.upper_addr))
        a, b = cyc.interfaces
        _hash_arrays(h, eng.host_layout(cyc.n_cells, cyc.lower_addr, cyc.upper_addr, [a.face_cells, b.face_cells], patch_nbr_cells=[b.face_cells, a.face_cells]))
        perm = pkg.synthetic.splitmix_uniform(9, plain.n_cells).argsort().astype(np.int32)   # a numbering without locality: the Cuthill-McKee path
        inv = np.empty_like(perm); inv[perm] = np.arange(plain.n_cells, dtype=np.int32)
        lo, up = inv[plain.lower_addr], inv[plain.upper_addr]
        lo, up = np.minimum(lo, up), np.maximum(lo, up)
        order = np.lexsort((up, lo))
        _hash_arrays(h, eng.host_layout(plain.n_cells, lo[order], up[order]))
        _cache["layout"] = h.hexdigest()
    return _cache["layout"]


def hierarchy_fingerprint():
    """sha256 over every level's maps of the GAMG hierarchy of the reference cases (pair agglomeration, both sweep directions)"""
    if "hier" not in _cache:
        pkg = _pkg(); eng = pkg.engine
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import workloads
        h = hashlib.sha256()
        plain, _ = _cases(pkg)
        w = workloads.box_pair_weights(plain)
        for fwd in (True, False):
            for lvl in eng.gamg_host_hierarchy(plain.n_cells, plain.lower_addr, plain.upper_addr, w, 50, fwd):
                _hash_arrays(h, {k: v for k, v in lvl.items() if isinstance(v, np.ndarray)})
        w2 = 0.5 + pkg.synthetic.splitmix_uniform(21, plain.n_faces)
        for lvl in eng.gamg_host_hierarchy(plain.n_cells, plain.lower_addr, plain.upper_addr, w2, 50, True):
            _hash_arrays(h, {k: v for k, v in lvl.items() if isinstance(v, np.ndarray)})
        _cache["hier"] = h.hexdigest()
    return _cache["hier"]


def layout_source_hash():
    """the Amul figure: the tile kernels by text, the layout by its output (tiling.hpp / tiling.cpp hold the host structures and
    the builder: every table they hand to the kernels is in the fingerprint)"""
    return hashlib.sha256((text_hash(("kernels.hip.hpp",)) + layout_fingerprint()).encode()).hexdigest()[:16]


def gamg_source_hash():
    """the V-cycle figure: additionally the GAMG kernels / cycle code by text (gamg_engine.inc outside its start-up region) and
    the hierarchy by its output (gamg.hpp / gamg.cpp hold host structures and builders only)"""
    return hashlib.sha256((text_hash(("kernels.hip.hpp", "gamg_engine.inc")) + layout_fingerprint() + hierarchy_fingerprint()).encode()).hexdigest()[:16]


if __name__ == "__main__":
    print("layout", layout_source_hash(), "gamg", gamg_source_hash())
