"""Oracle results at the BASELINE sizes as committed fixtures (VERDICT r04 "next" 1b).

The serial CPU oracle needs minutes per 10 M / 40 M / 80 M-cell solve; inside the GPU suite that was ~40 % of the driver's
window.  tests/golden/make_full_size.py runs the oracle ON THE CPU BOX and writes, for every full-size GPU test,

  * residual histories, iteration counts, flags and normFactor (small arrays, stored whole),
  * sha256 of the result BITS of every operator that is compared bit for bit,
  * of every solution vector: its values at SAMPLE positions (sample_idx: a fixed multiplicative walk over the cells), its
    max |.|, its sum and its sum of magnitudes, and the sum / sum of magnitudes of each of N_CHUNKS contiguous chunks of it (so that
    an error confined to one tile or one processor-patch strip cannot hide in the whole-vector sums),
  * per section, sha256 of what the section was derived from: the oracle's C sources and its ctypes wrapper (by text), the
    section's own generator function (by text), the case generator's OUTPUT on small reference cases and -- for the assembly
    section, whose ordered variant runs on the mesh renumbered into the engine's tile order -- the tile layout's OUTPUT on
    reference cases; a fixture older than the code it restates fails a CPU test (tests/test_full_size_fixture.py) instead of
    passing silently; that file also RE-DERIVES a sample of the records with the live oracle.

The GPU tests (test_gpu_full_size.py, test_gpu_configs.py) compare the engine against these records with the bars they had
against the live oracle.  With MI_LIVE_ORACLE=1 they run the oracle in-process again (the pre-r05 behaviour).
"""
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FIXTURE = os.path.join(HERE, "golden", "full_size_v1.npz")
N_SAMPLE = 4096
N_CHUNKS = 4096
PERF_KEYS = ("nIterations", "converged", "singular")
PERF_REALS = ("normFactor", "initialResidual", "finalResidual")


def live_oracle():
    return os.environ.get("MI_LIVE_ORACLE", "0") == "1"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sample_idx(n, k=N_SAMPLE):
    """k positions spread over [0, n): a multiplicative walk (golden-ratio hashing), first and last cell included"""
    i = (np.arange(k, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(n)
    i = i.astype(np.int64)
    i[0], i[-1] = 0, n - 1
    return i


def _hash_case(h, case):
    for name in ("lower_addr", "upper_addr", "diag", "upper", "lower", "source", "global_cells"):
        a = getattr(case, name, None)
        h.update(name.encode())
        if a is not None:
            a = np.ascontiguousarray(a)
            h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    for itf in case.interfaces:
        h.update(f"{itf.nbr_domain},{itf.nbr_patch}".encode())
        for a in (itf.face_cells, itf.bou_coeffs, itf.int_coeffs):
            a = np.ascontiguousarray(a)
            h.update(str(a.dtype).encode()); h.update(a.tobytes())


def synthetic_fingerprint():
    """the case generator by its OUTPUT on small cases of every kind the records use (box, symmetric and asymmetric, the cyclic
    pairs, directly built sub-domains): synthetic.py may be rewritten for speed, its arrays may not change"""
    import __graft_entry__ as graft
    syn = graft.load_package().synthetic
    h = hashlib.sha256()
    for symmetric in (True, False):
        _hash_case(h, syn.box_case(12, 10, 8, symmetric=symmetric))
        for r in range(8):
            _hash_case(h, syn.box_subdomain((8, 6, 6), (2, 2, 2), r, symmetric=symmetric))
    _hash_case(h, syn.add_cyclic_x(syn.box_case(7, 5, 4)))
    _hash_case(h, syn.add_cyclic_y(syn.box_case(6, 5, 4)))
    h.update(syn.splitmix_uniform(99, 1000).tobytes())
    return h.hexdigest()


ASSEMBLY_SECTIONS = ("assembly216",)


def source_hashes(section=None, generator=None):
    """what a section's records were derived from: the oracle's C sources and ctypes wrapper (by text), the synthetic case
    generator (by output), the section's generator function in make_full_size.py (by text; `generator` = that function, looked up
    in make_full_size.SECTIONS when omitted) and, for the assembly section, the assembly case module (by text) and the tile layout
    (by output: the ordered variant runs on the mesh renumbered into the engine's tile order)"""
    import inspect
    h = {}
    rels = ["oracle/ldu_oracle.c", "oracle/ldu_oracle.h", "oracle/oracle.py"]
    rels += ["oracle/fvm_oracle.c", "tests/assembly_full_size.py"] if section in ASSEMBLY_SECTIONS else ["oracle/gamg_oracle.c"]
    for rel in rels:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h[rel] = hashlib.sha256(f.read()).hexdigest()
    h["synthetic.py outputs"] = synthetic_fingerprint()
    if section is not None:
        if generator is None:
            import sys
            sys.path.insert(0, os.path.join(HERE, "golden"))
            import make_full_size
            generator = make_full_size.SECTIONS[section]
        h["generator"] = hashlib.sha256(inspect.getsource(generator).encode()).hexdigest()
    if section in ASSEMBLY_SECTIONS:
        import sys
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import source_fingerprint
        h["tile layout outputs"] = source_fingerprint.layout_fingerprint()
    return h


# ---- writing (make_full_size.py) ----------------------------------------------------------------------------------------
def pack_perf(out, key, perf):
    out[key + "/history"] = np.asarray(perf["history"], dtype=np.float64)
    out[key + "/flags"] = np.array([int(perf[k]) for k in PERF_KEYS], dtype=np.int64)
    out[key + "/reals"] = np.array([float(perf.get(k, np.nan)) for k in PERF_REALS], dtype=np.float64)


def pack_solution(out, key, psi):
    psi = np.asarray(psi)
    out[key + "/psi_sample"] = psi[sample_idx(psi.shape[0])].copy()
    out[key + "/psi_stats"] = np.array([np.max(np.abs(psi)), psi.sum(), np.abs(psi).sum(), psi.shape[0]], dtype=np.float64)
    out[key + "/psi_chunks"] = chunk_sums(psi)


def chunk_sums(psi):
    """[2, N_CHUNKS]: sum and sum of magnitudes of each of N_CHUNKS contiguous chunks (np.array_split boundaries)"""
    b = np.linspace(0, psi.shape[0], N_CHUNKS + 1).astype(np.int64)[:-1]
    return np.stack([np.add.reduceat(psi, b), np.add.reduceat(np.abs(psi), b)])


def pack_sha(out, key, a):
    out[key + "/sha256"] = np.array(sha(a))


# ---- reading (the GPU tests) ---------------------------------------------------------------------------------------------
class Records:
    def __init__(self, path=FIXTURE):
        self.z = np.load(path)

    def has(self, key):
        return any(k.startswith(key + "/") for k in self.z.files)

    def perf(self, key):
        """the oracle's perf dict of that solve (history, nIterations, converged, singular, normFactor, ...)"""
        d = {"history": self.z[key + "/history"]}
        for k, v in zip(PERF_KEYS, self.z[key + "/flags"]):
            d[k] = int(v)
        for k, v in zip(PERF_REALS, self.z[key + "/reals"]):
            d[k] = float(v)
        return d

    def scalar(self, key):
        return self.z[key]

    def sha(self, key):
        return str(self.z[key + "/sha256"])

    def solution(self, key):
        return SolutionRecord(self.z[key + "/psi_sample"], self.z[key + "/psi_stats"], self.z[key + "/psi_chunks"] if key + "/psi_chunks" in self.z.files else None)


class SolutionRecord:
    """stands in for the oracle's solution vector: check(psi, tol) is the test's  max|psi - ref| < tol * max|ref|  on the sample
    positions plus the vector's sum and sum of magnitudes to the same tolerance (of the sum of magnitudes), and the same two sums of
    every one of N_CHUNKS contiguous chunks to tol * max|ref| * chunk length"""

    def __init__(self, sample, stats, chunks=None):
        self.sample = sample
        self.chunks = chunks
        self.maxabs, self.sum, self.abssum, self.n = float(stats[0]), float(stats[1]), float(stats[2]), int(stats[3])

    def deviation(self, psi):
        psi = np.asarray(psi)
        assert psi.shape[0] == self.n
        return float(np.max(np.abs(psi[sample_idx(self.n)] - self.sample)) / self.maxabs)

    def check(self, psi, tol):
        psi = np.asarray(psi)
        assert self.deviation(psi) < tol
        assert abs(float(psi.sum()) - self.sum) < tol * self.abssum
        assert abs(float(np.abs(psi).sum()) - self.abssum) < tol * self.abssum
        if self.chunks is not None:
            bar = tol * self.maxabs * np.diff(np.linspace(0, self.n, N_CHUNKS + 1).astype(np.int64))
            assert np.all(np.abs(chunk_sums(psi) - self.chunks) <= bar), "a chunk of the solution deviates"


def check_solution(psi, ref, tol):
    """ref: the oracle's vector (live oracle) or a SolutionRecord (fixture)"""
    if isinstance(ref, SolutionRecord):
        ref.check(psi, tol)
    else:
        assert np.max(np.abs(np.asarray(psi) - ref)) < tol * np.max(np.abs(ref))


def check_bits(got, ref):
    """ref: the oracle's vector (live oracle) or the sha256 of its bits (fixture)"""
    if isinstance(ref, str):
        assert sha(got) == ref
    else:
        assert np.array_equal(got, ref)
