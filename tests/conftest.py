import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Value per second first (VERDICT r04 "next" 1a): the broad, cheap parity files run before the multi-process ones, the 10 M / 40 M /
# 80 M-cell cases last -- a suite that is cut short loses its most expensive evidence, not its broadest.
_FILE_ORDER = ("test_abi", "test_gpu_parity", "test_gpu_fuzz", "test_ref_dropin", "test_golden", "test_assembly", "test_gamg", "test_ami",
               "test_polymesh", "test_c_abi_demo", "test_foam_mirror", "test_layout", "test_oracle", "test_host_build", "test_traffic_record",
               "test_full_size_fixture", "test_distributed", "test_bench_contract", "test_gpu_full_size", "test_gpu_configs")


def _world_of(item):
    """ranks a multi-process test runs on (0: none): tests/rank_pool.py keeps ONE pool of rank processes alive, so the tests of
    test_distributed.py run grouped by world size"""
    p = getattr(getattr(item, "callspec", None), "params", {})
    if "world" in p:
        return int(p["world"])
    if "parts" in p and isinstance(p["parts"], tuple):
        return int(np.prod(p["parts"]))
    name = item.name.split("[")[0]
    return {"test_transformed_processor_patches_between_engine_ranks": 4, "test_distributed_pcg_real_engine_ragged_graph_random_partition": 3}.get(name, 0)


def _uses_windows(item):
    """ranks of this test wait for each other inside kernels (hipIpc windows, persistent grids): a rank PROCESS that has done
    that once stays slow for plain-transport jobs too (measured: 60 x, profiles/r05_c_native_solve_timings.tsv), so within one world
    size the plain-transport tests run first"""
    p = getattr(getattr(item, "callspec", None), "params", {})
    return int(bool(p.get("peer", False)) or any(w in item.name for w in ("peer", "window", "persistent")))


def pytest_collection_modifyitems(config, items):
    def key(item):
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        rank = _FILE_ORDER.index(name) if name in _FILE_ORDER else _FILE_ORDER.index("test_distributed") - 0.5
        return (rank,) + ((_world_of(item), _uses_windows(item)) if name == "test_distributed" else (0, 0))
    items.sort(key=key)          # stable: apart from the grouping by world size the order inside a file stays the file's


def pytest_runtest_logreport(report):
    """every phase's duration as it completes (gpurun_out/test_durations.tsv): a suite that is cut short still says where its time
    went -- pytest's own --durations table is only printed at the end"""
    if os.environ.get("MI_TEST_DURATIONS", "1") == "0" or report.duration < 0.05:
        return
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "test_durations.tsv"), "a") as f:
            f.write(f"{report.duration:9.2f}\t{report.when}\t{report.outcome}\t{report.nodeid}\n")
    except OSError:
        pass


@pytest.fixture(scope="session", autouse=True)
def _rank_pools():
    """the pooled rank processes of tests/rank_pool.py live for the session"""
    yield
    import rank_pool
    rank_pool.close_all()


@pytest.fixture(scope="session")
def pkg():
    graft.build()  # compiles the HIP engine (cross-compiles without a GPU) and the oracle
    return graft.load_package()


@pytest.fixture(scope="session")
def orc(pkg):
    from oracle import oracle
    return oracle


def random_graph_case(pkg, n=700, extra=2.5, seed=3, symmetric=True):
    """Ragged non-box LDU addressing: a random connected graph with varying row lengths."""
    syn = pkg.synthetic
    nf_target = int(n * extra)
    u = syn.splitmix_uniform(seed, 2 * nf_target + 2 * n)
    a = (u[:nf_target] * n).astype(np.int64)
    b = (u[nf_target:2 * nf_target] * n).astype(np.int64)
    chain = np.arange(n - 1)
    lo = np.concatenate([np.minimum(a, b), chain])
    up = np.concatenate([np.maximum(a, b), chain + 1])
    keep = lo != up
    pairs = np.unique(np.stack([lo[keep], up[keep]], axis=1), axis=0)  # sorted by (lower, upper): upper-triangular order
    lo, up = pairs[:, 0].astype(np.int32), pairs[:, 1].astype(np.int32)
    nf = lo.shape[0]
    c = syn.splitmix_uniform(seed + 1, 2 * nf)
    upper = -(0.2 + c[:nf])
    lower = None if symmetric else -(0.2 + c[nf:])
    diag = np.zeros(n)
    np.subtract.at(diag, lo, upper if symmetric else lower)
    np.subtract.at(diag, up, upper)
    diag += 0.05 + syn.splitmix_uniform(seed + 2, n)
    source = syn.splitmix_uniform(seed + 3, n) - 0.5
    return syn.LduCase(n, lo, up, diag, upper, lower, source)
