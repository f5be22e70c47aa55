"""GPU (-m gpu): csrc/pcg_fused.inc -- PCG's residual update of iteration `it`, its convergence test and the direction update of
iteration it + 1 (PCG.C:166-204, then :139-160) as ONE launch that keeps z = rD o rA on the chip between its two phases.

It replaces k_pcg_update_psi_r + k_pcg_final + k_pcg_update_p with the same grid, the same order of additions and the same roundings,
so everything must equal the separate kernels BIT FOR BIT: iteration counts, the whole residual history, psi, the convergence flags --
for every stopping rule of the reference's loop (tolerance, relTol, the maxIter quirk, minIter), both preconditioners, chunks that end
in registers, in LDS and beyond the chip, odd chunk lengths, mi_pcg_solve and the session API with uneven batches -- and when
workgroups LEAVE the launch's barrier (a device shared with another process: the launch never depends on co-residency): every third
workgroup, workgroup 0 (which collects the arrivals and closes the iteration), both; k_pcg_fused_finish then completes their chunks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def host(t):
    return t.detach().cpu().numpy()


def _ctx(pkg, monkeypatch, fuse, leave=0):
    monkeypatch.setenv("MI_PCG_GRAPH", "0")       # (the graph-replayed loop of small matrices keeps the separate kernels)
    monkeypatch.setenv("MI_PCG_PERSIST", "0")     # (so does the whole-solve persistent kernel)
    ctx = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("pcg_fuse_rp", fuse)
    ctx.set_option("pcg_fuse_test", leave)        # 1: every third workgroup leaves the barrier at once, 2: workgroup 0 does, 3: both
    return ctx


def _make(pkg, ctx, case):
    eng = pkg.engine
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    mat.set_coeffs(dev(case.diag), dev(case.upper), None)
    return addr, mat


CONTROLS = (dict(tolerance=1e-9, maxIter=400), dict(tolerance=0.0, maxIter=7), dict(tolerance=1e-30, relTol=1e-3, maxIter=400),
            dict(tolerance=1e30, minIter=3, maxIter=400), dict(tolerance=0.0, maxIter=0), dict(tolerance=0.0, maxIter=1))


@pytest.mark.parametrize("dims", [(1, 1, 1), (3, 1, 1), (11, 9, 7), (40, 32, 24), (65, 63, 61)])
@pytest.mark.parametrize("precond", ["diagonal", "none"])
def test_fused_launch_equals_the_separate_kernels_bit_for_bit(pkg, dims, precond, monkeypatch):
    case = pkg.synthetic.box_case(*dims)
    n = case.n_cells
    out = {}
    for fuse, leave in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3)):
        ctx = _ctx(pkg, monkeypatch, fuse, leave)
        addr, mat = _make(pkg, ctx, case)
        res = []
        for kw in CONTROLS:
            psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
            perf = mat.pcg(psi, dev(case.source), precond, **kw)
            res.append((perf, host(psi)))
        # a second solve from the last solution (residual already small: the first test ends it)
        psi2 = psi.clone()
        res.append((mat.pcg(psi2, dev(case.source), precond, tolerance=1e-9, maxIter=400), host(psi2)))
        # the session API as bench.py drives it, uneven batches
        hl = 40
        mat.pcg_begin(dev(np.zeros(n)), dev(case.source), precond, tolerance=0.0, maxIter=30, history_len=hl)
        for k in (1, 2, 5, 3, 16):
            mat.pcg_iterate(k)
        pe = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        res.append((mat.pcg_end(pe, hl), host(pe)))
        out[fuse, leave] = res
        assert (ctx.stat(4) > 0) == (fuse == 1), (fuse, ctx.stat(4))
    for key in ((1, 0), (1, 1), (1, 2), (1, 3)):
        for (pa, xa), (pb, xb) in zip(out[0, 0], out[key]):
            for k in ("nIterations", "converged", "singular", "initialResidual", "finalResidual", "normFactor"):
                assert pa[k] == pb[k] or (np.isnan(pa[k]) and np.isnan(pb[k])), (key, k, pa[k], pb[k])
            assert np.array_equal(pa["history"], pb["history"], equal_nan=True), key
            assert np.array_equal(xa, xb), key


@pytest.mark.parametrize("dims, where", [((176, 176, 176), "lds"), ((224, 224, 224), "beyond the chip")])
def test_fused_launch_where_a_chunk_reaches_lds_and_past_it(pkg, orc, dims, where, monkeypatch):
    """5.45 M cells: a chunk's z values fill the registers and part of LDS; 11.2 M cells: the tail of every chunk is formed again
    from rD and rA.  Separate kernels and fused launch bit for bit; the first iterations against the oracle (PCG.C:133-204) to 1e-10."""
    case = pkg.synthetic.box_case(*dims)
    n = case.n_cells
    out = {}
    for fuse, leave in ((0, 0), (1, 0), (1, 3)):
        ctx = _ctx(pkg, monkeypatch, fuse, leave)
        addr, mat = _make(pkg, ctx, case)
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        perf = mat.pcg(psi, dev(case.source), "diagonal", tolerance=0.0, maxIter=24)
        out[fuse, leave] = (perf, host(psi))
        assert (ctx.stat(4) > 0) == (fuse == 1)
        del mat, addr, ctx
    (pa, xa), (pb, xb), (pc, xc) = out[0, 0], out[1, 0], out[1, 3]
    assert pa["nIterations"] == pb["nIterations"] == 25 and np.array_equal(pa["history"], pb["history"]) and np.array_equal(xa, xb)
    assert pc["nIterations"] == 25 and np.array_equal(pa["history"], pc["history"]) and np.array_equal(xa, xc)
    _, ref = orc.System([case]).pcg(np.zeros(n), case.source, "diagonal", tolerance=0.0, maxIter=5)
    assert np.max(np.abs(pb["history"][:ref["history"].shape[0]] - ref["history"])) < 1e-10 * ref["history"][0]


def test_fused_launch_on_a_singular_direction(pkg, monkeypatch):
    """wApA = 0 (zero matrix coefficients against a non-zero source): checkSingularity ends the loop before psi or rA move
    (SolverPerformance.C:32-44, PCG.C:168); same flags from both forms"""
    case = pkg.synthetic.box_case(11, 9, 7)
    n = case.n_cells
    out = {}
    for fuse in (0, 1):
        ctx = _ctx(pkg, monkeypatch, fuse)
        eng = pkg.engine
        addr = eng.Addressing(ctx, n, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr)
        mat.set_coeffs(dev(np.zeros(n)), dev(np.zeros(case.n_faces)), None)
        psi = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        perf = mat.pcg(psi, dev(case.source), "none", tolerance=0.0, maxIter=5)
        out[fuse] = (perf, host(psi))
    (pa, xa), (pb, xb) = out[0], out[1]
    assert pa["singular"] == pb["singular"] and pa["nIterations"] == pb["nIterations"] and np.array_equal(xa, xb, equal_nan=True)
    assert np.array_equal(pa["history"], pb["history"], equal_nan=True)
