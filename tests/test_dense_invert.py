"""GPU (-m gpu): the dense inversions behind GAMG's direct coarsest-level solve (mi_debug_dense_invert).

The register-resident kernels (k_dense_invert_reg, round 3; k_dense_invert_reg2, round 6: two barriers per pivot step, the pivot
column block a compile-time constant, swizzle-butterfly pivot search, one division per step) perform, element by element, the host
elimination's operations (gamg.cpp invert_dense: Gauss-Jordan with partial pivoting, first largest magnitude wins) -- so they must
return the host's BITS, also where rows are swapped in every step (GAMG's own coarsest matrices are diagonally dominant and never
reach that path), where magnitudes tie, where multipliers are exactly zero, at every register-tile size (n = 1 ... 192) and on a
singular matrix."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mats(n, rng):
    out = {"random": rng.standard_normal((n, n))}                        # a pivot search in every column, swaps almost everywhere
    a = rng.standard_normal((n, n)); a[np.abs(a) < 0.6] = 0.0
    out["sparse"] = a + np.diag(np.where(rng.random(n) < 0.5, 0.0, 3.0)) + 1e-3 * np.eye(n)[::-1]   # zero multipliers, zero diagonals
    t = rng.integers(-2, 3, size=(n, n)).astype(np.float64)              # ties in magnitude (+-1, +-2): the smallest row index wins
    out["ties"] = t + np.eye(n)[rng.permutation(n)] * 0.5
    d = -rng.random((n, n)) / n; np.fill_diagonal(d, 1.0 + rng.random(n))
    out["dominant"] = d                                                   # GAMG-like: never swaps
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 31, 32, 33, 63, 64, 65, 96, 100, 128, 151, 160, 161, 192])
def test_register_kernels_return_the_host_eliminations_bits(pkg, n):
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(1000 + n)
    for name, a in _mats(n, rng).items():
        ad = torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
        res = {}
        for which in (0, 1, 2, 3):
            inv = torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda:0")
            ok = ctx.dense_invert(ad, inv, n, which)
            torch.cuda.synchronize()
            res[which] = (ok, inv.cpu().numpy())
        ok0, ref = res[0]
        for which in (1, 2):
            assert res[which][0] == ok0, (n, name, which)
            if ok0:
                assert np.array_equal(res[which][1], ref), (n, name, which, np.max(np.abs(res[which][1] - ref)))
        if ok0:
            assert res[3][0] and np.allclose(res[3][1], ref, rtol=1e-9, atol=1e-9 * np.max(np.abs(ref)))
            resid = np.max(np.abs(a @ ref - np.eye(n)))
            assert resid < 1e-6 * max(1.0, np.linalg.cond(a)), (n, name, resid)


@pytest.mark.parametrize("n", [2, 40, 151])
def test_singular_matrices_are_reported(pkg, n):
    eng = pkg.engine
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    a = np.random.default_rng(7).standard_normal((n, n))
    a[:, n // 2] = 0.0                                                    # a zero column: no pivot
    ad = torch.from_numpy(a).to("cuda:0")
    for which in (0, 1, 2):
        inv = torch.zeros((n, n), dtype=torch.float64, device="cuda:0")
        assert ctx.dense_invert(ad, inv, n, which) is False, which
