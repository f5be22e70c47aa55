"""Time the coarsest-level dense inversions (mi_debug_dense_invert) on a GAMG-like matrix:  N=151 python tools/bench_invert.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
ctx = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
for n in [int(v) for v in os.environ.get("N", "60,100,151,192").split(",")]:
    rng = np.random.default_rng(3)
    a = -rng.random((n, n)) / n; np.fill_diagonal(a, 1.0 + rng.random(n))
    ad = torch.from_numpy(a).to("cuda:0"); inv = torch.zeros((n, n), dtype=torch.float64, device="cuda:0")
    line = [f"n {n}"]
    for which, name in ((1, "reg v1"), (2, "reg v2"), (3, "global")):
        ctx.dense_invert(ad, inv, n, which); torch.cuda.synchronize()
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); ctx.dense_invert(ad, inv, n, which); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
        line.append(f"{name} {min(t) * 1e6:.0f} us (call incl. alloc + sync)")
    print(" | ".join(line))
