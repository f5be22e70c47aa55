"""The layout does not depend on the caller's numbering: the 216^3 box (or DIMS) with its cells renumbered at random
(the worst numbering a mesh generator could hand over) against the lexicographic one -- layout statistics, engine-order Amul,
caller-order mi_amul (whose permutation passes do see the numbering) and diagonal PCG.  Writes gpurun_out/shuffled.json."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
dims = [int(v) for v in os.environ.get("DIMS", "160,160,160").split(",")]
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
ctx = eng.Context(0, stream.cuda_stream)
base = syn.box_case(*dims)
rng = np.random.default_rng(7)
out = {}
for tag, case in (("lexicographic", base), ("shuffled", syn.renumber(base, rng.permutation(base.n_cells).astype(np.int32)))):
    N, F = case.n_cells, case.n_faces
    t0 = time.perf_counter(); addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr); tb = time.perf_counter() - t0
    A = eng.Matrix(addr); A.set_coeffs(t(case.diag), t(case.upper), None)
    x = t(syn.splitmix_uniform(1, N) - 0.5); y = torch.empty(N, dtype=torch.float64, device=dev)
    xe, ye = torch.empty(N + addr.n_ext, dtype=torch.float64, device=dev), torch.empty(N + addr.n_ext, dtype=torch.float64, device=dev)
    addr.to_engine(x, xe)
    def timeit(fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize(); A.event_record(0)
        for _ in range(reps): fn()
        A.event_record(1)
        return A.event_elapsed_ms(0, 1) / reps * 1e3
    r = dict(layout_build_s=round(tb, 2), stats=addr.stats(), amul_engine_us=round(timeit(lambda: A.amul_engine(xe, ye)), 1),
             amul_caller_us=round(timeit(lambda: A.amul(x, y)), 1))
    z = torch.zeros(N, dtype=torch.float64, device=dev)
    A.pcg_begin(z, t(case.source), "diagonal", tolerance=0.0, maxIter=400, history_len=0); A.pcg_iterate(20); torch.cuda.synchronize(); t0 = time.perf_counter()
    A.pcg_iterate(200); torch.cuda.synchronize(); r["pcg_us_per_iteration"] = round((time.perf_counter() - t0) / 200 * 1e6, 1)
    A.pcg_end(None, 0)
    r["amul_frac_of_8TBps"] = round((24 * N + 16 * F) / (r["amul_engine_us"] * 1e-6) / 8e12, 3)
    out[tag] = r
    print(tag, r, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(dict(dims=dims, **out), open(os.path.join(ROOT, "gpurun_out", "shuffled.json"), "w"), indent=1)
