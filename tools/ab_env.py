"""In-process interleaved A/B of engine build-time/runtime switches read from the environment at
context creation.  usage: AB_VAR=MI_PCG_FUSE_FINAL AB_VALUES=0,1 python tools/ab_env.py"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
case = syn.box_case(216, 216, 216)
N, F = case.n_cells, case.n_faces
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
diag, upper, src = t(case.diag), t(case.upper), t(case.source)
var = os.environ["AB_VAR"]
values = os.environ["AB_VALUES"].split(",")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
objs = {}
for v in values:
    os.environ[var] = v
    ctx = eng.Context(0, stream.cuda_stream)
    addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr); mat.set_coeffs(diag, upper, None)
    objs[v] = (ctx, addr, mat)
res = {v: dict(amul=[], pcg=[], amul_in_pcg=[]) for v in values}
psi0 = torch.zeros(N, dtype=torch.float64, device=dev)
for rnd in range(int(os.environ.get("AB_ROUNDS", "6"))):
    for v in values:
        ctx, addr, mat = objs[v]
        mat.bench_amul(5)
        res[v]["amul"].append(mat.bench_amul(50) / 50 * 1e3)
        mat.pcg_begin(psi0, src, "diagonal", tolerance=0.0, maxIter=400, history_len=0)
        mat.pcg_iterate(10); torch.cuda.synchronize(); t0 = time.perf_counter()
        mat.pcg_iterate(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        mat.pcg_end(None, 0)
        res[v]["pcg"].append(dt / 200 * 1e6)
for v in values:
    r = res[v]
    print(json.dumps({var: v, "amul_us_med": round(float(np.median(r["amul"])), 2), "pcg_us_med": round(float(np.median(r["pcg"])), 2),
                      "pcg_us_min": round(float(np.min(r["pcg"])), 2)}), flush=True)
