"""A/B of tile-kernel load policies (MI_TILE_FLAGS) inside one process, interleaved rounds."""
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
variants = [int(v) for v in os.environ.get("FLAGS", "0,1,2,3,4,7").split(",")]
objs = {}
for fl in variants:
    os.environ["MI_TILE_FLAGS"] = str(fl)
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr); mat.set_coeffs(diag, upper, None)
    objs[fl] = (ctx, addr, mat)
res = {fl: dict(amul=[], pcg=[], amul_in_pcg=[]) for fl in variants}
psi0 = torch.zeros(N, dtype=torch.float64, device=dev)
for rnd in range(5):
    for fl in variants:
        ctx, addr, mat = objs[fl]
        mat.bench_amul(5)
        res[fl]["amul"].append(mat.bench_amul(50) / 50 * 1e3)
        mat.pcg_begin(psi0, src, "diagonal", tolerance=0.0, maxIter=400, history_len=0)
        mat.pcg_iterate(10); torch.cuda.synchronize(); t0 = time.perf_counter()
        ams = mat.pcg_iterate(100, time_amul=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        mat.pcg_end(None, 0)
        res[fl]["pcg"].append(dt * 1e4); res[fl]["amul_in_pcg"].append(ams * 10)
for fl in variants:
    r = res[fl]
    print(json.dumps(dict(flags=fl, amul_us_med=float(np.median(r["amul"])), amul_us_min=float(np.min(r["amul"])),
                          amul_in_pcg_us_med=float(np.median(r["amul_in_pcg"])), pcg_us_med=float(np.median(r["pcg"])), pcg_us_min=float(np.min(r["pcg"])))), flush=True)
