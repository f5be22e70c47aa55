"""GPU tuning sweep (not part of the product): Amul / PCG timing over tile size and block size."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
dims = [int(v) for v in os.environ.get("SWEEP_DIMS", "216,216,216").split(",")]
case = syn.box_case(*dims)
N, F = case.n_cells, case.n_faces
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
res = []
# streaming calibration: torch copy of 3 x 80 MB-class vectors
a = torch.empty(64 * 1024 * 1024, dtype=torch.float64, device=dev); b = torch.empty_like(a)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): b.copy_(a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"copy 512MiB->512MiB: {2*a.numel()*8/dt/1e9:.0f} GB/s", flush=True)
del a, b
diag, upper, src = t(case.diag), t(case.upper), t(case.source)
tiles = [int(v) for v in os.environ.get("SWEEP_TILES", "512,1024,2048").split(",")]
bss = [int(v) for v in os.environ.get("SWEEP_BS", "256,512,1024").split(",")]
for tile in tiles:
    os.environ["MI_TILE_CELLS"] = str(tile)
    os.environ["MI_TILE_SLOTS"] = str(min(32000, tile * 4 + 64))
    for bs in bss:
        os.environ["MI_AMUL_BS"] = str(bs)
        ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
        addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr)
        mat.set_coeffs(diag, upper, None)
        st = addr.stats()
        occ = mat.occupancy()
        mat.bench_amul(5)
        ms = min(mat.bench_amul(50) for _ in range(3)) / 50
        psi0 = torch.zeros(N, dtype=torch.float64, device=dev)
        mat.pcg_begin(psi0, src, "diagonal", tolerance=0.0, maxIter=400, history_len=0)
        mat.pcg_iterate(10); torch.cuda.synchronize(); t0 = time.perf_counter()
        mat.pcg_iterate(100); torch.cuda.synchronize(); pcg_ms = (time.perf_counter() - t0) * 10
        mat.pcg_end(None, 0)
        gbs = (24 * N + 16 * F) / (ms * 1e-3) / 1e9
        r = dict(tile=tile, bs=bs, amul_us=ms * 1e3, amul_GBs=gbs, frac=gbs / 8000, pcg_ms=pcg_ms, its=1e3 / pcg_ms,
                 pcg_GBs=(160 * N + 16 * F) / (pcg_ms * 1e-3) / 1e9, lds=st["lds_bytes_sym"], slots_per_row=st["slots"] / N,
                 halo_per_row=st["halo"] / N, occ=occ["blocks_per_cu"], entries_per_row=st["entries"] / N)
        print(json.dumps(r), flush=True)
        res.append(r)
        mat.close(); addr.close(); ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w"), indent=1)
