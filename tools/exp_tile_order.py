"""Round 6 experiment: does the ORDER of the tiles matter?  Neighbour-side faces / halo cells of a tile belong to the tiles one row (14
tiles) and one plane (378 tiles) back in the lexicographic tile order of the 216^3 box: by the time a tile runs, the plane-back neighbour's
sectors have left the XCD's L2 and the Infinity Cache, so every cut face / halo value is fetched from HBM a second time (PMC: traffic /
algorithmic 1.10 for Amul, 1.3 - 1.7 for the assembly row passes).  Here the SAME 16 x 8 x 8 bricks are numbered along (a) the
lexicographic order (what the layout produces today), (b) a Morton curve over the brick coordinates (y, z) with x innermost, (c) a 3-D
Morton curve -- through mi_addr_create_ordered with given tile starts, no engine change -- and Amul and the row passes are timed.
   python tools/exp_tile_order.py            -> gpurun_out/r06_exp_tile_order.json"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
dims = [int(v) for v in os.environ.get("DIMS", "216,216,216").split(",")]
nx, ny, nz = dims
BX, BY, BZ = 16, 8, 8
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
E = lambda n: torch.empty(n, dtype=torch.float64, device=dev)
ctx = eng.Context(0, stream.cuda_stream)
base = syn.box_case(*dims)
N, F = base.n_cells, base.n_faces
c = np.arange(N, dtype=np.int64)
ci, cj, ck = c % nx, (c // nx) % ny, c // (nx * ny)
tx, ty, tz = ci // BX, cj // BY, ck // BZ
ntx, nty, ntz = (nx + BX - 1) // BX, (ny + BY - 1) // BY, (nz + BZ - 1) // BZ


def part1by1(v):
    v = v.astype(np.uint64) & np.uint64(0xFFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x33333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x55555555)
    return v


def part1by2(v):
    v = v.astype(np.uint64) & np.uint64(0x3FF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x30000FF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x300F00F)
    v = (v | (v << np.uint64(4))) & np.uint64(0x30C30C3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x9249249)
    return v


def tile_key(order):
    if order == "lexicographic":
        return (tx + ntx * (ty + nty * tz)).astype(np.uint64)
    if order == "morton_yz_x_inner":
        return ((part1by1(ty) | (part1by1(tz) << np.uint64(1))) * np.uint64(ntx) + tx.astype(np.uint64))
    if order == "morton_xyz":
        return part1by2(tx) | (part1by2(ty) << np.uint64(1)) | (part1by2(tz) << np.uint64(2))
    if order == "pencils_4x4":     # (y, z) in blocks of 4 x 4 tiles, x innermost: neighbours at most a few rows away
        by, bz = ty // 4, tz // 4
        return ((((bz * ((nty + 3) // 4) + by) * 4 + (tz % 4)) * 4 + (ty % 4)) * ntx + tx).astype(np.uint64)
    raise ValueError(order)


out = {}
for order in os.environ.get("ORDERS", "lexicographic,morton_yz_x_inner,morton_xyz,pencils_4x4").split(","):
    key = tile_key(order)
    inner = (ci % BX) + BX * ((cj % BY) + BY * (ck % BZ))
    perm = np.lexsort((inner, key))                     # new cell i = old cell perm[i]: tiles in key order, cells x-fastest inside
    _, counts = np.unique(key[perm], return_counts=True)
    # np.unique sorts by key value == the order of appearance in perm
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    case = syn.renumber(base, perm)
    addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr, ordered=True, tile_cell_start=starts)
    assert addr.is_ordered
    A = eng.Matrix(addr); A.set_coeffs(t(case.diag), t(case.upper), None)
    rows = {"tiles": int(addr.n_tiles)}
    rows["Amul_us"] = min(A.bench_amul(100) for _ in range(3)) * 1e3
    asm = eng.Assembly(addr)
    ff, fw = t(syn.splitmix_uniform(2, F)), t(syn.splitmix_uniform(3, F))
    fl, fu, fd, y = E(F), E(F), E(N), E(N)
    Sf = [t(syn.splitmix_uniform(10 + k, F)) for k in range(3)]
    vol = t(np.full(N, 1.0)); g3 = [E(N) for _ in range(3)]

    def timeit(name, fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            A.event_record(0)
            for _ in range(reps): fn()
            A.event_record(1)
            best = min(best, A.event_elapsed_ms(0, 1) / reps * 1e3)
        rows[name + "_us"] = round(best, 1)
    timeit("fvm::laplacian", lambda: asm.fvm_laplacian(ff, fw, fu, fd))
    timeit("fvm::div", lambda: asm.fvm_div(fw, ff, fl, fu, fd))
    timeit("surfaceIntegrate", lambda: asm.surface_integrate(ff, None, y))
    timeit("gaussGrad", lambda: asm.gauss_grad(Sf, ff, vol, g3))
    psi = torch.zeros(N, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter(); perf = A.pcg(psi, t(case.source), "diagonal", tolerance=0.0, maxIter=400); torch.cuda.synchronize()
    rows["pcg_400_iterations_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
    out[order] = rows
    print(order, rows, flush=True)
    del A, addr, asm
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_exp_tile_order.json"), "w"), indent=1)
