"""One process, repeated teardown / re-allocation of the whole engine state: does the PCG rate depend on WHERE the
arrays land (physical placement / fragment sizes), and does cycling a big allocation first change it?"""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); syn, eng = pkg.synthetic, pkg.engine
dev = torch.device("cuda", 0)
case = syn.box_case(216, 216, 216)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
t_start = time.perf_counter()

def cycle(tag):
    ctx = eng.Context(0, stream.cuda_stream)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
    src = t(case.source); psi0 = torch.zeros(case.n_cells, dtype=torch.float64, device=dev)
    mat.pcg_begin(psi0, src, "diagonal", tolerance=0.0, relTol=0.0, maxIter=100000, history_len=4)
    mat.pcg_iterate(50)
    res = []
    for b in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ms = mat.pcg_iterate(400, time_amul=True, event_stride=4)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res.append((dt / 400 * 1e6, ms / 400 * 1e3))
    mat.pcg_end(None, history_len=4)
    print(f"t={time.perf_counter() - t_start:6.1f}s {tag:14s} " + "  ".join(f"{a:6.1f} us/it (amul {b:5.1f})" for a, b in res), flush=True)
    del mat, addr, ctx, src, psi0
    gc.collect(); torch.cuda.empty_cache()

for i in range(5):
    cycle(f"cycle {i}")
free, total = torch.cuda.mem_get_info()
big = torch.empty(int(free * 0.9), dtype=torch.uint8, device=dev); big.fill_(1); torch.cuda.synchronize()
del big; gc.collect(); torch.cuda.empty_cache()
print(f"cycled {free * 0.9 / 2**30:.0f} GiB", flush=True)
for i in range(5, 9):
    cycle(f"cycle {i}")
