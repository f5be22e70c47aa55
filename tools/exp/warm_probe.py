"""How long does a fresh MI355X box need before the PCG loop runs at its steady rate?  Prints the average iteration time of
consecutive batches of 500 iterations for ~70 s of continuous work (round 2: first minute of a fresh lease is ~10 % slower)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as graft
import torch
pkg = graft.load_package(); syn, eng = pkg.synthetic, pkg.engine
dev = torch.device("cuda", 0)
case = syn.box_case(216, 216, 216)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = eng.Context(0, stream.cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
src = t(case.source); psi0 = torch.zeros(case.n_cells, dtype=torch.float64, device=dev)
total = int(os.environ.get("WARM_SECONDS", "70"))
t_start = time.perf_counter()
while time.perf_counter() - t_start < total:
    mat.pcg_begin(psi0, src, "diagonal", tolerance=0.0, relTol=0.0, maxIter=100000, history_len=4)
    for b in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mat.pcg_iterate(500)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"t={time.perf_counter() - t_start:6.1f}s  {dt / 500 * 1e6:7.1f} us/iteration", flush=True)
    mat.pcg_end(None, history_len=4)
    psi0.zero_()
