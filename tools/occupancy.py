"""resident workgroups per CU of the Amul kernel for a few LDS footprints (tile slot caps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
case = syn.box_case(64, 64, 64)
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
mat = eng.Matrix(addr)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
mat.set_coeffs(t(case.diag), t(case.upper), None)
print(addr.stats(), mat.occupancy())
