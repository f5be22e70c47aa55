"""per-solve GAMG fixed cost at 216^3: level matrices re-agglomerated from the fine coefficients (GAMGSolver.C:88-97) + the dense
inverse of the coarsest level, by where the inverse is formed (registers of one workgroup / host / global-memory kernel)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
g.build()
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
from oracle import oracle as orc
import ctypes as C
case = syn.box_case(216, 216, 216)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
out = {}
for mode, env in (("registers (default)", {}), ("host", {"MI_GAMG_DEVICE_INVERT": "0"}), ("global-memory kernel", {"MI_GAMG_DEVICE_INVERT": "1", "MI_GAMG_REG_INVERT": "0"})):
    for k in ("MI_GAMG_DEVICE_INVERT", "MI_GAMG_REG_INVERT"): os.environ.pop(k, None)
    os.environ.update(env)
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
    G = eng.Gamg(addr, orc.box_face_weights(case), 100)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = eng.lib().mi_gamg_update(G.h, mat.h)          # agglomerate every level + coarsest inverse + synchronise
        assert rc == 0
        ts.append(1e3 * (time.perf_counter() - t0))
    out[mode] = {"ms_per_update": round(float(np.median(ts[1:])), 3), "coarsest_cells": G.level_sizes(G.n_levels - 1)["n_coarse"]}
    del G, mat, addr
print(json.dumps({"workload": "216^3 box, 16 levels, mi_gamg_update = agglomerateMatrix on every level + coarsest inverse", "result": out}, indent=1))
