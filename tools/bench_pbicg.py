"""PBiCG (momentum-like asymmetric matrix): device-resident loop vs the host-stepped one (MI_PBICG_HOST_STEPPED=1)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); syn, eng = pkg.synthetic, pkg.engine
out = {}
for dims in ((108, 108, 108), (216, 216, 216)):
    case = syn.box_case(*dims, symmetric=False)
    ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
    mat.set_coeffs(t(case.diag), t(case.upper), t(case.lower))
    b = t(case.source)
    for pre in ("diagonal", "DILU"):
        for mode in ("0", "1", "0", "1"):
            os.environ["MI_PBICG_HOST_STEPPED"] = mode
            psi = torch.zeros(case.n_cells, dtype=torch.float64, device="cuda:0")
            mat.pbicg(psi, b, pre, tolerance=0.0, maxIter=3)
            psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            p = mat.pbicg(psi, b, pre, tolerance=0.0, maxIter=63)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out.setdefault(f"PBiCG {dims[0]}^3 {pre}", {}).setdefault("host_stepped" if mode == "1" else "device_resident", []).append(round(1e6 * dt / p["nIterations"], 1))
            psi.zero_(); mat.pbicgstab(psi, b, pre, tolerance=0.0, maxIter=3)
            psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            p = mat.pbicgstab(psi, b, pre, tolerance=0.0, maxIter=63)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out.setdefault(f"PBiCGStab {dims[0]}^3 {pre}", {}).setdefault("host_stepped" if mode == "1" else "device_resident", []).append(round(1e6 * dt / max(p["nIterations"], 1), 1))
print(json.dumps(out, indent=1))
