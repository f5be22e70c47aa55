"""One PISO-like time step on the 216^3 box, the per-rank workload of BASELINE config 4 (rhoPimpleFoam-style: full fvMatrix
assembly + PBiCG momentum + GAMG pressure) on ONE MI355X, every stage through the C ABI, timed with HIP events:
  momentum predictor: upwind weights, fvm::div, fvm::laplacian, fvm::ddt, fvMatrix operators (UEqn = ddt + div - laplacian),
                      relax, coefficient binding, PBiCG + DILU per component (relTol 0.1)
  pressure corrector: rAU = 1/A, face interpolation, fvm::laplacian(rAUf), coefficient binding, GAMG (relTol 0.05),
                      flux (faceH), fvc::div(phi) = surfaceIntegrate, fvc::grad(p) (Gauss), U -= rAU grad p
Synthetic fields of the right shapes (uniform box geometry); the point is the cost per stage, not the flow."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
graft.build()
pkg = graft.load_package()
syn, eng = pkg.synthetic, pkg.engine
dims = [int(v) for v in os.environ.get("DIMS", "216,216,216").split(",")]
nx = dims[0]
case = syn.box_case(*dims)
N, F = case.n_cells, case.n_faces
h = 1.0 / nx
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
E = lambda n: torch.empty(n, dtype=torch.float64, device=dev)
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
asm = eng.Assembly(addr)
lo, up = case.lower_addr.astype(np.int64), case.upper_addr.astype(np.int64)
dirn = np.where(up - lo == 1, 0, np.where(up - lo == nx, 1, 2))
Sf = [t(h * h * (dirn == d)) for d in range(3)]
magSf, delta, vol = t(np.full(F, h * h)), t(np.full(F, 1.0 / h)), t(np.full(N, h ** 3))
phi = t(0.3 * h * h * (dirn == 0) * (1.0 + 0.2 * (syn.splitmix_uniform(3, F) - 0.5)))
nuMagSf = t(np.full(F, 1e-3 * h * h))
U = [t(0.1 * (syn.splitmix_uniform(10 + d, N) - 0.5)) for d in range(3)]
p = t(np.zeros(N))
w = (1.0 / nx) * np.array([1.0, 1.01, 1.02])[dirn]
UM, PM = eng.Matrix(addr), eng.Matrix(addr)
G = eng.Gamg(addr, w, 100)
xmin = np.nonzero(np.arange(N) % nx == 0)[0]
patch = eng.Patch(ctx, N, xmin)
icU, icP = t(np.full(xmin.shape[0], 2.0 * 1e-3 * h)), t(np.full(xmin.shape[0], -2.0 * h))
wts, cl, cu, cd, lu, ld = E(F), E(F), E(F), E(N), E(F), E(N)
dd, ds = E(N), [E(N) for _ in range(3)]
ul, uu, ud = E(F), E(F), E(N)
rAU, rAUf, pu, pd, psrc = E(N), E(F), E(F), E(N), E(N)
fh, grad = E(F), [E(N) for _ in range(3)]
stages = {}
ev = {}
def stage(name):
    class S:
        def __enter__(self_):
            self_.a = torch.cuda.Event(enable_timing=True); self_.b = torch.cuda.Event(enable_timing=True); self_.a.record()
        def __exit__(self_, *x):
            self_.b.record(); ev.setdefault(name, []).append((self_.a, self_.b))
    return S()

def step():
    with stage("momentum: upwind weights + fvm::div + fvm::laplacian"):
        asm.upwind_weights(phi, wts); asm.fvm_div(wts, phi, cl, cu, cd); asm.fvm_laplacian(delta, nuMagSf, lu, ld)
    with stage("momentum: fvm::ddt x3 + UEqn = ddt + div - laplacian + boundary diag"):
        for d in range(3): asm.fvm_ddt_euler(1.0 / 1e-3, 1.0, vol, U[d], dd, ds[d])
        asm.axpby(1.0, cl, -1.0, lu, ul); asm.axpby(1.0, cu, -1.0, lu, uu)
        asm.axpby(1.0, dd, 1.0, cd, ud); asm.axpby(1.0, ud, -1.0, ld, ud)
        patch.add(icU, ud, 0)
    with stage("momentum: relax(0.7) + bind coefficients"):
        asm.relax(0.7, ud, ul, uu, ds[0], U[0])
        UM.set_coeffs(ud, uu, ul)
    its = []
    with stage("momentum: PBiCG + DILU, 3 components (relTol 0.1)"):
        for d in range(3): its.append(UM.pbicg(U[d], ds[d], "DILU", tolerance=1e-12, relTol=0.1, maxIter=50)["nIterations"])
    with stage("pressure: rAU, interpolate, fvm::laplacian(rAUf), bind, div(phi) source"):
        torch.reciprocal(ud, out=rAU); rAU.mul_(vol)
        asm.face_interpolate(wts, rAU, rAUf); rAUf.mul_(magSf)
        asm.fvm_laplacian(delta, rAUf, pu, pd); patch.add(icP, pd, 0)
        PM.set_coeffs(pd, pu, None)
        asm.surface_integrate(phi, None, psrc)
    with stage("pressure: GAMG (relTol 0.05)"):
        p.zero_()                       # same work every step: 3 V-cycles from a zero start (a restart from the old p takes fewer)
        gperf = G.solve(PM, p, psrc, tolerance=1e-12, relTol=0.05, maxIter=50)
        cyc = gperf["nIterations"]
        if os.environ.get("VERBOSE"): print({k: v for k, v in gperf.items() if k != "history"}, gperf["history"][:4], flush=True)
    with stage("corrector: flux (faceH), fvc::grad(p), U -= rAU grad p"):
        PM.faceH(p, fh); phi.sub_(fh * 0.0)
        asm.face_interpolate(wts, p, rAUf); asm.gauss_grad(Sf, rAUf, vol, grad)
        for d in range(3): asm.axpby(1.0, U[d], -1e-3, grad[d], U[d])
    return its, cyc

step(); torch.cuda.synchronize(); ev.clear()
t0 = time.perf_counter()
reps = int(os.environ.get("STEPS", "5"))
for _ in range(reps): its, cyc = step()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
out = {"workload": f"{dims[0]}x{dims[1]}x{dims[2]} hex box, N={N}, F={F}, fp64, 1 x MI355X", "ms_per_time_step_wall": 1e3 * wall,
       "pbicg_iterations_per_component": its, "gamg_cycles": cyc, "stages_ms": {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}}
print(json.dumps(out, indent=1))
