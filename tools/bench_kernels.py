"""Per-op timing on the 216^3 box, HIP events on the engine's stream: every hot op of SURVEY.md 8(a)
against its algorithmic bytes.  Writes gpurun_out/kernel_table.json / .md."""
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
sym = syn.box_case(*dims)
N, F = sym.n_cells, sym.n_faces
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
ctx = eng.Context(0, stream.cuda_stream)
addr = eng.Addressing(ctx, N, sym.lower_addr, sym.upper_addr)
A = eng.Matrix(addr); A.set_coeffs(t(sym.diag), t(sym.upper), None)
u = syn.splitmix_uniform(5, F)
lower_asym = t(sym.upper * (1.0 + 0.05 * u))
B = eng.Matrix(addr); B.set_coeffs(t(sym.diag), t(sym.upper), lower_asym)
E = lambda n: torch.empty(n, dtype=torch.float64, device=dev)
x, y, b = t(syn.splitmix_uniform(1, N) - 0.5), E(N), t(sym.source)
xe, ye = E(N), E(N)
addr.to_engine(x, xe)
asm = eng.Assembly(addr)
fl, fu, fd, ff = E(F), E(F), E(N), t(syn.splitmix_uniform(2, F))
fw = t(syn.splitmix_uniform(3, F))
rows = []

def timeit(name, fn, alg_bytes, reps=30, note=""):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    A.event_record(0)
    for _ in range(reps): fn()
    A.event_record(1)
    us = A.event_elapsed_ms(0, 1) / reps * 1e3
    gbs = alg_bytes / (us * 1e-6) / 1e9
    rows.append(dict(op=name, us=round(us, 1), algorithmic_MB=round(alg_bytes / 1e6, 1), GBps=round(gbs), frac_of_8TBps=round(gbs / 8000, 3), note=note))
    print(rows[-1], flush=True)

timeit("Amul symmetric (engine order)", lambda: A.amul_engine(xe, ye), 24 * N + 16 * F)
timeit("Amul asymmetric (engine order)", lambda: B.amul_engine(xe, ye), 24 * N + 24 * F)
timeit("Tmul asymmetric (engine order)", lambda: B.tmul_engine(xe, ye), 24 * N + 24 * F)
timeit("Amul symmetric (caller order: gather + Amul + scatter)", lambda: A.amul(x, y), 24 * N + 16 * F, note="boundary call incl. 2 permutation passes")
timeit("residual symmetric (caller order)", lambda: A.residual(x, b, y), 32 * N + 16 * F, note="incl. 3 permutation passes")
timeit("sumA (caller order)", lambda: A.sumA(y), 16 * N + 8 * F, note="incl. 1 permutation pass")
timeit("AINV precondition (caller order)", lambda: A.precondition("AINV", x, y), 32 * N + 16 * F, note="incl. 2 permutation passes")
psi = x.clone()
timeit("Jacobi smooth, 2 sweeps (caller order)", lambda: A.jacobi_smooth(psi, b, 2), 2 * (32 * N + 16 * F), note="incl. 3 permutation passes")
timeit("gSumProd (deterministic reduction)", lambda: ctx.sum_prod(x, b), 16 * N, note="includes host read-back")
timeit("set_coeffs symmetric (caller -> slots)", lambda: A.set_coeffs(fd, fu, None), 16 * N + 16 * F, note="K22 calcSortCoeffs analogue, once per assembled matrix")
timeit("fvm::laplacian (face pass + row pass)", lambda: asm.fvm_laplacian(ff, fw, fu, fd), 8 * N + 24 * F + 8 * N, note="reads delta,gamma; writes upper, diag; + row tables")
timeit("fvm::div (face pass + row pass)", lambda: asm.fvm_div(fw, ff, fl, fu, fd), 8 * N + 32 * F + 8 * N, note="reads w,phi; writes lower, upper, diag")
timeit("negSumDiag", lambda: asm.row_face_op(1, fl, fu, fd), 16 * N + 16 * F)
timeit("fvc::surfaceIntegrate", lambda: asm.surface_integrate(ff, None, y), 8 * N + 8 * F)
timeit("face interpolate", lambda: asm.face_interpolate(fw, x, fu), 8 * N + 16 * F + 8 * F)
# scheme front-end
nx = dims[0]; hh = 1.0 / nx
cc = np.arange(N)
Cx, Cy, Cz = t((cc % nx + 0.5) * hh), t(((cc // nx) % dims[1] + 0.5) * hh), t((cc // (nx * dims[1]) + 0.5) * hh)
Sf = [t(syn.splitmix_uniform(10 + k, F) * hh * hh) for k in range(3)]
g3 = [E(N) for _ in range(3)]
vol = t(np.full(N, hh ** 3))
timeit("fvc::grad (Gauss, 3 components)", lambda: asm.gauss_grad(Sf, ff, vol, g3), 8 * N + 32 * F + 24 * N, note="reads Sf(3), ssf, V; writes grad(3)")
timeit("limitedLinear weights (one face pass)", lambda: asm.limited_linear_weights(1.0, fw, ff, x, g3, [Cx, Cy, Cz], fu), 24 * F + 56 * N, note="reads cdw, flux, addr(8F); phi, grad(3), C(3) gathered; writes w")
timeit("fvm::ddt Euler", lambda: asm.fvm_ddt_euler(1e3, 1.0, vol, x, fd, y), 32 * N)
# whole solvers
A.set_coeffs(t(sym.diag), t(sym.upper), None)
for pre in ("diagonal", "AINV", "none"):
    z = torch.zeros(N, dtype=torch.float64, device=dev)
    A.pcg_begin(z, b, pre, tolerance=0.0, maxIter=500, history_len=0); A.pcg_iterate(10); torch.cuda.synchronize(); t0 = time.perf_counter()
    A.pcg_iterate(100); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
    A.pcg_end(None, 0)
    alg = (160 * N + (32 if pre == "AINV" else 16) * F)
    rows.append(dict(op=f"PCG iteration, {pre}", us=round(dt * 1e6, 1), algorithmic_MB=round(alg / 1e6, 1), GBps=round(alg / dt / 1e9), frac_of_8TBps=round(alg / dt / 8e12, 3), note="reference op sequence bytes (SURVEY 8d)"))
    print(rows[-1], flush=True)
asym = syn.box_case(*dims, symmetric=False)
C_ = eng.Matrix(addr); C_.set_coeffs(t(asym.diag), t(asym.upper), t(asym.lower))
for name, fn in (("PBiCG+DILU(AINV)", lambda p: C_.pbicg(p, b, "DILU", tolerance=1e-10, maxIter=100)),
                 ("PBiCGStab+DILU(AINV)", lambda p: C_.pbicgstab(p, b, "DILU", tolerance=1e-10, maxIter=100))):
    z = torch.zeros(N, dtype=torch.float64, device=dev); fn(z)
    z.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter(); perf = fn(z); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    rows.append(dict(op=f"{name} momentum-like solve to 1e-10", us=round(dt * 1e6, 1), iterations=perf["nIterations"], note="device-resident scalars, batches of 16"))
    print(rows[-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "kernel_table.json"), "w"), indent=1)
with open(os.path.join(ROOT, "gpurun_out", "kernel_table.md"), "w") as f:
    f.write(f"| op ({dims[0]}x{dims[1]}x{dims[2]}: N={N}, F={F}) | us | algorithmic MB | GB/s | frac of 8 TB/s | note |\n|---|---:|---:|---:|---:|---|\n")
    for r in rows:
        f.write(f"| {r['op']} | {r['us']} | {r.get('algorithmic_MB','')} | {r.get('GBps','')} | {r.get('frac_of_8TBps','')} | {r.get('note','')}{' its='+str(r['iterations']) if 'iterations' in r else ''} |\n")
