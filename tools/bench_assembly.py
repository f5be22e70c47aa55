"""Assembly row passes on the 216^3 box, HIP events on the engine's stream, algorithmic bytes as in tools/bench_kernels.py:
the caller's numbering (fixed blocks) and ordered addressing (blocks = the layout's tiles).
Writes gpurun_out/assembly_row_passes.json."""
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
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
E = lambda n: torch.empty(n, dtype=torch.float64, device=dev)
ctx = eng.Context(0, stream.cuda_stream)
case = syn.box_case(*dims)
N, F = case.n_cells, case.n_faces
out = {}

def run(tag, addr):
    asm = eng.Assembly(addr)
    A = eng.Matrix(addr)
    ff, fw = t(syn.splitmix_uniform(2, F)), t(syn.splitmix_uniform(3, F))
    fl, fu, fd, y = E(F), E(F), E(N), E(N)
    Sf = [t(syn.splitmix_uniform(10 + k, F)) for k in range(3)]
    vol = t(np.full(N, 1.0)); g3 = [E(N) for _ in range(3)]
    rows = {}
    def timeit(name, fn, alg, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            A.event_record(0)
            for _ in range(reps): fn()
            A.event_record(1)
            best = min(best, A.event_elapsed_ms(0, 1) / reps * 1e3)
        rows[name] = dict(us=round(best, 1), algorithmic_MB=round(alg / 1e6, 1), frac_of_8TBps=round(alg / (best * 1e-6) / 8e12, 3))
        print(tag, name, rows[name], flush=True)
    timeit("fvm::laplacian", lambda: asm.fvm_laplacian(ff, fw, fu, fd), 8 * N + 24 * F + 8 * N)
    timeit("fvm::div", lambda: asm.fvm_div(fw, ff, fl, fu, fd), 8 * N + 32 * F + 8 * N)
    timeit("negSumDiag (asymmetric)", lambda: asm.row_face_op(1, fl, fu, fd), 16 * N + 16 * F)
    timeit("negSumDiag (symmetric)", lambda: asm.row_face_op(1, None, fu, fd), 16 * N + 8 * F)
    timeit("fvc::surfaceIntegrate", lambda: asm.surface_integrate(ff, None, y), 8 * N + 8 * F)
    timeit("fvc::grad (Gauss, 3 components)", lambda: asm.gauss_grad(Sf, ff, vol, g3), 8 * N + 32 * F + 24 * N)
    # round 5: rhoPimpleFoam's phiHbyA = (interpolate(rho HbyA) & Sf) + rhorAUf ddtCorr and its surfaceIntegrate in ONE row pass, and
    # fvc::ddtCorr(rho, U, phi) in one face pass (algorithmic: every cell / face array once)
    lam, rho, aA, aB = t(syn.splitmix_uniform(20, F)), t(0.8 + syn.splitmix_uniform(21, N)), t(syn.splitmix_uniform(22, F)), t(syn.splitmix_uniform(23, F))
    V3 = [t(syn.splitmix_uniform(24 + k, N) - 0.5) for k in range(3)]
    timeit("phiHbyA + fvc::div(phiHbyA) (mi_flux_div: face pass + row sum)", lambda: asm.flux_div(lam, Sf, V3, fl, y, cell_scale=rho, add_a=aA, add_b=aB), 32 * N + 8 * N + (8 + 24 + 16 + 8) * F)
    # round 6: a whole transport equation in one row pass (mi_fvm_assemble) -- momentum-like: ddt(rho, U) + div(phi, U) [upwind] - laplacian(mu, U),
    # three sources and sumMagOffDiag (algorithmic: flux, delta, gamma in, lower, upper out = 40F; rho, rho0, V, 3 psi0 in, diag, sumMag, 3 sources out = 88N)
    srcs, mag = [E(N) for _ in range(3)], E(N)
    timeit("ddt + div - laplacian, 3 sources (mi_fvm_assemble)", lambda: asm.assemble(fu, fd, lower_out=fl, sources_out=srcs, ddt=dict(r_delta_t=1e4, vol=vol, psi_old=V3, rho=rho, rho_old=rho),
                                                                                 div=dict(flux=ff), laplacian=dict(delta_coeffs=fw, gamma_magsf=aB), sum_mag_out=mag), 40 * F + 88 * N)
    ps = E(N)
    timeit("ddt - laplacian, symmetric (mi_fvm_assemble)", lambda: asm.assemble(fu, fd, sources_out=[ps], ddt=dict(r_delta_t=1e4, vol=vol, psi_old=[V3[0]], rho=rho, rho_old=rho),
                                                                           laplacian=dict(delta_coeffs=fw, gamma_magsf=aB)), 24 * F + 48 * N)
    timeit("fvc::ddtCorr(rho, U, phi) (mi_ddt_phi_corr)", lambda: asm.ddt_phi_corr(1e4, lam, Sf, V3, rho, ff, fu), 32 * N + (8 + 24 + 8 + 8) * F)
    out[tag] = rows

run("caller numbering (blocks of 1024 cells; gradient 256)", eng.Addressing(ctx, N, case.lower_addr, case.upper_addr))
a0 = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
rc = syn.renumber(case, a0.cell_perm())
run("ordered addressing, blocks = tiles", eng.Addressing(ctx, N, rc.lower_addr, rc.upper_addr, ordered=True, tile_cell_start=a0.tile_starts()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "assembly_row_passes.json"), "w"), indent=1)
