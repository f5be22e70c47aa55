"""The DECOMPOSED forms of BASELINE configs 3 / 4 / 5 on ONE MI355X with the communication really issued:
the box is periodic in y and the two y-patches are posed as PROCESSOR patches whose neighbour is this rank, so every halo store,
flag wait and all-reduce of the N > 1 path of GAMG (processor interfaces on every level), PBiCG + DILU (single and the batched
three-component solve) and of a whole PISO-like time step runs -- to self.  Next to each number: the same solver on the same
matrix with the y-pair as a LOCAL cyclic patch (no communicator: the single-rank path).

    python tools/bench_selfcomm_solvers.py [--dims 108 108 108] [--solver gamg,pbicg,timestep] [--transport peer|rccl] [--out f.json]

Environment A/B switches of the engine that matter here: MI_WIN_DIRECT=0 (boundary tiles behind k_halo_pull instead of reading
the window), MI_GAMG_GRAPH_ATTACHED=0 (attached V-cycle enqueued launch by launch).
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[108, 108, 108])
ap.add_argument("--solver", default="gamg,pbicg")
ap.add_argument("--transport", default="peer")
ap.add_argument("--cycles", type=int, default=20)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--out", default=None)
args = ap.parse_args()
graft.build()
pkg = graft.load_package()
from importlib import import_module
par = import_module(graft.PKG_NAME + ".parallel")
import workloads as wl
syn, eng = pkg.synthetic, pkg.engine
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
os.environ["MI_ALLREDUCE"] = "peer" if args.transport == "peer" else "rccl"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
res = {"dims": args.dims, "transport": args.transport}


def local_matrix(ctx, case):
    """the same matrix with the y-pair as a LOCAL cyclic patch: what one rank solves when nothing is decomposed"""
    a, b = case.interfaces
    addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr, [a.face_cells, b.face_cells], [b.face_cells, a.face_cells])
    mat = eng.Matrix(addr)
    mat.set_coeffs(t(case.diag), t(case.upper), None if case.lower is None else t(case.lower))
    for p, itf in enumerate(case.interfaces):
        mat.set_interface_coeffs(p, t(itf.bou_coeffs), None if case.lower is None else t(itf.int_coeffs))
    return addr, mat


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


solvers = args.solver.split(",")
FORMS = tuple(os.environ.get("MI_SELFCOMM_ONLY", "attached,local").split(","))   # MI_SELFCOMM_ONLY=attached: one form only (profiler runs)
if "gamg" in solvers:
    case = syn.add_cyclic_y(syn.box_case(*args.dims))
    w = wl.box_pair_weights(case)
    src = t(case.source)
    out = {}
    for form in FORMS:
        ctx = eng.Context(0, stream.cuda_stream)
        if form == "attached":
            dm = par.DistributedMatrix(ctx, case, dev, n_global=case.n_cells)
            G, mat = dm.gamg(w, 100), dm.mat
        else:
            addr, mat = local_matrix(ctx, case)
            G = eng.Gamg(addr, w, 100)
        psi = torch.zeros(case.n_cells, dtype=torch.float64, device=dev)
        def run():
            psi.zero_()
            p = G.solve(mat, psi, src, tolerance=0.0, maxIter=args.cycles)
            assert p["nIterations"] == args.cycles
        el = timed(run)
        psi.zero_(); pc = G.solve(mat, psi, src, tolerance=1e-6, maxIter=200)
        out[form] = {"ms_per_v_cycle": 1e3 * el / args.cycles, "levels": G.n_levels, "cycles_to_1e-6": int(pc["nIterations"]),
                     "final_residual": float(pc["finalResidual"])}
        if form == "attached":
            out[form]["halo_windows"], out[form]["wait_timeouts"] = mat.peer_halo_status()
        print("gamg", form, json.dumps(out[form]), flush=True)
    if "attached" in out and "local" in out: out["attached_over_local"] = out["attached"]["ms_per_v_cycle"] / out["local"]["ms_per_v_cycle"]
    res["gamg"] = out

if "pbicg" in solvers:
    case = syn.add_cyclic_y(syn.box_case(*args.dims, symmetric=False))
    srcs = [t(case.source * (1.0 + 0.1 * c) + 0.01 * c) for c in range(3)]
    out = {}
    for form in FORMS:
        ctx = eng.Context(0, stream.cuda_stream)
        if form == "attached":
            dm = par.DistributedMatrix(ctx, case, dev, n_global=case.n_cells)
            mat = dm.mat
        else:
            addr, mat = local_matrix(ctx, case)
        psis = [torch.zeros(case.n_cells, dtype=torch.float64, device=dev) for _ in range(3)]
        K = args.iters
        def single():
            psis[0].zero_()
            p = mat.pbicg(psis[0], srcs[0], "DILU", tolerance=0.0, maxIter=K - 1)
            assert p["nIterations"] == K
        def multi():
            for q in psis: q.zero_()
            ps = mat.pbicg_multi(psis, srcs, "DILU", tolerance=0.0, maxIter=K - 1)
            assert all(p["nIterations"] == K for p in ps)
        e1 = timed(single)
        row = {"single_us_per_iteration": 1e6 * e1 / K}
        try:
            e3 = timed(multi)
            row["batched3_us_per_component_iteration"] = 1e6 * e3 / (3 * K)
        except eng.MiError as e:
            row["batched3"] = f"refused: {e}"
        out[form] = row
        print("pbicg", form, json.dumps(row), flush=True)
    res["pbicg"] = out

if "timestep" in solvers:
    case = syn.add_cyclic_y(syn.box_case(*args.dims))
    out = {}
    for form in FORMS:
        ctx = eng.Context(0, stream.cuda_stream)
        if form == "attached":
            dm = par.DistributedMatrix(ctx, case, dev, n_global=case.n_cells)
            out[form] = wl.timestep_supplement(eng, syn, case, dm.addr, ctx, dev, steps=args.steps, coupled=dict(case=case, comms=dm.comms, n_global=case.n_cells))
        else:
            addr, _ = local_matrix(ctx, case)
            out[form] = wl.timestep_supplement(eng, syn, case, addr, ctx, dev, steps=args.steps, coupled=dict(case=case, comms=None, n_global=case.n_cells))
        print("timestep", form, json.dumps(out[form]), flush=True)
    if "attached" in out and "local" in out: out["attached_over_local"] = out["attached"]["ms_per_time_step"] / out["local"]["ms_per_time_step"]
    res["timestep"] = out

print(json.dumps(res))
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
