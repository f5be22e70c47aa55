#!/usr/bin/env python
"""bench.py -- PCG iterations/s + HBM GB/s on the 10M-cell lduMatrix (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one full PCG iteration of the reference's loop (PCG.C:133-204: precondition,
gSumProd, p update, Amul, gSumProd, psi/r update, gSumMag) on the 216^3 hex-box pressure
matrix (N = 10 077 696 cells, F = 30 093 120 faces), diagonal preconditioner, with all inputs
resident in HBM.  For N > 1 the same mesh is domain-decomposed over the ranks (strong scaling,
one process per GPU, halo exchange + global sums on RCCL).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel, lduMatrix::Amul
(tile_kernel<OP_AMUL>): algorithmic bytes 24N+16F per launch / its average launch duration,
measured with HIP events on the engine's stream inside the timed region.  `cpu_baseline` is
upstream OpenFOAM's CPU lduMatrix path as it runs on the node (one rank per host core over a slab decomposition,
face-loop Amul, diagonal PCG; oracle/baseline_oracle.c) timed on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / cross-process buffers); already exported on the boxes
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft  # noqa: E402
import workloads  # noqa: E402  (tools/workloads.py: configs 3 and 4/5 as callable measurements)
from workloads import gamg_cycle_bytes  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parts_for(n):
    return {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(n) or (1, 1, n)


def layout_source_hash():
    """what the committed PMC figure of an Amul launch is tied to: the tile kernel's source text and the tile layout's OUTPUT on
    fixed reference cases (tools/source_fingerprint.py) -- host-side changes that leave every layout table bit-identical keep
    the figure, anything that moves a slot or touches the kernel voids it"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import source_fingerprint
    return source_fingerprint.layout_source_hash()


def traffic_from_profile(nx, ny, nz, n_gpus):
    """HBM bytes per Amul launch from the committed rocprofv3 PMC passes (profiles/traffic_latest.json: FETCH_SIZE x 2 +
    WRITE_SIZE as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read inside this process, so the value is the
    one measured on this exact workload AND this exact kernel source / layout output (layout_source_hash, stored with the
    figure by tools/prof_round.sh); null for any other configuration or once those have changed."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
        if rec.get("workload") == f"{nx}x{ny}x{nz}" and rec.get("n_gpus") == n_gpus and rec.get("layout_source_sha256_16") == layout_source_hash() \
                and not any(os.environ.get(k) for k in ("MI_TILE_CELLS", "MI_TILE_SLOTS", "MI_ENTRY16", "MI_TILE_FLAGS", "MI_ENGINE_LIB")):
            return float(rec["amul_traffic_bytes_per_launch"]), str(rec.get("source", "profiles/traffic_latest.json")).split(" ")[0]
    except Exception:
        pass
    return None, None


def gamg_traffic_from_profile(nx, ny, nz):
    """HBM bytes per V-cycle from the committed PMC passes (profiles/traffic_latest.json "gamg": the difference of a 25- and a
    5-cycle solve, tools/gpu_r03_g.sh); quoted only for this workload and while the sources it was measured with are unchanged"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_fingerprint import gamg_source_hash
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))["gamg"]
        if rec.get("workload") == f"{nx}x{ny}x{nz}" and rec.get("gamg_source_sha256_16") == gamg_source_hash() \
                and not any(os.environ.get(k) for k in ("MI_TILE_CELLS", "MI_TILE_SLOTS", "MI_ENTRY16", "MI_TILE_FLAGS", "MI_ENGINE_LIB", "MI_SMALL_TILES")):
            return float(rec["traffic_bytes_per_cycle"])
    except Exception:
        pass
    return None


def cpu_baseline(case, syn, iters):
    """CPU leg: upstream OpenFOAM's lduMatrix path as it runs on this node -- one rank per core over a z-slab decomposition,
    face-loop Amul, diagonal PCG (oracle/baseline_oracle.c; one OpenMP thread plays each MPI rank).  Reported, not a target.
    iters <= 0: size the sample to about 10 s of wall time."""
    from oracle import oracle as orc
    cores = max(1, min(os.cpu_count() or 1, case.dims[2], 64))
    subs = syn.decompose_box(case, (1, 1, cores)) if cores > 1 else [case]
    S = orc.System(subs)
    src = np.concatenate([s.source for s in subs])
    _, sec, _ = S.baseline_pcg(src, 10)                      # touch memory + estimate the rate
    if iters <= 0:
        iters = int(min(2000, max(20, 10.0 / max(sec / 10, 1e-6))))
    n, sec, _ = S.baseline_pcg(src, iters)
    return n / sec, n, sec, cores


def bench_gamg(args, eng, syn, ctx, dev, json_fd):
    """BASELINE config 3: GAMG pressure solve (pair agglomeration, GaussSeidel(=Jacobi) smoother) on the 10 M-cell box, one GPU.
    A step = one V-cycle incl. its finest residual; value = V-cycles/s inside mi_gamg_solve (tolerance 0, K cycles per solve)."""
    import torch
    nx, ny, nz = args.dims
    K, W, R = args.steps, args.warmup, max(1, args.repeats)
    case = syn.box_case(nx, ny, nz)
    N, F = case.n_cells, case.n_faces
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d = case.upper_addr.astype(np.int64) - case.lower_addr
    w = (1.0 / nx) * np.array([1.0, 1.01, 1.02])[np.where(d == 1, 0, np.where(d == nx, 1, 2))]   # faceAreaPair weights of the box
    t0 = time.perf_counter()
    addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
    mat = eng.Matrix(addr); mat.set_coeffs(t(case.diag), t(case.upper), None)
    G = eng.Gamg(addr, w, 100)
    torch.cuda.synchronize()
    log(f"[bench] layout + GAMG hierarchy ({G.n_levels} levels) in {time.perf_counter() - t0:.1f}s")
    src = t(case.source)
    psi = torch.zeros(N, dtype=torch.float64, device=dev)
    G.solve(mat, psi, src, tolerance=0.0, maxIter=max(W, 2))           # warm-up: level matrices, scratch, the cycle graph
    rep_s = []
    for _ in range(R):
        psi.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        perf = G.solve(mat, psi, src, tolerance=0.0, maxIter=K)
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t0)
        h = perf["history"]
        above = h > 1e-12                     # (past ~70 cycles the residual sits at the rounding floor; the cycles cost the same)
        assert perf["nIterations"] == K and np.all(np.diff(h)[above[1:]] < 0), perf
    elapsed = float(np.median(rep_s))
    levels = [(G.level_sizes(l)["n_coarse"], G.level_sizes(l)["n_coarse_faces"]) for l in range(G.n_levels)]
    ctl = dict(nPostSweeps=2, postSweepsLevelMultiplier=1, maxPostSweeps=4, nFinestSweeps=2)
    alg = gamg_cycle_bytes(levels, N, F, ctl)
    # to convergence, next to diagonal PCG (what the multigrid buys)
    psi.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
    pc = G.solve(mat, psi, src, tolerance=1e-6, maxIter=200); torch.cuda.synchronize(); t_conv = time.perf_counter() - t0
    out = {
        "metric": "PCG iterations/s + HBM GB/s on 10M-cell lduMatrix at 1/2/4/8 MI355X",
        "value": K / elapsed, "unit": "V-cycles/s", "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE config 3: {nx}x{ny}x{nz} hex box (N={N}, F={F}), GAMG pressure solve: faceAreaPair agglomeration, "
                               f"{G.n_levels} levels down to {levels[-1][0]} cells, GaussSeidel(=Jacobi) smoother 2 finest / 2..4 post sweeps, "
                               "correction scaling, direct coarsest solve, 1xMI355X",
                   "solver": "GAMG", "cells": N, "faces": F, "levels": [n for n, _ in levels],
                   "timing": f"median of {R} repeats of mi_gamg_solve with {K} V-cycles each (tolerance 0; the solve's prologue -- Amul, normFactor -- is inside the clock)",
                   "repeat_ms_per_step": [1e3 * x / K for x in rep_s],
                   "solve_to_1e-6": {"cycles": pc["nIterations"], "seconds": t_conv}},
        "roofline": {"kernel": "whole V-cycle (tile_kernel<OP_JACOBI> / <OP_AMUL> on every level + transfer and scaling passes)", "bound": "hbm",
                     "achieved": alg / (elapsed / K) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_cycle": alg, "traffic": gamg_traffic_from_profile(nx, ny, nz),
                     "traffic_unit": "bytes per V-cycle (rocprofv3 PMC: FETCH_SIZE x 2 + WRITE_SIZE over all dispatches, 25-cycle minus 5-cycle solve; profiles/r03_g_gamg_cycle_traffic.json)"
                                     if gamg_traffic_from_profile(nx, ny, nz) is not None else "not measured for this configuration"},
    }
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def negotiate_peer_path(solver, make_solver, trial, persist_launches, all_ok, any_rank, set_persist, force_rccl):
    """The trial + fallback chain of the multi-GPU bench (persistent kernel over peer windows -> five launches over peer windows -> RCCL),
    as a function of its collaborators so that tests/test_bench_contract.py can drive it between real ranks on the CPU with stand-ins:
      trial(solver) -> (came through on THIS rank, history | None)     a few iterations
      persist_launches() -> counter of persistent-kernel launches on this rank
      all_ok(flag) / any_rank(flag) -> the MIN / MAX of flag over the ranks (collectives: every rank must call them in the same order)
      set_persist(0 | 1), force_rccl(), make_solver() -> a fresh solver (after force_rccl: on RCCL)
    Returns (solver, note): the solver the timed region runs on and why it is not the first choice ("" if it is).  Every rank takes the same
    branch at every step -- each decision is made from values that were reduced over the ranks -- so the ranks issue the same collectives."""
    note = ""
    before = persist_launches()
    ok, h_persist = trial(solver)
    # the sub-domain fits the persistent kernel (csrc/persist.inc): decided from ALL ranks' counters -- a rank whose trial
    # threw before its launch was counted would otherwise issue a different sequence of collectives below
    persistent = any_rank(persist_launches() > before)
    good = all_ok(ok)
    if persistent:
        # the same 12 iterations through the five-launch loop: the persistent kernel must reproduce them
        set_persist(0)
        ok5, h5 = trial(solver) if good else (0, None)
        same = int(bool(good and ok5 and h_persist is not None and h5 is not None and float(np.max(np.abs(h_persist - h5))) < 1e-10 * float(h5[0])))
        if all_ok(same):
            set_persist(1)
        else:
            note = " (the persistent kernel's trial did not reproduce the five-launch loop on every rank: five launches)"
            if not good:                                   # windows possibly out of step after a timed-out wait: once more from scratch
                del solver
                solver = make_solver()
                good = all_ok(trial(solver)[0])
            else:
                good = all_ok(ok5)
    if not good:
        force_rccl()
        del solver
        solver = make_solver()
        note = " (the peer-window trial did not come through on every rank: RCCL)"
    return solver, note


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dims", type=int, nargs=3, default=[216, 216, 216])
    ap.add_argument("--precond", default="diagonal")
    ap.add_argument("--cpu-iters", type=int, default=int(os.environ.get("MI_BENCH_CPU_ITERS", "0")))  # 0: about 10 s
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--repeats", type=int, default=5)   # median of 5 repeats of the timed region
    ap.add_argument("--solver", choices=["pcg", "gamg"], default="pcg")   # gamg: BASELINE config 3 (a second mode, 1 GPU; a step = one V-cycle)
    ap.add_argument("--no-supplements", action="store_true")   # skip config.supplements (configs 3 and 4/5 on the same box, N = 1 only)
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher (one rank per GPU, as the driver's
    # `python -m torch.distributed.run --nproc-per-node N ... bench.py` does) instead of silently measuring one GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import subprocess
        # (--standalone: the launcher hosts the rendezvous itself on a port it binds, instead of a port found free here and bound a
        #  moment later by someone else -- EADDRINUSE was seen in 2 of ~60 such rendezvous in one GPU-suite run of round 5)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--standalone", "--local-addr", "127.0.0.1",
               os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[bench] --gpus {args.gpus} without a launcher: re-executing under torch.distributed.run")
        raise SystemExit(subprocess.call(cmd))

    # Only the JSON line may reach stdout: libraries (RCCL prints a version banner through C stdio at exit)
    # write to fd 1 behind Python's back, so fd 1 is pointed at stderr and the JSON goes to a private copy.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        log(f"warning: --gpus {args.gpus} != WORLD_SIZE {world}; using WORLD_SIZE")
    n_gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    if os.environ.get("MI_BENCH_BACKEND", "nccl") == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU over RCCL; "
                         "MI_BENCH_BACKEND=gloo rehearses the code path with ranks sharing devices)")
    # MI_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices, the
    # exchange goes through gloo and the torch.distributed loop; timings of such a run mean nothing).  Default: RCCL.
    backend = os.environ.get("MI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("MI_BENCH_FORCE_DIST") == "2":  # "2": also run the RCCL calls on a 1-rank group
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            os.environ.setdefault("MI_DPCG_DRIVER", "torch")
            dist.init_process_group(backend, rank=rank, world_size=world)

    if world > 1:          # one rank compiles (if anything is stale), the others wait: no concurrent writers of the .so files
        if rank == 0:
            graft.build()
        dist.barrier()
    graft.build()
    pkg = graft.load_package()
    syn, eng = pkg.synthetic, pkg.engine
    dev = torch.device("cuda", local_rank)
    nx, ny, nz = args.dims
    t0 = time.perf_counter()
    N, F = nx * ny * nz, 3 * nx * ny * nz - (ny * nz + nx * nz + nx * ny)
    force_dist = bool(os.environ.get("MI_BENCH_FORCE_DIST"))  # exercise the N>1 code path on one GPU
    single = world == 1 and not force_dist
    # N = 1: the whole case (the CPU baseline needs it too); N > 1: every rank builds only its own sub-domain
    case = syn.box_case(nx, ny, nz) if (single or (rank == 0 and not args.no_cpu and world == 1)) else None
    if rank == 0:
        log(f"[bench] case {nx}x{ny}x{nz}: N={N} F={F} built in {time.perf_counter() - t0:.1f}s")

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    # one explicit HIP stream shared by torch (copies, RCCL collectives) and the engine's kernels
    if os.environ.get("MI_BENCH_NULL_STREAM"):   # A/B hook: engine-owned stream next to torch's legacy default stream
        ctx = eng.Context(local_rank, None)
    else:
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        ctx = eng.Context(local_rank, stream.cuda_stream)
    if args.solver == "gamg":
        return bench_gamg(args, eng, syn, ctx, dev, json_fd)
    K, W, R = args.steps, args.warmup, max(1, args.repeats)
    amul_ms = None
    rep_s, rep_amul_ms, amul_alone_us = [], [], None
    host_enqueue_us = None
    host_loop = "single-GPU device-resident pipeline (mi_pcg_iterate)"
    allreduce_kind = "none (one rank)"
    path_taken, fallback_reason = "single-gpu five-launch pipeline", None
    fused_rp = False
    weak = None
    supplements = {}
    if single:
        t0 = time.perf_counter()
        addr = eng.Addressing(ctx, N, case.lower_addr, case.upper_addr)
        mat = eng.Matrix(addr)
        mat.set_coeffs(t(case.diag), t(case.upper), None)
        src = t(case.source)
        psi0 = torch.zeros(N, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        log(f"[bench] engine layout: {addr.stats()} in {time.perf_counter() - t0:.1f}s")
        mat.pcg_begin(psi0, src, args.precond, tolerance=0.0, relTol=0.0, maxIter=W + R * K + 8, history_len=W + R * K + 2)
        mat.pcg_iterate(W)
        ev_stride = int(os.environ.get("MI_BENCH_EVENT_STRIDE", "4"))
        for _ in range(R):                       # R repeats of the timed region of EXACTLY K steps; the median is reported
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a_ms = mat.pcg_iterate(K, time_amul=True, event_stride=ev_stride)  # HIP events around every 4th Amul launch of the timed region
            torch.cuda.synchronize()
            rep_s.append(time.perf_counter() - t0); rep_amul_ms.append(a_ms)
        perf = mat.pcg_end(None, history_len=W + R * K + 2)
        assert perf["nIterations"] == W + R * K, perf          # every timed step really iterated (no device-side early exit)
        fused_rp = ctx.stat(4) > 0
        if fused_rp:                                             # (csrc/pcg_fused.inc ran: three launches per iteration instead of five)
            host_loop += ": Amul, fold, then ONE launch for residual update + test + next direction (z = rD o rA stays on the chip)"
            path_taken = "single-gpu pipeline, fused residual / direction update (csrc/pcg_fused.inc)"
        assert np.all(np.isfinite(perf["history"])) and perf["history"][-1] < perf["history"][0]
        n_amul_cells, n_amul_faces = N, F
        # Amul alone, out of the solver loop: rotating vectors (4 input/output pairs = 645 MB > the 256 MiB Infinity Cache),
        # so that no sample finds its vectors on-die; kernel-bracketing events on the engine's stream
        nrot, nl = 4, 40
        xs = [torch.full((N,), 1.0 + 0.1 * i, dtype=torch.float64, device=dev) for i in range(nrot)]
        ys = [torch.empty(N, dtype=torch.float64, device=dev) for _ in range(nrot)]
        for i in range(nrot):
            mat.amul_engine(xs[i], ys[i])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        samples = []
        for _ in range(R):
            torch.cuda.synchronize(); e0.record()
            for i in range(nl):
                mat.amul_engine(xs[i % nrot], ys[i % nrot])
            e1.record(); torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e3 / nl)
        amul_alone_us = float(np.median(samples))
        del xs, ys
        # BASELINE configs 3 and 4/5 on the same box, reported beside `value` (never part of it): the driver's record then carries
        # the GAMG cycle rate and the cost of a whole time step as well (VERDICT r02 "missing" 3, 6)
        if not args.no_supplements and not os.environ.get("MI_BENCH_NO_SUPPLEMENTS"):
            try:
                t0 = time.perf_counter()
                sup_gamg, G = workloads.gamg_supplement(eng, case, addr, mat, dev)
                sup_gamg["traffic_bytes_per_cycle"] = gamg_traffic_from_profile(nx, ny, nz)   # PMC (profiles/traffic_latest.json), null off this workload
                supplements["gamg_216"] = sup_gamg
                log(f"[bench] supplement GAMG: {sup_gamg['ms_per_v_cycle']:.3f} ms per V-cycle ({time.perf_counter() - t0:.1f}s)")
                t0 = time.perf_counter()
                supplements["timestep_216"] = workloads.timestep_supplement(eng, syn, case, addr, ctx, dev, gamg=G, steps=3)
                log(f"[bench] supplement time step: {supplements['timestep_216']['ms_per_time_step']:.2f} ms ({time.perf_counter() - t0:.1f}s)")
                # BASELINE config 5's own solver: rhoPimpleFoam's UEqn / EEqn / pEqn (non-transonic, and the transonic form whose pressure
                # matrix is asymmetric), per-rank share, on the same box
                t0 = time.perf_counter()
                try:
                    supplements["rhopimple_timestep_216"] = {"non_transonic": workloads.rhopimple_supplement(eng, syn, case, addr, ctx, dev, gamg=G, steps=3),
                                                             "transonic": workloads.rhopimple_supplement(eng, syn, case, addr, ctx, dev, gamg=G, steps=3, transonic=True)}
                    log(f"[bench] supplement rhoPimpleFoam step: {supplements['rhopimple_timestep_216']['non_transonic']['ms_per_time_step']:.2f} ms, transonic "
                        f"{supplements['rhopimple_timestep_216']['transonic']['ms_per_time_step']:.2f} ms ({time.perf_counter() - t0:.1f}s)")
                except Exception as e:
                    supplements["rhopimple_timestep_216"] = {"error": f"{type(e).__name__}: {e}"}
                    log(f"[bench] rhoPimpleFoam supplement failed: {type(e).__name__}: {e}")
                del G
            except Exception as e:  # a supplement must never cost the headline line
                supplements["error"] = f"{type(e).__name__}: {e}"
                log(f"[bench] supplements failed: {supplements['error']}")
    else:
        from importlib import import_module
        par = import_module(graft.PKG_NAME + ".parallel")
        sub = syn.box_subdomain((nx, ny, nz), parts_for(world), rank)   # == decompose_box(box_case(...))[rank], built directly
        solver = par.DistributedPCG(ctx, sub, dev, precond=args.precond)
        # Peer windows (halo + all-reduce as stores into the neighbours' memory) are the default between ranks; their set-up
        # self-tests and the ranks agree on it.  On top of that a TRIAL solve: a few iterations, every rank reports whether it
        # came through (no wait ran out of polls, iteration count right); unless all did, every rank rebuilds on RCCL.
        peer_note = ""
        if world > 1 and getattr(solver.comms[0] if solver.comms else None, "peer_mode", False):
            def trial(sv):
                """12 iterations; (came through on THIS rank, history)"""
                try:
                    sv.begin(tolerance=0.0, max_iter=64)
                    sv.iterate(12)
                    st = sv.end()
                    return int(st["nIterations"] == 12 and bool(np.all(np.isfinite(st["history"][:13])))), np.array(st["history"][:13])
                except Exception as e:  # MiError: a window wait / a grid barrier ran out of polls
                    log(f"[bench] rank {rank}: peer-window trial failed: {e}")
                    return 0, None

            def all_ok(ok):
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1

            def any_rank(flag):                                    # the same value on every rank: they branch on it together
                v = torch.tensor([int(bool(flag))], dtype=torch.int32, device=dev)
                dist.all_reduce(v, op=dist.ReduceOp.MAX)
                return int(v.item()) == 1

            def force_rccl():
                os.environ["MI_ALLREDUCE"] = "rccl"

            solver, peer_note = negotiate_peer_path(solver, lambda: par.DistributedPCG(ctx, sub, dev, precond=args.precond), trial, lambda: ctx.stat(1),
                                                    all_ok, any_rank, lambda v: ctx.set_option("pcg_persist", v), force_rccl)
        host_loop = {"native": "C++ loop over RCCL (mi_dpcg_comm_iterate)", "torch": "torch.distributed loop (parallel.py)"}[solver.driver]
        solver.begin(tolerance=0.0, max_iter=W + (R + 1) * K + 8)
        before = ctx.stat(1)
        solver.iterate(W)
        in_kernel = ctx.stat(1) > before                           # batches run as ONE persistent cooperative kernel each
        if solver.driver == "native" and getattr(solver.comms[0], "peer_mode", False):
            host_loop = ("C++ loop, no collective calls: halo and all-reduce through peer windows, " +
                         ("one persistent cooperative kernel per batch of iterations (csrc/persist.inc)" if in_kernel else "five launches per iteration") +
                         " (mi_dpcg_comm_iterate)")
        allreduce_kind = (getattr(solver, "allreduce", "torch.distributed") + peer_note) if world > 1 else "none (one rank)"
        # the same, machine-readable (VERDICT r04 "next" 9): which inner loop the timed region ran and, if it is not the first choice, why
        if world == 1:
            path_taken = "single-gpu pipeline, fused residual / direction update (csrc/pcg_fused.inc)" if fused_rp else "single-gpu five-launch pipeline"
        elif solver.driver == "native" and getattr(solver.comms[0], "peer_mode", False):
            path_taken = "peer windows, persistent kernel" if in_kernel else "peer windows, five launches"
        elif solver.driver == "native":
            path_taken = "rccl, five launches"
        else:
            path_taken = "torch.distributed loop"
        fallback_reason = peer_note.strip(" ()") or None
        for _ in range(R):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # sampled Amul events: their records cost ~3 us each in this latency-bound loop; the persistent kernel has no Amul
            # launch to bracket -- its repeats run bare and the Amul of the same sub-domain is sampled in one more batch below
            a_ms = solver.iterate(K, time_amul=not in_kernel, event_stride=8)
            host_enqueue_us = 1e6 * solver.last_enqueue_s / K   # host time to enqueue one iteration (incl. collectives)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            el = time.perf_counter() - t0
            if world > 1:                                          # the slowest rank's clock, per repeat
                tmax = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                el = float(tmax.item())
            rep_s.append(el); rep_amul_ms.append(a_ms)
        if in_kernel:
            a_ms = solver.iterate(K, time_amul=True, event_stride=8)
            torch.cuda.synchronize()
            rep_amul_ms = [a_ms] * len(rep_s)
        else:
            solver.iterate(K); torch.cuda.synchronize()
        perf = solver.end()
        assert perf["nIterations"] == W + (R + 1) * K, perf
        n_amul_cells, n_amul_faces = sub.n_cells, sub.n_faces + sum(len(i.face_cells) for i in sub.interfaces)
        # supplement (not `value`): the same solver with the per-GPU work held at the N = 1 size (weak scaling;
        # at 8 GPUs this is the 80 M-cell box of BASELINE config 5), so that the latency-bound strong-scaling number
        # above can be read beside the regime the halo/all-reduce overlap is designed for
        if world > 1 or os.environ.get("MI_BENCH_WEAK"):
            px, py, pz = parts_for(world)
            del solver
            wsub = syn.box_subdomain((nx * px, ny * py, nz * pz), (px, py, pz), rank)
            ws = par.DistributedPCG(ctx, wsub, dev, precond=args.precond)
            ws.begin(tolerance=0.0, max_iter=W + K + 8)
            ws.iterate(W)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            ws.iterate(K)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            wel = time.perf_counter() - t0
            if world > 1:
                tmax = torch.tensor([wel], dtype=torch.float64, device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                wel = float(tmax.item())
            wperf = ws.end()
            assert wperf["nIterations"] == W + K, wperf
            weak = {"cells_per_gpu": wsub.n_cells, "global_cells": nx * px * ny * py * nz * pz, "iterations_per_s": K / wel,
                    "ms_per_step": 1e3 * wel / K, "cell_iterations_per_s": nx * px * ny * py * nz * pz * K / wel}
            # configs 3 / 4 / 5 as decomposed workloads (round 4): GAMG V-cycles and a whole time step per rank, attached
            # matrices, on the weak sub-domain (216^3 per rank: at 8 GPUs the 80 M-cell box of config 5) and GAMG also on the
            # strong share of the 10 M-cell box.  Never part of `value`.
            if not args.no_supplements and not os.environ.get("MI_BENCH_NO_SUPPLEMENTS") and ws.comms is not None:
                def rmax(v):
                    if world == 1:
                        return float(v)
                    tm = torch.tensor([float(v)], dtype=torch.float64, device=dev)
                    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                    return float(tm.item())
                ok_sup = 1
                try:
                    # (MI_BENCH_DECOMP_CYCLES / _STEPS: shorter supplements for the rehearsals between processes that SHARE a GPU, where every
                    #  exchange costs a scheduling quantum -- tests/test_bench_contract.py)
                    dkw = dict(cycles=int(os.environ.get("MI_BENCH_DECOMP_CYCLES", "10")), steps=int(os.environ.get("MI_BENCH_DECOMP_STEPS", "3")))
                    d = workloads.decomposed_supplements(eng, syn, par, wsub, ctx, dev, ws.comms, nx * px * ny * py * nz * pz, reduce_max=rmax, **dkw)
                    supplements[f"decomposed_{wsub.n_cells}_cells_per_rank"] = d
                    d2 = workloads.decomposed_supplements(eng, syn, par, sub, ctx, dev, ws.comms, N, reduce_max=rmax, **dkw)
                    supplements[f"decomposed_{sub.n_cells}_cells_per_rank"] = d2
                except Exception as e:  # a supplement must never cost the headline line (the other ranks' waits are bounded: they end up here too)
                    ok_sup = 0
                    supplements["error"] = f"{type(e).__name__}: {e}"
                    log(f"[bench] rank {rank}: decomposed supplements failed: {supplements['error']}")
                if world > 1:
                    fl = torch.tensor([ok_sup], dtype=torch.int32, device=dev)
                    dist.all_reduce(fl, op=dist.ReduceOp.MIN)
                    if int(fl.item()) != 1 and "error" not in supplements:
                        supplements["error"] = "another rank failed in the decomposed supplements"

    mid = int(np.argsort(rep_s)[len(rep_s) // 2])             # the median repeat: its wall clock and ITS Amul events
    elapsed, amul_ms = rep_s[mid], rep_amul_ms[mid]
    its = K / elapsed
    amul_avg_s = (amul_ms / 1e3) / K
    amul_bytes = 24 * n_amul_cells + 16 * n_amul_faces  # symmetric Amul, per launch (per rank)
    achieved = amul_bytes / amul_avg_s / 1e9
    out = {
        "metric": "PCG iterations/s + HBM GB/s on 10M-cell lduMatrix at 1/2/4/8 MI355X",
        "value": its,
        "unit": "iterations/s",
        "n_gpus": n_gpus,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{nx}x{ny}x{nz} hex box (N={N}, F={F}), scalar PCG + {args.precond} preconditioner, "
                        f"symmetric pressure-like lduMatrix, {n_gpus}xMI355X",
            "cells": N, "faces": F, "parallelism": f"domain-decomposition {parts_for(world)}" if world > 1 else "single GPU",
            "pcg_algorithmic_GBps": (160 * N + 16 * F) * its / 1e9,
            "host_enqueue_us_per_step": host_enqueue_us, "host_loop": host_loop,
            "allreduce": allreduce_kind, "path_taken": path_taken, "fallback_reason": fallback_reason,
            "timing": f"median of {R} repeats of the timed region of {K} steps (barrier + synchronize on both sides of each repeat, max over ranks)",
            "repeat_ms_per_step": [1e3 * t / K for t in rep_s], "repeat_amul_us_in_loop": [1e3 * a / K for a in rep_amul_ms],
            "amul_alone_us_rotating_buffers": amul_alone_us,
            "amul_alone_frac_of_peak": (None if amul_alone_us is None else (24 * N + 16 * F) / (amul_alone_us * 1e-6) / 1e9 / HBM_PEAK_GBS),
            "weak_scaling_supplement": weak,
            "supplements": supplements or None,
        },
        "roofline": {
            "kernel": "tile_kernel<OP_AMUL> (lduMatrix::Amul)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": amul_bytes,
            "traffic_unit": (f"bytes per launch (rocprofv3 PMC, {traffic_from_profile(nx, ny, nz, n_gpus)[1]})"
                             if traffic_from_profile(nx, ny, nz, n_gpus)[0] is not None else "not measured for this configuration"),
            "avg_launch_us": amul_avg_s * 1e6,
            "traffic": traffic_from_profile(nx, ny, nz, n_gpus)[0],
        },
    }
    if rank == 0:
        if not args.no_cpu and world == 1:
            if case is None:
                case = syn.box_case(nx, ny, nz)
            v, n_it, dt, cores = cpu_baseline(case, syn, args.cpu_iters)
            out["cpu_baseline"] = {"value": v, "unit": "iterations/s", "cores": cores, "kind": "port",
                                   "sample": f"{n_it} diagonal-PCG iterations of the same {nx}x{ny}x{nz} matrix in {dt:.1f}s: upstream OpenFOAM "
                                             f"face-loop Amul, z-slab decomposition into {cores} domains, one rank (OpenMP thread) per "
                                             "host core, oracle/baseline_oracle.c, gcc -O3"}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
