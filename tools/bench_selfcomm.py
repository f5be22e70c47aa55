"""The 8-GPU share of the 10 M-cell benchmark (108^3 cells per rank) on ONE MI355X, communication included: the box is made
periodic in y and the two y-patches are posed as PROCESSOR patches whose neighbour is this rank, so every halo store, flag wait
and all-reduce of the N > 1 code path is really issued (to self).  Compares the phase loop over RCCL (7 kernels + 2
ncclAllReduce + 1 send/recv group per iteration) with the fused three-launch iteration over peer windows (peer.inc).

    python tools/bench_selfcomm.py [--dims 108 108 108] [--iters 400] [--mode both | rccl,peer3,peer4,peer5,peer4_t512] [--out file.json]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--dims", type=int, nargs=3, default=[108, 108, 108])
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--mode", default="both")
ap.add_argument("--out", default=None)
args = ap.parse_args()
graft.build()
pkg = graft.load_package()
from importlib import import_module
par = import_module(graft.PKG_NAME + ".parallel")
syn, eng = pkg.synthetic, pkg.engine
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
case = syn.add_cyclic_y(syn.box_case(*args.dims))
res = {"dims": args.dims, "cells": case.n_cells, "ext_values": int(sum(len(i.face_cells) for i in case.interfaces))}
modes = ["rccl", "peer5", "persist"] if args.mode == "both" else args.mode.split(",")
for mode in modes:
    os.environ["MI_ALLREDUCE"] = "rccl" if mode == "rccl" else "peer"
    # persist: the whole batch of iterations as ONE persistent cooperative kernel (csrc/persist.inc, DIST form)
    os.environ["MI_PCG_PERSIST"] = "1" if mode.startswith("persist") else "0"
    ctx = eng.Context(0, stream.cuda_stream)
    # peer3: every all-reduce inside the passes; peer4 (the default between devices): the two-scalar one inside the p-update,
    # wA.pA in a one-workgroup kernel; peer5: both in one-workgroup kernels (ranks that share a device)
    os.environ["MI_DPCG_FUSED"] = mode[4] if mode.startswith("peer") and len(mode) > 4 and mode[4] in "345" else "1"
    if "_t" in mode: os.environ["MI_TILE_CELLS"] = mode.split("_t")[1]       # e.g. peer5_t640: 640-cell tiles
    else: os.environ.pop("MI_TILE_CELLS", None)
    s = par.DistributedPCG(ctx, case, dev, precond="diagonal", n_global=case.n_cells)
    K = args.iters
    s.begin(tolerance=0.0, max_iter=5 * K + 64)
    s.iterate(32); torch.cuda.synchronize()
    reps = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s.iterate(K)
        host = time.perf_counter() - t0
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        reps.append((1e6 * el / K, 1e6 * host / K))
    a_ms = s.iterate(64, time_amul=True, event_stride=8); torch.cuda.synchronize()
    st = s.end()
    assert st["nIterations"] == 32 + 4 * K + 64, st
    used, bad = s.ops.mat.peer_halo_status()
    res[mode] = {"us_per_iteration": sorted(r[0] for r in reps)[len(reps) // 2], "host_enqueue_us_per_iteration": sorted(r[1] for r in reps)[len(reps) // 2],
                 "all_repeats_us": [r[0] for r in reps], "amul_us": 1e3 * a_ms / 64, "halo_windows": used, "wait_timeouts": bad,
                 "allreduce": s.allreduce, "persistent_kernel_launches": ctx.stat(1),
                 "launches_per_iteration": "1 per batch of iterations" if ctx.stat(1) else int(os.environ["MI_DPCG_FUSED"]) if used and os.environ["MI_DPCG_FUSED"] in "345" else 4 if used else "7 kernels + 2 ncclAllReduce + 1 ncclGroup(send, recv)"}
    print(mode, json.dumps(res[mode]), flush=True)
    del s
print(json.dumps(res))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)
