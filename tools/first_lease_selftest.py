"""Set-up self-tests BETWEEN devices (tools/first_lease.sh step 1), one rank per GPU over RCCL: which of the window mechanisms
come up on this node, and how long a store takes to become visible on another device."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import torch.distributed as dist
import __graft_entry__ as graft

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
if rank == 0:
    graft.build()
dist.barrier()
pkg = graft.load_package()
from importlib import import_module
par = import_module(graft.PKG_NAME + ".parallel")
import workloads as wl
syn, eng = pkg.synthetic, pkg.engine
dev = torch.device("cuda", local)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = eng.Context(local, stream.cuda_stream)
res = {"rank": rank, "world": world}
os.environ["MI_ALLREDUCE"] = "auto"
comms = par.make_comms(ctx)
res["allreduce_windows"] = bool(comms[0].peer_mode)
if comms[0].peer_mode:
    st, fine = comms[0].peer_status()
    res["allreduce_window_fine_grained"] = bool(fine); res["allreduce_timeouts"] = int(st)
    x = torch.zeros(2, dtype=torch.float64, device=dev)
    comms[0].allreduce_sum(x); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(200): comms[0].allreduce_sum(x)
    torch.cuda.synchronize()
    res["window_allreduce_us"] = 1e6 * (time.perf_counter() - t0) / 200
parts = {2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(world, (1, 1, world))
sub = syn.box_subdomain((108 * parts[0], 108 * parts[1], 108 * parts[2]), parts, rank)
dm = par.DistributedMatrix(ctx, sub, dev, comms=comms)
used, bad = dm.mat.peer_halo_status()
res["halo_windows"] = bool(used); res["halo_timeouts"] = int(bad)
psi = torch.zeros(sub.n_cells, dtype=torch.float64, device=dev)
src = torch.from_numpy(sub.source).to(dev)
perf = dm.solve("PCG", psi, src, precond="diagonal", tolerance=0.0, maxIter=63)
res["pcg_persistent_kernel_launches"] = ctx.stat(1); res["barrier_litmus_runs"] = ctx.stat(2)
res["pcg_final_residual"] = float(perf["finalResidual"])
G = dm.gamg(wl.box_pair_weights(sub), 100)
psi.zero_()
pg = G.solve(dm.mat, psi, src, tolerance=1e-6, maxIter=100)
res["gamg_cycles_to_1e-6"] = int(pg["nIterations"]); res["gamg_cycles_replayed_as_hipGraph"] = ctx.stat(3)
res["timeouts_after_solves"] = {"halo": int(dm.mat.peer_halo_status()[1]), "allreduce": int(comms[0].peer_status()[0]) if comms[0].peer_mode else None}
allres = [None] * world
dist.all_gather_object(allres, res)
if rank == 0:
    print("[first_lease_selftest] " + json.dumps(allres))
dist.barrier()
dist.destroy_process_group()
