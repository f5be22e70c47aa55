"""Probe: can two RCCL ranks share ONE GPU on this stack?  (would allow real multi-rank tests on a 1-GPU box)"""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group(os.environ.get("PROBE_BACKEND", "nccl"), rank=rank, world_size=world)
t = torch.full((4,), float(rank + 1), device="cuda:0", dtype=torch.float64)
dist.all_reduce(t)
torch.cuda.synchronize()
print("rank", rank, "allreduce ->", t.tolist(), flush=True)
if rank == 0:
    dist.send(torch.arange(3, device="cuda:0", dtype=torch.float64), 1)
else:
    r = torch.empty(3, device="cuda:0", dtype=torch.float64); dist.recv(r, 0); print("rank 1 received", r.tolist(), flush=True)
dist.destroy_process_group()
