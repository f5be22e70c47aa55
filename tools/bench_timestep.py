"""One PISO-like time step on the 216^3 box, the per-rank workload of BASELINE configs 4 / 5 on ONE MI355X (tools/workloads.py:
timestep_supplement; bench.py reports the same measurement as config.supplements.timestep_216).
   DIMS=216,216,216 STEPS=5 [MI_TIMESTEP_SEGREGATED=1] python tools/bench_timestep.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft
import workloads
graft.build()
pkg = graft.load_package()
dims = [int(v) for v in os.environ.get("DIMS", "216,216,216").split(",")]
case = pkg.synthetic.box_case(*dims)
dev = torch.device("cuda:0")
ctx = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
addr = pkg.engine.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
print(json.dumps(workloads.timestep_supplement(pkg.engine, pkg.synthetic, case, addr, ctx, dev, steps=int(os.environ.get("STEPS", "5"))), indent=1))
