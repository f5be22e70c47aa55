"""One rhoPimpleFoam time step on the 216^3 box (BASELINE config 5's solver, per-rank share; tools/workloads.py: rhopimple_supplement).
   DIMS=216,216,216 STEPS=3 [TRANSONIC=1] python tools/bench_rhopimple.py          (under rocprofv3 --kernel-trace: tools/gpu_r06_h.sh)"""
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
print(json.dumps(workloads.rhopimple_supplement(pkg.engine, pkg.synthetic, case, addr, ctx, dev, steps=int(os.environ.get("STEPS", "3")),
                                                transonic=bool(int(os.environ.get("TRANSONIC", "0")))), indent=1))
