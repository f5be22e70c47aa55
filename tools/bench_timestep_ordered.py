"""The PISO-like time step of tools/bench_timestep.py on the mesh RENUMBERED into the engine's tile order (what renumberMesh with the engine's
cell map gives: mi_addr_create_ordered -- operators and assembly passes run on the caller's arrays, blocks = tiles), beside the default
(permuting) addressing of the caller's numbering.   DIMS=216,216,216 STEPS=4 python tools/bench_timestep_ordered.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as graft
import workloads
graft.build()
pkg = graft.load_package()
eng, syn = pkg.engine, pkg.synthetic
dims = [int(v) for v in os.environ.get("DIMS", "216,216,216").split(",")]
steps = int(os.environ.get("STEPS", "4"))
case = syn.box_case(*dims)
dev = torch.device("cuda:0")
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
a0 = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr)
out = {}
r = workloads.timestep_supplement(eng, syn, case, a0, ctx, dev, steps=steps)
out["caller numbering (default addressing)"] = {"ms_per_time_step": r["ms_per_time_step"], "gamg_cycles": r.get("gamg_cycles"), "stages_ms": r["stages_ms"]}
rc = syn.renumber(case, a0.cell_perm())
rc.dims = case.dims
a1 = eng.Addressing(ctx, rc.n_cells, rc.lower_addr, rc.upper_addr, ordered=True, tile_cell_start=a0.tile_starts())
r = workloads.timestep_supplement(eng, syn, rc, a1, ctx, dev, steps=steps)
out["mesh renumbered into the tile order (ordered addressing)"] = {"ms_per_time_step": r["ms_per_time_step"], "gamg_cycles": r.get("gamg_cycles"), "stages_ms": r["stages_ms"]}
print(json.dumps(out, indent=1))
