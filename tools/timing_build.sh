#!/bin/bash
# timing build of the engine (MI_TIMING) on the GPU box: where the layout / hierarchy seconds go
# (build/libtiming.so built in the container travels with the snapshot: `tools/timing_build.sh --build-only` there first saves
#  the two minutes of hipcc on the box)
set -e
LIB=build/libtiming.so
if [ ! -f $LIB ] || [ "$1" = "--build-only" ]; then
  mkdir -p build
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread -DMI_TIMING rapidcfd-dev_amd/csrc/engine.hip rapidcfd-dev_amd/csrc/tiling.cpp rapidcfd-dev_amd/csrc/gamg.cpp -o $LIB
fi
[ "$1" = "--build-only" ] && exit 0
MI_ENGINE_LIB=$PWD/$LIB python - 2> /tmp/timing.err <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); eng, syn = pkg.engine, pkg.synthetic
import workloads
case = syn.box_case(216, 216, 216)
ctx = eng.Context(0, torch.cuda.current_stream().cuda_stream)
w = workloads.box_pair_weights(case)      # the caller's input, generated outside the timed calls
for rep in range(2):
    print(f"== rep {rep}", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); addr = eng.Addressing(ctx, case.n_cells, case.lower_addr, case.upper_addr); t1 = time.perf_counter()
    print(f"== layout done {t1-t0:.3f}", file=sys.stderr, flush=True)
    G = eng.Gamg(addr, w, 100); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"== hierarchy done {t2-t1:.3f}", file=sys.stderr, flush=True)
    del G, addr
PY
mkdir -p gpurun_out && cp /tmp/timing.err gpurun_out/timing_build_${TAG:-latest}.err
python - <<'PY'
import collections, re
lines = open("/tmp/timing.err").read().splitlines()
i = max(k for k, l in enumerate(lines) if l.startswith("== rep"))
agg = collections.OrderedDict(); build = 0
for l in lines[i:]:
    if l.startswith("[gamg]   level"):
        if int(l.split()[2]) < 3: print(l)
        continue
    if l.startswith("==") or l.startswith("[gamg]") or l.startswith("[addr]"): print(l); continue
    m = re.match(r"\[gamg-host\] level +(-?\d+) (.*?) +([0-9.]+) s", l)
    if m:
        if int(m.group(1)) < 3: print(l)
        continue
    m = re.match(r"\[tiling\]\s+(.*?)\s+([0-9.]+) s", l)
    if not m: continue
    key = (build, m.group(1).strip()); agg[key] = agg.get(key, 0.0) + float(m.group(2))
    if m.group(1).strip() == "slots / halos / entries": build += 1
for b in range(build):
    tot = {k[1]: v for k, v in agg.items() if k[0] == b}
    top = sum(v for k, v in tot.items() if k in ("face lists", "clustering", "renumbering", "slots / halos / entries", "Cuthill-McKee"))
    if top > 0.02: print(f"layout build {b}: total {top:.3f} s :", {k: round(v, 3) for k, v in tot.items()})
PY
