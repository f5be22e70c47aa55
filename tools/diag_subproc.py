"""diagnostic: which libamdhip64 copies are mapped when the engine creates its context"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
import __graft_entry__ as g
def maps(tag):
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l or "librccl" in l})
    print(tag, libs, flush=True)
if mode == "build_first":
    g.build()
pkg = g.load_package()
maps("after load_package")
if mode == "oracle_first":
    from oracle import oracle
    oracle.lib()
    maps("after oracle")
import torch
maps("after import torch")
s = torch.cuda.current_stream().cuda_stream
maps("after torch stream")
try:
    ctx = pkg.engine.Context(0, s)
    print("ctx ok")
except Exception as e:
    print("ctx FAIL", e)
maps("end")
