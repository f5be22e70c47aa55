"""GPU (-m gpu): bench.py's driver contract on a small box -- the single-GPU line, and a rehearsal of the N > 1 path with
several ranks sharing the one GPU over gloo (what the driver launches with torch.distributed.run on a multi-GPU node)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline", "cpu_baseline"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _check(line, n):
    d = json.loads(line)
    assert KEYS - ({"cpu_baseline"} if n > 1 else set()) <= set(d), set(d) ^ KEYS      # the CPU leg runs at N = 1 only
    assert d["n_gpus"] == n and d["steps"] == 12 and d["warmup"] == 3 and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    return d


def _run_with_port(make_cmd, env, attempts=4):
    """subprocess.run of a launcher command built for a free rendezvous port; the port is found free and bound a moment later by the
    launcher's store, so on a busy box another socket can take it in between: try again with another port"""
    for k in range(attempts):
        out = subprocess.run(make_cmd(_free_port()), capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        if out.returncode == 0 or "address already in use" not in out.stderr.lower() or k == attempts - 1:
            return out


def test_single_gpu_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "3", "--dims", "40", "32", "24", "--cpu-iters", "5"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                       # ONE JSON line on stdout, everything else on stderr
    d = _check(lines[0], 1)
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    # configs 3 and 4/5 ride in the same line as supplements (never in `value`)
    sup = d["config"]["supplements"]
    assert "error" not in sup, sup
    assert sup["gamg_216"]["v_cycles_per_s"] > 0 and 0 < sup["gamg_216"]["roofline_frac"] < 1 and sup["gamg_216"]["solve_to_1e-6"]["cycles"] > 0
    ts = sup["timestep_216"]
    assert ts["ms_per_time_step"] > 0 and len(ts["pbicg_iterations_per_component"]) == 3 and ts["gamg_cycles"] >= 1 and len(ts["stages_ms"]) == 6
    # round 5: config 5's own solver (rhoPimpleFoam: UEqn / EEqn / pEqn), non-transonic and transonic (asymmetric pressure matrix)
    rp = sup["rhopimple_timestep_216"]
    assert "error" not in rp, rp
    for form, kind in (("non_transonic", "symmetric"), ("transonic", "asymmetric")):
        q = rp[form]
        assert q["ms_per_time_step"] > 0 and q["gamg_cycles"] >= 1 and q["pressure_matrix"] == kind and len(q["stages_ms"]) == 7 and "rhoPimpleFoam" in q["workload"]


def test_gamg_mode_line():
    """`--solver gamg` (BASELINE config 3): one JSON line, V-cycles/s, a roofline over the whole cycle; a run long enough to
    reach the rounding floor of the residual (the default K does at full size) must not trip the bench's own checks"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--solver", "gamg", "--steps", "90", "--warmup", "3", "--dims", "40", "32", "24"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["unit"] == "V-cycles/s" and d["steps"] == 90 and d["n_gpus"] == 1 and d["value"] > 0
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    assert d["config"]["solver"] == "GAMG" and d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1


@pytest.mark.parametrize("n", [2, 4])
def test_multi_rank_rehearsal_over_gloo(n):
    env = dict(os.environ, MI_BENCH_BACKEND="gloo", MI_BENCH_DECOMP_CYCLES="3", MI_BENCH_DECOMP_STEPS="1")   # (ranks share ONE GPU here: short supplements)
    make = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "12", "--warmup", "3",
           "--dims", "40", "32", "24", "--no-cpu"]
    out = _run_with_port(make, env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 only
    d = _check(lines[0], n)
    assert d["scaling"] == "strong" and "domain-decomposition" in d["config"]["parallelism"]
    assert d["config"]["weak_scaling_supplement"]["cells_per_gpu"] == 40 * 32 * 24


def test_two_rank_rehearsal_with_peer_windows_and_the_persistent_kernel():
    """the N > 1 path of bench.py as an 8-GPU node takes it -- peer windows, the trial solve that compares the persistent kernel's
    history with the five-launch loop's, bare timed repeats, one more sampled batch for the Amul duration -- rehearsed with two
    ranks that share this box's GPU: communicators over the host transport, windows over hipIpc, the two cooperative grids side by
    side (MI_PERSIST_GRID workgroups each)"""
    env = dict(os.environ, MI_BENCH_BACKEND="gloo", MI_DPCG_DRIVER="native", MI_COMM_TRANSPORT="host", MI_PERSIST_SHARED="1", MI_PERSIST_GRID="96", MI_PEER_POLLS="3000000",
               MI_BENCH_DECOMP_CYCLES="3", MI_BENCH_DECOMP_STEPS="1")     # (two processes on one GPU: a scheduling quantum per exchange -- short supplements)
    make = lambda port: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--dims", "96", "64", "48", "--no-cpu"]
    out = _run_with_port(make, env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = _check(lines[0], 2)
    assert "one persistent cooperative kernel per batch" in d["config"]["host_loop"], d["config"]["host_loop"]
    assert d["config"]["allreduce"].startswith("peer windows"), d["config"]["allreduce"]
    # round 4: configs 3 / 4 / 5 as decomposed workloads ride in the N > 1 line too (GAMG with processor interfaces, a whole time
    # step with attached matrices), on the weak sub-domain and on the strong share
    sup = d["config"]["supplements"]
    assert sup and "error" not in sup, sup
    for cells in (96 * 64 * 48, 96 * 64 * 24):
        q = sup[f"decomposed_{cells}_cells_per_rank"]
        assert q["gamg"]["ms_per_v_cycle"] > 0 and q["gamg"]["halo_through_peer_windows"] and q["gamg"]["wait_timeouts"] == 0
        assert q["gamg"]["cycles_replayed_as_hipGraph"] > 0, q["gamg"]
        assert q["timestep"]["ms_per_time_step"] > 0 and q["timestep"]["momentum_solve"].startswith("one batched"), q["timestep"]


def test_gpus_flag_without_a_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no torch.distributed.run around it) must not fall back to one GPU silently: it re-executes
    itself under the launcher.  Rehearsed over gloo (two ranks share this box's GPU); over RCCL it refuses when fewer GPUs than
    ranks are visible."""
    env = dict(os.environ, MI_BENCH_BACKEND="gloo", MI_BENCH_DECOMP_CYCLES="3", MI_BENCH_DECOMP_STEPS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--dims", "40", "32", "24", "--no-cpu"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    d = _check(lines[0], 2)
    assert d["config"]["timing"].startswith("median of 5 repeats")
    import torch
    if torch.cuda.device_count() < 2:
        env.pop("MI_BENCH_BACKEND")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dims", "40", "32", "24", "--no-cpu"],
                             capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert out.returncode != 0 and "GPU(s) visible" in out.stderr

