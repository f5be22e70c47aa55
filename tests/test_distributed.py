"""CPU, world_size 2 and 4 over gloo: the N>1 path of rapidcfd-dev_amd/parallel.py (halo exchange
into the ext region, merged all-reduces, device-side convergence protocol) against the serial oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rank_pool import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_body(rank, world, parts, dims, kw, out_dir):
    """one rank of the CPU test (tests/rank_pool.py has joined the gloo group)"""
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from importlib import import_module
    par = import_module(graft.PKG_NAME + ".parallel")
    from oracle import oracle as orc
    from np_ops import NumpyOps
    case = pkg.synthetic.box_case(*dims)
    sub = pkg.synthetic.decompose_box(case, parts)[rank]
    solver = par.DistributedPCG(None, sub, "cpu", ops=NumpyOps(orc, sub))
    perf = solver.solve(**kw)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), psi=solver.ops.solution(), cells=sub.global_cells,
             hist=perf["history"], nit=perf["nIterations"], conv=perf["converged"], n_global=solver.n_global)
    dist.barrier()


@pytest.mark.parametrize("parts,kw", [((1, 1, 2), dict(tolerance=1e-8, max_iter=300)),
                                      ((2, 2, 1), dict(tolerance=1e-8, max_iter=300)),
                                      ((1, 2, 1), dict(tolerance=0.0, max_iter=7, batch=4)),
                                      ((2, 1, 1), dict(tolerance=1e30, max_iter=50, min_iter=3, batch=2))])
def test_distributed_pcg_matches_serial_oracle(pkg, orc, tmp_path, parts, kw):
    dims = (10, 8, 6)
    world = parts[0] * parts[1] * parts[2]
    run_ranks(world, "test_distributed", "_cpu_body", parts, dims, kw, str(tmp_path))
    case = pkg.synthetic.box_case(*dims)
    okw = dict(tolerance=kw["tolerance"], maxIter=kw["max_iter"], minIter=kw.get("min_iter", 0))
    ref_psi, ref = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", **okw)
    psi = np.zeros(case.n_cells)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        psi[d["cells"]] = d["psi"]
        assert int(d["n_global"]) == case.n_cells
        assert int(d["nit"]) == ref["nIterations"] and int(d["conv"]) == ref["converged"]
        h = d["hist"]
        assert h.shape == ref["history"].shape
        assert np.max(np.abs(h - ref["history"])) < 1e-10 * ref["history"][0]
    assert np.max(np.abs(psi - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


def _sum_ranks_body(rank, world, out_dir):
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    np.save(os.path.join(out_dir, f"s{rank}.npy"), t.numpy())


def test_rank_pool_does_not_depend_on_a_free_rendezvous_port(tmp_path, monkeypatch):
    """tests/rank_pool.py: a rendezvous port that is found free and bound by rank 0 seconds later can be taken in between on a busy box
    (round 5: EADDRINUSE in 2 of 58 pools of one GPU-suite run).  The ranks therefore meet through a FileStore and are handed NO port at
    all (round 6, ADVICE r05: a job body that wants a TCPStore picks its own); a pool comes up and runs its job with MASTER_PORT of the
    parent pointing at a port that is taken, and leaves no store file behind."""
    import glob, tempfile
    import rank_pool
    busy = socket.socket()
    busy.bind(("127.0.0.1", 0)); busy.listen(1)
    monkeypatch.setenv("MASTER_PORT", str(busy.getsockname()[1]))
    before = set(glob.glob(os.path.join(tempfile.gettempdir(), "mi_rank_pool_*.store")))
    try:
        rank_pool.run_ranks(2, "test_distributed", "_sum_ranks_body", str(tmp_path))
    finally:
        busy.close()
    assert all(float(np.load(tmp_path / f"s{r}.npy")[0]) == 3.0 for r in range(2))
    assert set(glob.glob(os.path.join(tempfile.gettempdir(), "mi_rank_pool_*.store"))) <= before


def test_direct_subdomain_equals_decomposed_global_case(pkg):
    # bench.py builds every rank's sub-domain directly (an 8 x 216^3 run never forms the 80 M-cell global case):
    # same addressing, coefficients, source, interface order and pairing as decompose_box(box_case(...))
    syn = pkg.synthetic
    for symmetric in (True, False):     # pressure-like and momentum-like (config 5's PBiCG runs on the latter)
        for dims, parts in (((9, 7, 5), (2, 2, 2)), ((8, 6, 5), (1, 2, 2)), ((7, 5, 6), (1, 1, 2)), ((10, 9, 4), (3, 2, 1))):
            subs = syn.decompose_box(syn.box_case(*dims, symmetric=symmetric), parts)
            for r, ref in enumerate(subs):
                got = syn.box_subdomain(dims, parts, r, symmetric=symmetric)
                assert got.n_cells == ref.n_cells and got.dims == ref.dims
                for name in ("lower_addr", "upper_addr", "upper", "source", "global_cells") + (() if symmetric else ("lower",)):
                    assert np.array_equal(getattr(got, name), getattr(ref, name)), (dims, parts, r, name)
                assert (got.lower is None) == symmetric
                assert np.max(np.abs(got.diag - ref.diag)) < 1e-12 * np.max(np.abs(ref.diag))
                assert len(got.interfaces) == len(ref.interfaces)
                for a, b in zip(got.interfaces, ref.interfaces):
                    assert (a.nbr_domain, a.nbr_patch) == (b.nbr_domain, b.nbr_patch)
                    assert np.array_equal(a.face_cells, b.face_cells) and np.array_equal(a.bou_coeffs, b.bou_coeffs)
                    assert np.array_equal(a.int_coeffs, b.int_coeffs)


# ---- the same N>1 path with the REAL engine: several ranks share the one GPU of the box, torch.distributed over gloo ----
def _gpu_body(rank, world, parts, dims, kw, out_dir, driver):
    os.environ["MI_DPCG_DRIVER"] = driver
    kw = dict(kw)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from importlib import import_module
    par = import_module(graft.PKG_NAME + ".parallel")
    torch.cuda.set_device(0)
    ctx = pkg.engine.Context(0, torch.cuda.current_stream().cuda_stream)
    if parts == "graph":   # ragged graph cut by a random cell-to-processor map: many small patches, several neighbours per rank
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import random_graph_case
        case = random_graph_case(pkg, dims[0], extra=2.0, seed=11)
        dom = (pkg.synthetic.splitmix_uniform(5, case.n_cells) * world).astype(np.int64)
        sub = pkg.synthetic.decompose(case, dom, world)[rank]
    else:
        sub = pkg.synthetic.box_subdomain(dims, parts, rank)
    solver = par.DistributedPCG(ctx, sub, "cuda:0", precond=kw.pop("precond", "diagonal"))
    perf = solver.solve(**kw)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), psi=solver.ops.solution(), cells=sub.global_cells, hist=perf["history"],
             nit=perf["nIterations"], conv=perf["converged"], n_global=solver.n_global, driver=solver.driver)
    dist.barrier()


@pytest.mark.gpu
@pytest.mark.parametrize("parts,driver,kw", [((1, 1, 2), "torch", dict(tolerance=1e-8, max_iter=400)),
                                             ((2, 2, 2), "torch", dict(tolerance=1e-8, max_iter=400)),
                                             ((1, 2, 2), "torch", dict(tolerance=0.0, max_iter=9, batch=4)),
                                             ((2, 1, 1), "native", dict(tolerance=1e-8, max_iter=400))])
def test_distributed_pcg_real_engine_ranks_share_one_gpu(pkg, orc, tmp_path, parts, driver, kw):
    """2, 4 and 8 ranks, each with its own engine context on the SAME GPU (RCCL refuses duplicate devices, gloo does not):
    tiled sub-domain matrices, halo exchange into the ext region, merged all-reduces and the device-side convergence
    protocol with real inter-rank traffic, against the serial oracle.  driver "native": the C++ RCCL loop cannot initialise
    here (duplicate GPU), so all ranks must agree to fall back to the torch.distributed loop."""
    dims = (20, 16, 12)
    world = parts[0] * parts[1] * parts[2]
    run_ranks(world, "test_distributed", "_gpu_body", parts, dims, dict(kw), str(tmp_path), driver)
    case = pkg.synthetic.box_case(*dims)
    okw = dict(tolerance=kw["tolerance"], maxIter=kw["max_iter"], minIter=kw.get("min_iter", 0))
    ref_psi, ref = orc.System([case]).pcg(np.zeros(case.n_cells), case.source, "diagonal", **okw)
    psi = np.zeros(case.n_cells)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        psi[d["cells"]] = d["psi"]
        assert str(d["driver"]) == "torch"
        assert int(d["n_global"]) == case.n_cells
        assert int(d["nit"]) == ref["nIterations"] and int(d["conv"]) == ref["converged"]
        assert np.max(np.abs(d["hist"] - ref["history"])) < 1e-10 * ref["history"][0]
    assert np.max(np.abs(psi - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


@pytest.mark.gpu
def test_distributed_pcg_real_engine_ragged_graph_random_partition(pkg, orc, tmp_path):
    """3 ranks on one GPU over gloo, an unstructured (ragged-graph) matrix cut by a RANDOM cell-to-processor map: every rank
    has patches to both others, faces scattered all over its cells."""
    from conftest import random_graph_case
    n, world = 3000, 3
    kw = dict(tolerance=1e-9, max_iter=400)
    run_ranks(world, "test_distributed", "_gpu_body", "graph", (n,), dict(kw), str(tmp_path), "torch")
    case = random_graph_case(pkg, n, extra=2.0, seed=11)
    ref_psi, ref = orc.System([case]).pcg(np.zeros(n), case.source, "diagonal", tolerance=1e-9, maxIter=400)
    psi = np.zeros(n)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        psi[d["cells"]] = d["psi"]
        assert int(d["nit"]) == ref["nIterations"] and int(d["conv"]) == ref["converged"]
        assert np.max(np.abs(d["hist"] - ref["history"])) < 1e-10 * ref["history"][0]
    assert np.max(np.abs(psi - ref_psi)) < 1e-9 * np.max(np.abs(ref_psi))


# ---- the engine's OWN (C++) loops with several ranks: communicators over the caller's transport (mi_comm_create_external) ----
def _native_rccl_worker(rank, world, port, spec, out_dir):
    """RCCL proper needs its own group (backend nccl, one device per rank): fresh processes, not the pooled gloo ranks"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    _native_body(rank, world, spec, out_dir, True)
    dist.destroy_process_group()


def _native_body(rank, world, spec, out_dir, rccl=False, peer=False):
    persist = isinstance(peer, str) and peer.startswith("persist")
    if peer == "mixed":
        # even ranks run the one-launch operators (boundary tiles poll their pairs), odd ranks the four-launch form (k_halo_push /
        # k_halo_pull behind flags) -- what a case with cyclicAMI / transformed patches on SOME ranks looks like
        os.environ["MI_WIN_DIRECT"] = "2" if rank % 2 == 0 else "0"
        peer = "auto"
    elif peer == "auto":
        # boundary tiles read the halo window themselves, waiting for the OTHER process's push inside the tile kernel -- the form
        # one-rank-per-device runs take; between processes that share a device it is opt-in (small cases only: many waiting
        # workgroups of one process could keep the other's kernel off the device)
        os.environ["MI_WIN_DIRECT"] = "2"
    if persist:
        # the persistent kernel between processes that SHARE the GPU: every rank's cooperative grid gets a share of the CUs
        # (all grids must be resident together, or they wait for each other until the polls run out -- bounded, and shortened here)
        os.environ["MI_PERSIST_SHARED"] = "1"
        os.environ["MI_PERSIST_GRID"] = peer.split(":")[1]
        os.environ["MI_PEER_POLLS"] = "3000000"
        peer = "auto"
    else:
        os.environ["MI_PCG_PERSIST"] = "0"
    # RCCL: one device per rank; gloo transport: the ranks share device 0 -- unless the node has a device for every rank and
    # MI_TEST_DEVICE_PER_RANK=1 asks for it (tools/first_lease.sh: the windows then cross xGMI instead of staying in one HBM)
    per_rank = os.environ.get("MI_TEST_DEVICE_PER_RANK") == "1" and torch.cuda.device_count() >= world
    d = rank if (rccl or per_rank) else 0
    torch.cuda.set_device(d)
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from importlib import import_module
    par = import_module(graft.PKG_NAME + ".parallel")
    ctx = pkg.engine.Context(d, torch.cuda.current_stream().cuda_stream)
    subs, weights = _native_case(pkg, spec, world)
    sub = subs[rank]
    dname = f"cuda:{d}"
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dname)
    comms = par.make_comms(ctx) if rccl else par.make_host_comms(ctx, peer=(peer == "auto"))
    if peer == "auto":
        assert comms[0].peer_mode, "mi_comm_peer_auto did not come up between processes sharing one GPU"
    elif peer:
        par.enable_peer_allreduce(comms[0])
    dm = par.DistributedMatrix(ctx, sub, dname, comms=comms)
    res = dict(cells=sub.global_cells, n_global=dm.n_global)
    x = pkg.synthetic.splitmix_uniform(3, dm.n_global)[sub.global_cells] - 0.5
    out = torch.empty(sub.n_cells, dtype=torch.float64, device=dname)
    dm.mat.amul(dev(x), out); torch.cuda.synchronize(); res["amul"] = out.cpu().numpy()
    import time as _time
    timings = []
    for name, solver, kw in spec["solves"]:
        kw = dict(kw)
        _t0 = _time.perf_counter()
        if solver == "GAMG":
            kw["face_weights"] = weights[rank]
        if solver == "PBiCG3":   # the components of a vector equation in ONE solve on the attached matrix (mi_pbicg_solve_multi)
            psis = [torch.zeros(sub.n_cells, dtype=torch.float64, device=dname) for _ in PBICG3_COMPONENTS]
            perfs = dm.mat.pbicg_multi(psis, [dev(a * sub.source + b) for a, b in PBICG3_COMPONENTS], **kw)
            torch.cuda.synchronize()
            for c, (q, pf) in enumerate(zip(psis, perfs)):
                res[f"{name}{c}_psi"] = q.cpu().numpy(); res[f"{name}{c}_hist"] = pf["history"]; res[f"{name}{c}_nit"] = pf["nIterations"]
            timings.append((name, solver, _time.perf_counter() - _t0, sum(pf["nIterations"] for pf in perfs)))
            continue
        psi = torch.zeros(sub.n_cells, dtype=torch.float64, device=dname)
        perf = dm.solve(solver, psi, dev(sub.source), **kw)
        torch.cuda.synchronize()
        timings.append((name, solver, _time.perf_counter() - _t0, perf["nIterations"]))
        res[name + "_psi"] = psi.cpu().numpy(); res[name + "_hist"] = perf["history"]; res[name + "_nit"] = perf["nIterations"]
    if any(s[1] == "GAMG" for s in spec["solves"]):
        res["gamg_levels"] = dm._gamg.n_levels
    assert rccl or not dm.comms[0].errors, dm.comms[0].errors
    if peer:
        st, fine = comms[0].peer_status()
        assert st == 0, "a peer all-reduce ran out of polls"
        res["peer_fine_grained"] = fine
        assert dm.mat.peer_halo_status() == (True, 0), "the halo of the attached matrix did not go through windows"
    if persist:
        assert ctx.stat(1) > 0, "the persistent distributed kernel did not run"
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), **res)
    dist.barrier()
    if rank == 0:     # where a multi-process test spends its time, solve by solve (gpurun_out/native_solve_timings.tsv)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "native_solve_timings.tsv"), "a") as f:
                for name, solver, sec, nit in timings:
                    f.write(f"{sec:8.3f}\t{nit}\t{solver}\t{name}\t{spec['kind']}\tworld={world}\tpeer={peer}\tpersist={persist}\trccl={rccl}\n")
        except OSError:
            pass
    # everything this job created goes before the next one runs in this process (tests/rank_pool.py): the engine's handles have no
    # finalisers, a rank process that serves many jobs would otherwise keep every context, window and hierarchy of the jobs before
    if dm._gamg is not None:
        dm._gamg.close()
    dm.mat.close(); dm.addr.close()
    for c in {id(c): c for c in comms}.values():
        c.close()
    ctx.close()


PBICG3_COMPONENTS = ((1.0, 0.0), (0.5, 0.01), (-0.3, 0.0))   # source of component c = a * source + b


def _native_case(pkg, spec, world):
    syn = pkg.synthetic
    from oracle import oracle as orc
    if spec["kind"] == "graph":
        from conftest import random_graph_case
        case = random_graph_case(pkg, spec["n"], extra=2.0, seed=11, symmetric=spec["symmetric"])
        dom = (syn.splitmix_uniform(5, case.n_cells) * world).astype(np.int64)
        subs = syn.decompose(case, dom, world)
        weights = [0.5 + syn.splitmix_uniform(40 + d, s.n_faces) for d, s in enumerate(subs)]
    elif spec["kind"] == "ami_y":
        # the non-conformal y-interface of tests/test_ami.py with its two sides on DIFFERENT ranks (slabs in y)
        subs = syn.decompose_cyclic_ami_y(syn.box_case(*spec["dims"], symmetric=spec["symmetric"]), world, **spec.get("ami", {}))
        weights = [orc.box_face_weights(s) for s in subs]
    elif spec["kind"] == "ami_split":
        # ... with BOTH sides split over world / 2 ranks each, every piece overlapping several partner pieces (px x 2 blocks)
        subs = syn.decompose_cyclic_ami_split(syn.box_case(*spec["dims"], symmetric=spec["symmetric"]), world // 2, **spec.get("ami", {}))
        weights = [orc.box_face_weights(s) for s in subs]
    else:
        case = syn.box_case(*spec["dims"], symmetric=spec["symmetric"])
        subs = syn.decompose_box(case, spec["parts"])
        weights = [orc.box_face_weights(s) for s in subs]
        if spec.get("transform"):
            # processorCyclic patches: the received neighbour values are multiplied by the patch's transformCoupleField factor
            # (processorCyclicGAMGInterfaceField.C:1-79, processorGAMGInterfaceField.C:213,230; cyclicLduInterfaceField.C:45-62)
            for d, s in enumerate(subs):
                for p, itf in enumerate(s.interfaces):
                    itf.transform = spec["transform"][(d + p) % len(spec["transform"])]
    return subs, weights


NATIVE_SPECS = {
    "box_2": dict(kind="box", dims=(20, 16, 12), parts=(2, 1, 1), symmetric=True,
                  solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=1e-9, maxIter=400)), ("dic", "PCG", dict(precond="AINV", tolerance=1e-9, maxIter=400)),
                          ("smooth", "smoothSolver", dict(n_sweeps=2, tolerance=1e-4, maxIter=300)), ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=60))]),
    "box_4_asym": dict(kind="box", dims=(16, 14, 12), parts=(2, 2, 1), symmetric=False,
                       solves=[("bicg", "PBiCG", dict(precond="AINV", tolerance=1e-10, maxIter=300)), ("stab", "PBiCGStab", dict(precond="diagonal", tolerance=1e-10, maxIter=300)),
                               ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=60)),
                               ("bicg3_", "PBiCG3", dict(precond="AINV", tolerance=1e-10, maxIter=300))]),   # round 4: batched momentum solve on attached matrices
    # a box cut in 2 x 2 whose processor patches all carry a transformation factor != 1 (the transformed RECEIVE path: VERDICT r02
    # "missing" 5); the factors make the global operator non-symmetric, so the bi-conjugate solvers run on it
    "box_4_transformed": dict(kind="box", dims=(16, 14, 12), parts=(2, 2, 1), symmetric=False, transform=(0.8, -0.6, 0.9),
                              solves=[("bicg", "PBiCG", dict(precond="AINV", tolerance=1e-7, maxIter=300)),   # (it stalls near 1e-8 on this operator)
                                      ("stab", "PBiCGStab", dict(precond="diagonal", tolerance=1e-10, maxIter=300)),
                                      ("smooth", "smoothSolver", dict(n_sweeps=2, tolerance=1e-4, maxIter=300))]),
    # sub-domains for the persistent distributed kernel: two tiles per workgroup at 2 ranks x 96 workgroups, one at 4 x 48
    # (short solves: the processes' cooperative grids share one GPU and every exchange waits for the other process's kernel --
    #  a scheduling quantum per barrier; one solve of each spec ends by the convergence test INSIDE the kernel)
    "box_2_persist": dict(kind="box", dims=(96, 64, 48), parts=(2, 1, 1), symmetric=True,
                          solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=0.0, maxIter=24)), ("pcg0", "PCG", dict(precond="none", tolerance=0.15, maxIter=200))]),
    # ... and five tiles per workgroup (what the 8-GPU share of the benchmark needs) at 2 ranks x 96 workgroups
    "box_2_persist5": dict(kind="box", dims=(128, 96, 80), parts=(1, 2, 1), symmetric=True,
                           solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=0.0, maxIter=16))]),
    # the REAL 8-GPU topology: 2 x 2 x 2, three processor patches per rank, 108 tiles per rank on 24 workgroups = five tiles per
    # workgroup -> k_pcg_persist<5, true> (VERDICT r03 "weak" 1: that topology had never been through the DIST kernel)
    "box_8_persist5": dict(kind="box", dims=(96, 96, 96), parts=(2, 2, 2), symmetric=True,
                           solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=0.0, maxIter=16)), ("pcg0", "PCG", dict(precond="none", tolerance=0.15, maxIter=200))]),
    "box_4_persist": dict(kind="box", dims=(64, 64, 48), parts=(2, 2, 1), symmetric=True,
                          solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=0.15, maxIter=200)), ("pcg0", "PCG", dict(precond="none", tolerance=0.0, maxIter=16))]),
    # row f3: cyclicAMI whose halves live on different ranks (AMIInterpolation.C:940-1091): y-slabs, slab 0 holds the y-min side,
    # the last slab the refined and shifted y-max side (different face counts), processor patches between the slabs
    "ami_sym": dict(kind="ami_y", dims=(20, 16, 12), symmetric=True, ami=dict(shift=0.37),
                    solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=1e-9, maxIter=500)), ("dic", "PCG", dict(precond="AINV", tolerance=1e-9, maxIter=500)),
                            ("smooth", "smoothSolver", dict(n_sweeps=2, tolerance=1e-4, maxIter=300)),
                            ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=60, directSolveCoarsest=False))]),
    "ami_asym": dict(kind="ami_y", dims=(20, 16, 12), symmetric=False, ami=dict(shift=0.37, low_weight_every=7, transform=0.6),
                     solves=[("bicg", "PBiCG", dict(precond="AINV", tolerance=1e-10, maxIter=300)), ("stab", "PBiCGStab", dict(precond="diagonal", tolerance=1e-10, maxIter=300)),
                             ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=60, directSolveCoarsest=False))]),
    # row f3, round 6: both SIDES of the cyclicAMI pair split over several ranks (px x 2 blocks; the shifted, x-periodic y-max side makes
    # every piece overlap two partner pieces): one transport patch per partner piece (mi_addr_set_ami_patch_remote_multi)
    "ami_split_sym": dict(kind="ami_split", dims=(20, 16, 12), symmetric=True, ami=dict(shift=0.37),
                          solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=1e-9, maxIter=500)), ("dic", "PCG", dict(precond="AINV", tolerance=1e-9, maxIter=500)),
                                  ("smooth", "smoothSolver", dict(n_sweeps=2, tolerance=1e-4, maxIter=300)),
                                  ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=60, directSolveCoarsest=False))]),
    "ami_split_asym": dict(kind="ami_split", dims=(20, 16, 12), symmetric=False, ami=dict(shift=1.61, low_weight_every=7, transform=0.6),
                           solves=[("bicg", "PBiCG", dict(precond="AINV", tolerance=1e-10, maxIter=300)), ("stab", "PBiCGStab", dict(precond="diagonal", tolerance=1e-10, maxIter=300)),
                                   ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=60, directSolveCoarsest=False))]),
    "graph_3": dict(kind="graph", n=3000, symmetric=True,
                    solves=[("pcg", "PCG", dict(precond="diagonal", tolerance=1e-9, maxIter=400)), ("gamg", "GAMG", dict(tolerance=1e-9, maxIter=80))]),
}


# Ranks that SHARE a GPU and wait for each other inside kernels (window flags / pairs polled by a tile kernel, the one-shot window
# all-reduce, the persistent grids) pay one hardware scheduling quantum per exchange: 16 ms per all-reduce measured between two
# processes on one MI355X (profiles/r05_c_native_solve_timings.tsv: 144 PCG iterations 4.6 s over windows, 0.08 s over gloo; the
# same to three digits with fresh and with re-used processes).  One rank per device never sees that -- but these tests do, so the
# window variants run every solver for a FIXED, short number of iterations (histories compared entry by entry as always); the
# runs to convergence are the gloo-transport variants' (same C++ loops, same collectives).
_WINDOW_ITERATIONS = {"PCG": 24, "PBiCG": 12, "PBiCGStab": 10, "smoothSolver": 8, "GAMG": 3, "PBiCG3": 8}


def _windows_spec(spec):
    # With ONE DEVICE PER RANK (MI_TEST_DEVICE_PER_RANK=1: tools/first_lease.sh on a multi-GPU node) nothing shares a scheduler, an
    # exchange costs microseconds, and the window variants run the SAME solves as the plain transport -- to their tight tolerances,
    # hundreds of iterations: late-iteration epoch / parity / flag-reuse errors (round 4's cyclicAMI hang was of that kind) would show
    # there.  On the shared GPU of the default suite they stay short (below).
    if os.environ.get("MI_TEST_DEVICE_PER_RANK") == "1":
        return dict(spec)
    out = dict(spec)
    # (GAMG keeps a finite tolerance: with directSolveCoarsest = False its coarsest-level ICCG / BICCG inherits it,
    #  GAMGSolverSolve.C:572-613 -- tolerance 0 there means 1000 coarsest iterations per cycle)
    out["solves"] = [(name, solver, dict(kw, tolerance=1e-3 if solver == "GAMG" else 0.0, maxIter=_WINDOW_ITERATIONS[solver])) for name, solver, kw in spec["solves"]]
    # ... except that the first PCG still ENDS BY ITS TOLERANCE, in the middle of a batch of enqueued iterations: the exchanges of the
    # iterations behind the converged one must be counted alike by ranks in different window forms (round 4's 4-rank cyclicAMI hang)
    for k, (name, solver, kw) in enumerate(out["solves"]):
        if solver == "PCG":
            out["solves"][k] = (name, solver, dict(kw, tolerance=0.05, maxIter=200))
            break
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("peer", [False, "auto"])
def test_transformed_processor_patches_between_engine_ranks(pkg, orc, tmp_path, peer):
    """processorCyclic: every processor patch of a 2 x 2 decomposition multiplies the values it RECEIVES by its own
    transformCoupleField factor (scale_received after the send/recv exchange; inside k_halo_pull when the halo travels through
    peer windows).  Amul bit-exact across the cuts, PBiCG + DILU / PBiCGStab / smoothSolver histories against the multi-domain
    oracle whose interfaces carry the same factors."""
    spec = NATIVE_SPECS["box_4_transformed"] if not peer else _windows_spec(NATIVE_SPECS["box_4_transformed"])
    run_ranks(4, "test_distributed", "_native_body", spec, str(tmp_path), False, peer)
    _check_native(pkg, orc, spec, 4, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("box_2", 2), ("box_4_asym", 4), ("graph_3", 3)])
def test_native_attached_solvers_on_several_engine_ranks(pkg, orc, tmp_path, name, world):
    """The engine's own C++ loops -- device-resident distributed PCG (mi_dpcg_comm_iterate), the attached host-stepped
    solvers, and GAMG with processor interfaces on every level and the GLOBAL coarsest system assembled from per-rank block
    rows at distinct offsets (row a18; LUscalarMatrix.C:57-150) -- with 2, 3 and 4 REAL ranks.  RCCL refuses ranks that share a
    GPU, so the communicators run over the external-transport hook (mi_comm_create_external) fed by gloo: every collective the
    RCCL build issues is issued here too, by the same C++ code, between distinct ranks with distinct sub-domains.  Reference:
    the multi-domain oracle on the same decomposition."""
    spec = NATIVE_SPECS[name]
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path))
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("box_2", 2), ("box_4_asym", 4)])
def test_native_attached_solvers_over_rccl_one_device_per_rank(pkg, orc, tmp_path, name, world):
    """the same loops over RCCL proper (mi_comm_create on ncclUniqueIds, one device per rank, xGMI between them): runs where
    the box has that many GPUs -- the first multi-GPU lease executes it -- and skips on a single-GPU box"""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, {torch.cuda.device_count()} visible")
    spec = NATIVE_SPECS[name]
    mp.spawn(_native_rccl_worker, args=(world, _free_port(), spec, str(tmp_path)), nprocs=world, join=True)
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("box_2", 2), ("box_4_asym", 4)])
def test_native_solvers_with_the_one_shot_peer_allreduce(pkg, orc, tmp_path, name, world):
    """the same loops with every small all-reduce (the scalars of PCG / PBiCG / PBiCGStab, GAMG's scale factors and stop flag)
    going through the peer windows (mi_comm_peer_window / mi_comm_peer_connect): each rank writes its values + an epoch flag
    into every rank's window and adds the nRanks contributions in rank order.  Here the ranks are processes that share the
    GPU and map each other's windows over hipIpc; halo exchange and the large all-reduces stay on the gloo transport."""
    spec = _windows_spec(NATIVE_SPECS[name])
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path), False, True)
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("box_2", 2), ("graph_3", 3), ("box_4_asym", 4)])
def test_native_solvers_entirely_over_peer_windows(pkg, orc, tmp_path, name, world):
    """Round 3: mi_comm_peer_auto (windows, handle exchange over the communicator's own transport, coherence self-test, agreement)
    and, on top of it, the HALO of every attached matrix -- GAMG level matrices included -- through windows (k_halo_push /
    k_halo_pull) and the fused three-launch distributed PCG iteration, between 2, 3 and 4 processes that map each other's
    windows over hipIpc.  Only what does not fit the windows (all-reduces > 8 doubles, the hierarchy build) still uses gloo."""
    spec = _windows_spec(NATIVE_SPECS[name])
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path), False, "auto")
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("box_2", 2), ("box_4_asym", 4)])
def test_neighbours_in_different_window_forms(pkg, orc, tmp_path, name, world):
    """Which form a rank READS its halo in is its own choice (every push writes pairs AND values behind flags): ranks in the
    one-launch form next to ranks in the four-launch form must count the same exchanges -- also in the iterations a batch
    enqueues after the solve has converged, which the one-launch operators used to skip entirely while k_halo_push /
    k_halo_pull went on (round 4: a 4-rank cyclicAMI case ran out of polls exactly there)."""
    spec = _windows_spec(NATIVE_SPECS[name])
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path), False, "mixed")
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("peer", [False, "auto"])
@pytest.mark.parametrize("name,world", [("ami_sym", 2), ("ami_asym", 2), ("ami_sym", 4)])
def test_cyclic_ami_whose_halves_live_on_different_ranks(pkg, orc, tmp_path, name, world, peer):
    """Row f3 (round 4): the non-conformal interface of tests/test_ami.py cut across ranks -- the y-min side on rank 0, the
    refined / shifted y-max side on the last rank (the reference's distributed AMI, singlePatchProc == -1).  The partner's
    patch-internal field travels through a TRANSPORT processor patch (padded to the larger side, zero coefficients:
    mi_addr_set_ami_patch_remote), the interpolation reads what that patch receives, and every GAMG level derives the partner
    side's coarse faces from the coarse cells the transport patch receives.  Amul bit-exact across the interface; PCG / DIC-PCG /
    smoothSolver / PBiCG / PBiCGStab / GAMG (ICCG / BICCG on the coarsest level) against the multi-domain oracle (1e-10), over
    the external transport and over peer windows, with low-weight faces and a transformation factor in the asymmetric case."""
    spec = _windows_spec(NATIVE_SPECS[name]) if peer else NATIVE_SPECS[name]
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path), False, peer)
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,peer", [("ami_split_sym", 4, False), ("ami_split_sym", 4, "auto"), ("ami_split_asym", 4, False), ("ami_split_asym", 4, "auto"),
                                             ("ami_split_sym", 6, "auto")])   # (6 ranks over the plain transport: 157 s on the shared GPU -- the callbacks' hipMemcpy per exchange; windows: 12 s)
def test_cyclic_ami_side_split_over_ranks(pkg, orc, tmp_path, name, world, peer):
    """Row f3 (round 6; VERDICT r05 "missing" 1): BOTH sides of the non-conformal interface split over several ranks (px x 2 blocks:
    ranks 0 .. px-1 hold the pieces of the y-min side, the others the pieces of the refined, shifted, x-periodic y-max side), every
    piece overlapping faces of two partner pieces -- the reference's calcProcMap case (AMIInterpolation.C:940-1091,
    AMIInterpolationParallelOps.C).  One transport patch per partner piece, addresses numbering the pieces' faces concatenated
    (mi_addr_set_ami_patch_remote_multi); a face's weighted sum takes terms from several ranks in address order; every GAMG level
    derives each piece's coarse faces from what its own transport patch receives.  Amul bit-exact against the multi-domain oracle
    (whose split-side path equals the single-domain AMI bit for bit on every row off the processor cuts: tests/test_ami.py); Krylov
    solvers and GAMG against it to 1e-10, over the external transport and over peer windows."""
    spec = _windows_spec(NATIVE_SPECS[name]) if peer else NATIVE_SPECS[name]
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path), False, peer)
    _check_native(pkg, orc, spec, world, str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name,world,grid", [("box_2_persist", 2, 96), ("box_2_persist5", 2, 96), ("box_4_persist", 4, 48), ("box_8_persist5", 8, 24)])
def test_persistent_distributed_pcg_between_processes(pkg, orc, tmp_path, name, world, grid):
    """csrc/persist.inc, DIST form, between REAL ranks: 2 and 4 processes (distinct sub-domains, each other's windows mapped over
    hipIpc) run the whole PCG iteration -- halo stores into the neighbours' windows, flags, both all-reduces through the
    replicated window slots, rank-order sums -- inside their persistent cooperative kernels, which share the one GPU of this
    box (MI_PERSIST_GRID workgroups each).  Iteration counts, histories (1e-10) and solutions against the multi-domain oracle."""
    spec = NATIVE_SPECS[name]
    run_ranks(world, "test_distributed", "_native_body", spec, str(tmp_path), False, f"persist:{grid}")
    _check_native(pkg, orc, spec, world, str(tmp_path))


def _check_native(pkg, orc, spec, world, tmp_path):
    subs, weights = _native_case(pkg, spec, world)
    S = orc.System(subs)
    n = sum(s.n_cells for s in subs)
    src = np.concatenate([s.source for s in subs])
    offs = np.concatenate([[0], np.cumsum([s.n_cells for s in subs])])
    data = [np.load(os.path.join(str(tmp_path), f"r{r}.npz")) for r in range(world)]
    xg = pkg.synthetic.splitmix_uniform(3, n)
    x = np.concatenate([xg[s.global_cells] - 0.5 for s in subs])
    assert np.array_equal(np.concatenate([d["amul"] for d in data]), S.amul(x))          # bit-exact across the cut
    H = None
    for sname, solver, kw in spec["solves"]:
        if solver == "PBiCG3":
            for c, (a, b) in enumerate(PBICG3_COMPONENTS):
                ref_psi, ref = S.pbicg(np.zeros(n), a * src + b, **kw)
                for r, d in enumerate(data):
                    assert int(d[f"{sname}{c}_nit"]) == ref["nIterations"], (sname, c, r, int(d[f"{sname}{c}_nit"]), ref["nIterations"])
                    h = d[f"{sname}{c}_hist"]
                    assert h.shape == ref["history"].shape and np.max(np.abs(h - ref["history"])) < 1e-10 * ref["history"][0], (sname, c, r)
                psi = np.concatenate([d[f"{sname}{c}_psi"] for d in data])
                assert np.max(np.abs(psi - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi)), (sname, c)
            continue
        if solver == "GAMG":
            H = orc.GamgSysHierarchy(S, weights, 10)
            ref_psi, ref = H.solve(np.zeros(n), src, **kw)
        else:
            fn = {"PCG": S.pcg, "PBiCG": S.pbicg, "PBiCGStab": S.pbicgstab, "smoothSolver": S.smooth_solve}[solver]
            ref_psi, ref = fn(np.zeros(n), src, **kw)
        for r, d in enumerate(data):
            assert int(d[sname + "_nit"]) == ref["nIterations"], (sname, r, int(d[sname + "_nit"]), ref["nIterations"])
            h = d[sname + "_hist"]
            assert h.shape == ref["history"].shape and np.max(np.abs(h - ref["history"])) < 1e-10 * ref["history"][0], (sname, r)
        psi = np.concatenate([d[sname + "_psi"] for d in data])
        assert np.max(np.abs(psi - ref_psi)) < 1e-8 * np.max(np.abs(ref_psi)), sname
    if H is not None:
        assert all(int(d["gamg_levels"]) == H.n_levels for d in data)
        # the ranks' coarsest blocks differ in size, so their block offsets in the global coarsest system are distinct
        sizes = [H.level(d, H.n_levels - 2)["n_coarse"] for d in range(world)]
        assert all(v > 0 for v in sizes) and len(set(np.cumsum(sizes).tolist())) == world
        if world > 2 or spec["kind"] == "graph":
            assert len(set(sizes)) > 1 or world > 2
