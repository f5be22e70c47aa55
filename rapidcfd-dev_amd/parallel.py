"""One-rank-per-GPU domain-decomposed PCG.

The reference runs one MPI rank per GPU; every rank executes the same PCG and
meets the others in ``Foam::reduce`` (three scalar all-reduces per iteration,
PCG.C:142,166,195 -> src/Pstream/mpi/allReduceTemplates.C:195-208) and in the
processor-patch exchange inside every ``Amul``
(lduMatrixUpdateMatrixInterfaces.C:30-276, processorFvPatchScalarField.C:36-170,
host-staged MPI_Isend/Irecv by default).

MI355X-native form: the device-resident PCG pipeline of the engine is cut into
phases at exactly those points (``mi_dpcg_phase``); between phases this driver
issues RCCL collectives through ``torch.distributed`` on tensors that ARE the
engine's buffers (no staging copies):

* halo: the pack kernel writes the patch-internal values of ``pA`` into the send
  buffer; ``isend``/``irecv`` pairs (one per neighbour = one xGMI link each) run on
  a second HIP stream and land directly in ``pA[n_cells:]`` while the interior
  tiles of ``Amul`` run on the main stream; boundary tiles run after the wait.
* global sums: the ``sum|rA|`` of iteration k and the ``wA.rA`` of iteration k+1
  come out of the same pass over ``rA``, so they travel in ONE two-double
  all-reduce; with the ``wA.pA`` all-reduce that is two collectives per iteration
  instead of the reference's three.  The host never reads a scalar inside the loop:
  convergence is tested on the device and polled once per batch.

The arithmetic backend is pluggable (``ops``): the product backend is
:class:`HipOps` (HIP kernels through the C ABI).  Tests inject a numpy backend to
check the exchange/reduction logic with the ``gloo`` backend on CPU.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import engine as eng


class HipOps:
    """Per-rank state on the GPU: engine handles + the tensors the collectives act on."""

    def __init__(self, ctx: "eng.Context", sub, device, precond: str = "diagonal"):
        if sub.lower is not None:
            raise ValueError("PCG needs a symmetric matrix")
        self.device = device
        self.n = sub.n_cells
        self.addr = eng.Addressing(ctx, sub.n_cells, sub.lower_addr, sub.upper_addr,
                                   [itf.face_cells for itf in sub.interfaces])
        self.n_ext = self.addr.n_ext
        self.offsets = self.addr.patch_offsets()
        self.mat = eng.Matrix(self.addr)

        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(device)

        self.mat.set_coeffs(t(sub.diag), t(sub.upper), None)
        for p, itf in enumerate(sub.interfaces):
            self.mat.set_interface_coeffs(p, t(itf.bou_coeffs), None)
        nv = self.n + self.n_ext
        z = lambda: torch.zeros(nv, dtype=torch.float64, device=device)
        self.psi, self.src, self.pA, self.wA, self.rA = z(), z(), z(), z(), z()
        self.scal = torch.zeros(8, dtype=torch.float64, device=device)
        self.send = torch.zeros(max(self.n_ext, 1), dtype=torch.float64, device=device)
        self.addr.to_engine(t(sub.source), self.src)
        self.precond = precond
        self._psi_out = torch.zeros(self.n, dtype=torch.float64, device=device)

    def set_initial(self, psi0):
        if psi0 is None:
            self.psi.zero_()
        else:
            self.addr.to_engine(torch.from_numpy(np.ascontiguousarray(psi0)).to(self.device), self.psi)

    def begin(self, **controls):
        self.mat.dpcg_set_buffers(self.psi, self.src, self.pA, self.wA, self.rA, self.scal, self.send,
                                  precond=self.precond, **controls)

    def phase(self, k: int, it: int = 0, arg: float = 0.0):
        self.mat.dpcg_phase(k, it, arg)

    def status(self, history_len=0):
        return self.mat.dpcg_status(history_len)

    def solution(self) -> np.ndarray:
        self.addr.from_engine(self.psi, self._psi_out)
        torch.cuda.synchronize()
        return self._psi_out.cpu().numpy()

    def event_record(self, idx):
        self.mat.event_record(idx)

    def event_elapsed_ms(self, i0, i1):
        return self.mat.event_elapsed_ms(i0, i1)


class DistributedPCG:
    """PCG over the sub-domains of all ranks; ``sub`` is this rank's LduCase with interfaces."""

    def __init__(self, ctx, sub, device, precond: str = "diagonal", ops=None, n_global: Optional[int] = None):
        self.ops = ops if ops is not None else HipOps(ctx, sub, device, precond)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.nbrs: List[int] = [itf.nbr_domain for itf in sub.interfaces]
        self.nbr_patches: List[int] = [itf.nbr_patch for itf in sub.interfaces]
        # host loop: "native" = the engine's C++ loop calling RCCL directly (mi_dpcg_comm_*), "torch" = this
        # module's loop over torch.distributed.  Both run the same device phases; the injected test backend
        # (numpy over gloo) always uses the torch loop.
        self.driver = "torch" if ops is not None or torch.device(device).type != "cuda" else os.environ.get("MI_DPCG_DRIVER", "native")
        if self.driver == "torch" and len(set(self.nbrs)) != len(self.nbrs):
            raise ValueError("one processor patch per neighbour rank is supported (NCCL p2p is matched per peer, in order)")
        self.sizes = [len(itf.face_cells) for itf in sub.interfaces]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int64)
        self.n = sub.n_cells
        self.is_cuda = torch.device(device).type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=device) if self.is_cuda else None
        if n_global is None:
            ng = torch.tensor([float(self.n)], dtype=torch.float64, device=device)
            self._allreduce(ng)
            n_global = int(round(float(ng.item())))
        self.n_global = n_global
        self.it = 0
        self.history_len = 0
        # views and p2p descriptors are built once: nothing is allocated inside the iteration loop
        o = self.ops
        self._s01, self._s2, self._s3, self._s4 = o.scal[0:2], o.scal[2:3], o.scal[3:4], o.scal[4:5]
        self._p2p = {}
        self.comms = None
        self.allreduce = "torch.distributed"
        if self.driver == "native":
            # both loops drive the same HIP phases over RCCL; if the C++ side cannot set its communicators up on ANY
            # rank (e.g. librccl cannot be bound), all ranks agree to use the torch.distributed loop instead
            ok = 1
            try:
                self.comms = make_comms(ctx)
            except eng.MiError as e:  # pragma: no cover - needs a broken RCCL install
                ok = 0
                print(f"[parallel] native RCCL loop unavailable on rank {self.rank}: {e}", flush=True)
            if dist.is_initialized() and self.world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag.item())
            self.allreduce = "rccl"
            if ok and getattr(self.comms[0], "peer_mode", False):
                self.allreduce = "peer windows (one-shot stores into the peers' memory; self-tested at set-up)"
            if not ok:
                self.comms = None
                self.driver = "torch"
                self.allreduce = "torch.distributed"
                if len(set(self.nbrs)) != len(self.nbrs):
                    raise ValueError("one processor patch per neighbour rank is supported by the torch.distributed loop")

    # -- collectives ----------------------------------------------------------
    def _allreduce(self, t):
        if dist.is_initialized():  # also on a 1-rank group: same RCCL code path as N > 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _start_exchange(self, vec):
        """send the packed patch values, receive the neighbours' into vec[n:]; returns a waiter."""
        if self.world == 1 or not self.nbrs:
            return lambda: None
        o = self.ops
        p2p = self._p2p.get(id(vec))
        if p2p is None:
            p2p = []
            for k, nbr in enumerate(self.nbrs):
                a, b = int(self.offsets[k]), int(self.offsets[k + 1])
                p2p.append(dist.P2POp(dist.isend, o.send[a:b], nbr))
                p2p.append(dist.P2POp(dist.irecv, vec[self.n + a:self.n + b], nbr))
            self._p2p[id(vec)] = p2p
        if self.is_cuda and dist.get_backend() == "gloo":
            # test configuration only (several ranks sharing one GPU, tests/test_distributed.py): gloo's send/recv touch
            # the device buffers from the host without any stream ordering, so order them by hand
            torch.cuda.current_stream().synchronize()
            for r in dist.batch_isend_irecv(p2p):
                r.wait()
            return lambda: None
        if self.is_cuda:
            main = torch.cuda.current_stream()
            self.comm_stream.wait_stream(main)          # the pack kernel has to finish first
            with torch.cuda.stream(self.comm_stream):
                for r in dist.batch_isend_irecv(p2p):
                    r.wait()                            # stream-level wait on the comm stream only
            return lambda: main.wait_stream(self.comm_stream)
        reqs = dist.batch_isend_irecv(p2p)
        return lambda: [r.wait() for r in reqs]

    # -- solver ---------------------------------------------------------------
    def begin(self, psi0=None, tolerance=1e-6, rel_tol=0.0, max_iter=1000, min_iter=0):
        o = self.ops
        self.history_len = max_iter + 2
        o.set_initial(psi0)
        o.begin(tolerance=tolerance, relTol=rel_tol, maxIter=max_iter, minIter=min_iter, history_len=self.history_len)
        self.it = 0
        self.max_iter, self.min_iter = max_iter, min_iter
        if self.driver == "native":
            o.mat.dpcg_comm_begin(self.comms[0], self.comms[1], self.nbrs, self.nbr_patches, self.n_global)
            return
        o.phase(0)
        self._start_exchange(o.psi)()
        o.phase(1)
        self._allreduce(self._s3)
        avg = float(o.scal[3].item()) / self.n_global       # gAverage(psi): the one host read of the prologue
        o.phase(2, 0, avg)
        self._allreduce(self._s01)
        self._allreduce(self._s4)
        o.phase(3)
        self.it = 0
        self.max_iter, self.min_iter = max_iter, min_iter

    def iterate(self, n_iters: int, time_amul: bool = False, event_stride: int = 1):
        """enqueue n_iters iterations (device no-ops once converged); no host synchronisation.  time_amul: HIP events
        around the Amul phases of every event_stride-th iteration; returns their mean duration x n_iters (ms)."""
        o = self.ops
        import time
        t0 = time.perf_counter()
        if self.driver == "native":
            o.mat.dpcg_comm_iterate(n_iters, event_stride if time_amul else 0)
            self.it += n_iters
            self.last_enqueue_s = time.perf_counter() - t0   # host time to enqueue (no synchronisation inside)
            if time_amul:
                ns = (n_iters + event_stride - 1) // event_stride
                return sum(o.event_elapsed_ms(2 * k, 2 * k + 1) for k in range(ns)) * (n_iters / ns)
            return None
        for k in range(n_iters):
            it = self.it
            o.phase(10, it)
            wait = self._start_exchange(o.pA)
            rec = time_amul and k % event_stride == 0
            if rec:
                o.event_record(2 * (k // event_stride))
            o.phase(11, it)      # interior tiles overlap the exchange
            wait()
            o.phase(12, it)      # boundary tiles
            if rec:
                o.event_record(2 * (k // event_stride) + 1)
            self._allreduce(self._s2)
            o.phase(13, it)
            self._allreduce(self._s01)
            self.it += 1
        if n_iters > 0:
            o.phase(14, self.it - 1)
        self.last_enqueue_s = time.perf_counter() - t0
        if time_amul:
            ns = (n_iters + event_stride - 1) // event_stride
            return sum(o.event_elapsed_ms(2 * k, 2 * k + 1) for k in range(ns)) * (n_iters / ns)
        return None

    def end(self):
        return self.ops.status(self.history_len)

    def solve(self, psi0=None, tolerance=1e-6, rel_tol=0.0, max_iter=1000, min_iter=0, batch=16):
        self.begin(psi0, tolerance, rel_tol, max_iter, min_iter)
        st = self.ops.status(0)
        while not st["done"] and self.it <= max(max_iter, min_iter):
            self.iterate(batch)
            st = self.ops.status(0)
        return self.end()



def make_comms(ctx, device=None):
    """(reduce, halo) RCCL communicators of this rank: the unique ids come from rank 0 through torch.distributed."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world > 1 and os.environ.get("MI_COMM_TRANSPORT", "rccl") == "host":
        # rehearsals on a box with fewer GPUs than ranks (RCCL refuses ranks that share a device): the external-transport
        # hook over torch.distributed's host backend, peer windows on top unless MI_ALLREDUCE=rccl
        return make_host_comms(ctx, peer=os.environ.get("MI_ALLREDUCE", "auto") != "rccl")
    ids = [None]
    if rank == 0:
        ids = [[eng.Comm.unique_id(), eng.Comm.unique_id()]]
    if world > 1:
        dist.broadcast_object_list(ids, src=0)
    reduce_c = eng.Comm(ctx, world, rank, ids[0][0])
    halo_c = reduce_c if os.environ.get("MI_DPCG_ONE_COMM", "0") == "1" else eng.Comm(ctx, world, rank, ids[0][1])
    # Peer windows are the default between several ranks: mi_comm_peer_auto sets them up over the communicator's own
    # transport, self-tests their coherence and lets the ranks agree; when any rank cannot, every rank keeps RCCL.
    # MI_ALLREDUCE=rccl keeps RCCL; MI_ALLREDUCE=peer also switches a 1-rank communicator over (measurements).
    want = os.environ.get("MI_ALLREDUCE", "auto")
    reduce_c.peer_mode = False
    if want == "peer" or (want == "auto" and world > 1):
        reduce_c.peer_mode = reduce_c.peer_auto()
    return reduce_c, halo_c


def make_host_comms(ctx, peer: bool = False):
    """(reduce, halo) communicators over torch.distributed's HOST transport (gloo): the engine's external-transport hook
    (mi_comm_create_external) fed with device<->host copies + gloo collectives -- the shape of the reference's own path
    (host-staged MPI, processorFvPatchScalarField.C:77-84) and of a Pstream-backed shim.  Used where RCCL cannot connect the
    ranks: several engine ranks sharing one GPU (tests), or a process group that was initialised with gloo only."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    world, rank = dist.get_world_size(), dist.get_rank()

    def d2h(ptr, n):
        t = torch.empty(n, dtype=torch.float64)
        if n and hip.hipMemcpy(t.data_ptr(), ptr, 8 * n, 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        return t

    def h2d(ptr, t):
        if t.numel() and hip.hipMemcpy(ptr, t.data_ptr(), 8 * t.numel(), 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    def allreduce(ptr, n):
        t = d2h(ptr, n)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        h2d(ptr, t)

    def exchange(sends, recvs):
        reqs, landed = [], []
        for peer, tag, ptr, n in recvs:
            t = torch.empty(n, dtype=torch.float64)
            landed.append((ptr, t))
            if peer != rank:
                reqs.append(dist.irecv(t, src=peer, tag=tag))
        selfbox = {}
        for peer, tag, ptr, n in sends:
            t = d2h(ptr, n)
            if peer == rank:
                selfbox[tag] = t                     # a rank that is its own neighbour (cyclic posed as processor patches)
            else:
                reqs.append(dist.isend(t, dst=peer, tag=tag))
        for r in reqs:
            r.wait()
        for (peer, tag, ptr, n), (_, t) in zip(recvs, landed):
            h2d(ptr, selfbox[tag] if peer == rank else t)

    c = eng.ExternalComm(ctx, world, rank, allreduce, exchange)
    c.peer_mode = False
    if peer:
        c.peer_mode = c.peer_auto()      # windows between the processes (they share a GPU here); the set-up talks over gloo
    return c, c


def enable_peer_allreduce(comm):
    """Switch the small all-reduces of ``comm`` (<= 8 doubles: the scalars of the Krylov loops, GAMG's scale factors) to the
    one-shot peer-window form (mi_comm_peer_window / mi_comm_peer_connect): every rank exports its window, the 64-byte handles
    travel through torch.distributed, every rank maps its peers' windows.  All ranks call together."""
    world = dist.get_world_size()
    mine = comm.peer_window()
    handles = [None] * world
    dist.all_gather_object(handles, mine)
    comm.peer_connect(handles)
    dist.barrier()                                  # nobody writes into a window that is not mapped yet
    return comm


class DistributedMatrix:
    """This rank's part of a decomposed lduMatrix with its communicators attached (mi_matrix_attach_comm): the object
    the reference's solvers see on every MPI rank.  ``sub`` is an LduCase with processor interfaces; every solver entry
    point of the engine then solves the global system (all ranks call together), vectors are this rank's cells in
    caller order."""

    def __init__(self, ctx, sub, device, n_global: Optional[int] = None, comms=None):
        self.device = torch.device(device)
        self.sub = sub
        self.comms = comms if comms is not None else make_comms(ctx)
        # engine patches: the interfaces in order; a cyclicAMI interface whose partner lives on another rank gets a TRANSPORT
        # patch behind them (mi_addr_set_ami_patch_remote): a processor patch of the larger side's size that carries this side's
        # patch-internal field to the partner rank and receives the partner's, with zero coefficients
        patches = [np.asarray(i.face_cells, dtype=np.int32) for i in sub.interfaces]
        self.patch_rank = [i.nbr_domain for i in sub.interfaces]
        self.patch_nbr_patch = [i.nbr_patch for i in sub.interfaces]
        transport, transports = {}, {}
        for k, itf in enumerate(sub.interfaces):
            if getattr(itf, "ami_transport_nbr_patch", None) is not None:
                transport[k] = len(patches)
                patches.append(np.resize(np.asarray(itf.face_cells, dtype=np.int32), max(len(itf.face_cells), int(itf.ami_partner_size))))
                self.patch_rank.append(itf.nbr_domain); self.patch_nbr_patch.append(int(itf.ami_transport_nbr_patch))
            elif getattr(itf, "ami_parts", None):
                # the partner SIDE is split over several ranks: one transport patch per partner piece (mi_addr_set_ami_patch_remote_multi),
                # each of the larger of the two pieces' sizes (both ranks create it with the same size)
                transports[k] = []
                for (q, _), nq, tq in zip(itf.ami_parts, itf.ami_part_sizes, itf.ami_transport_nbr_patches):
                    transports[k].append(len(patches))
                    patches.append(np.resize(np.asarray(itf.face_cells, dtype=np.int32), max(len(itf.face_cells), int(nq))))
                    self.patch_rank.append(int(q)); self.patch_nbr_patch.append(int(tq))
        self.addr = eng.Addressing(ctx, sub.n_cells, sub.lower_addr, sub.upper_addr, patches)
        for k, itf in enumerate(sub.interfaces):
            if getattr(itf, "ami_start", None) is None:
                continue
            if k in transport:
                self.addr.set_ami_patch_remote(k, transport[k], int(itf.ami_partner_size), itf.ami_start, itf.ami_addr, itf.ami_w, itf.ami_low)
            elif k in transports:
                self.addr.set_ami_patch_remote_multi(k, transports[k], itf.ami_part_sizes, itf.ami_start, itf.ami_addr, itf.ami_w, itf.ami_low)
            else:
                self.addr.set_ami_patch(k, itf.nbr_patch, itf.ami_start, itf.ami_addr, itf.ami_w, itf.ami_low)
            if itf.ami_magsf is not None:
                self.addr.set_ami_face_areas(k, itf.ami_magsf)
        self.mat = eng.Matrix(self.addr)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.mat.set_coeffs(t(sub.diag), t(sub.upper), None if sub.lower is None else t(sub.lower))
        for p, itf in enumerate(sub.interfaces):
            self.mat.set_interface_coeffs(p, t(itf.bou_coeffs), None if sub.lower is None else t(itf.int_coeffs))
            if getattr(itf, "transform", 1.0) != 1.0:       # processorCyclic / cyclicAMI: transformCoupleField factor
                self.mat.set_patch_transform(p, itf.transform)
        for p in list(transport.values()) + [p for ps in transports.values() for p in ps]:
            z = t(np.zeros(len(patches[p])))
            self.mat.set_interface_coeffs(p, z, None if sub.lower is None else z)
        if n_global is None:
            ng = torch.tensor([float(sub.n_cells)], dtype=torch.float64, device=self.device)
            self.comms[0].allreduce_sum(ng)
            torch.cuda.synchronize()
            n_global = int(round(float(ng.item())))
        self.n_global = n_global
        self.mat.attach_comm(self.comms[0], self.comms[1], self.patch_rank, self.patch_nbr_patch, n_global)
        self._gamg = None

    def gamg(self, face_weights, n_cells_in_coarsest_level=10):
        """GAMG hierarchy of the decomposed case (cached, like the reference's GAMGAgglomeration MeshObject)"""
        if self._gamg is None:
            self._gamg = eng.Gamg(self.addr, face_weights, n_cells_in_coarsest_level, comms=self.comms,
                                  patch_rank=self.patch_rank, patch_nbr_patch=self.patch_nbr_patch)
        return self._gamg

    def solve(self, solver: str, psi, source, **kw):
        """solver: PCG | PBiCG | PBiCGStab | smoothSolver | GAMG (GAMG needs face_weights=...)"""
        if solver == "GAMG":
            g = self.gamg(kw.pop("face_weights"), kw.pop("n_cells_in_coarsest_level", 10))
            return g.solve(self.mat, psi, source, **kw)
        fn = {"PCG": self.mat.pcg, "PBiCG": self.mat.pbicg, "PBiCGStab": self.mat.pbicgstab, "smoothSolver": self.mat.smooth_solve}[solver]
        return fn(psi, source, **kw)
