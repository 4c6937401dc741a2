"""One process per GPU: control plane on torch.distributed (gloo), data plane on RCCL inside libelag_hip.so.

The reference runs one MPI rank per sub-domain and exchanges halos inside hypre.  Here ``init`` creates
the RCCL communicator the library uses for the neighbour exchange (rank 0 makes the unique id, the host broadcasts
it), and ``build_hierarchy`` hands every rank its rows of every level's operators.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib, linalg as la, partition


class Dist:
    def __init__(self, rank, world, td):
        self.rank, self.world, self._td = rank, world, td

    def barrier(self):
        self._td.barrier()

    def allreduce_max(self, v: float) -> float:
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self._td.all_reduce(t, op=self._td.ReduceOp.MAX)
        return float(t[0])

    def allreduce_sum(self, v: float) -> float:
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self._td.all_reduce(t, op=self._td.ReduceOp.SUM)
        return float(t[0])


def init(rank: int, world: int) -> Dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # Load the HIP library (and, across ranks, the system RCCL) BEFORE torch: the dynamic loader then binds
    # everything in this process to the one ROCm runtime the kernels were built with; torch is used on the CPU
    # only (gloo rendezvous / barriers), so its bundled copies of the same sonames are never needed.
    lib = _lib.init(int(os.environ.get("LOCAL_RANK", rank)))
    if world > 1:
        for name in ("/opt/rocm/lib/librccl.so.1", "librccl.so.1"):
            try:
                C.CDLL(name, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
    import torch
    import torch.distributed as td
    if not td.is_initialized():
        td.init_process_group("gloo", rank=rank, world_size=world)
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        _lib.check(lib.elag_hip_comm_unique_id(buf))
        uid = torch.tensor(list(buf), dtype=torch.uint8)
    td.broadcast(uid, src=0)
    buf = (C.c_ubyte * 128)(*uid.tolist())
    _lib.check(lib.elag_hip_comm_init(rank, world, buf))
    return Dist(rank, world, td)


def build_hierarchy(dist: Dist, A, Ps, seq, smoother="L1 Jacobi", params=None, replicate_from=1):
    """Distributed 0-form hierarchy.  Levels < ``replicate_from`` are row-partitioned over the ranks (halo exchange per
    operator apply, hypre's ParCSR model); the levels from ``replicate_from`` down are far too small to be worth a halo
    exchange and live, complete, on every GPU as an ordinary single-GPU hierarchy (fused levels, dense tail, hipGraph).
    The restriction into the first replicated level produces per-rank partial sums (its P is stored with global coarse
    columns) that one RCCL all-reduce completes, and the prolongation back needs no exchange at all -- with
    ``replicate_from = 1`` a V-cycle costs three RCCL rounds.  Returns (H, n_local, global level shapes, setup s)."""
    import scipy.sparse as sp
    t0 = time.time()
    d = partition.distributed_h1_hierarchy(A, Ps, seq, dist.world)
    nl = len(d["levels"])
    R = max(1, min(int(replicate_from), nl - 1))
    t = la.HYPRE_SMOOTHER_TYPES[smoother] if isinstance(smoother, str) else int(smoother)
    lib = _lib.init()
    # distributed head: levels 0..R-1; their Galerkin operators are formed on the device across the ranks
    # (elag_hip_parcsr_rap with distributed operands + FixZeroRows), inputs are the fine operator and the true P's
    H = la.Hierarchy(R + 1)
    ops = [la.ParallelCSRMatrix.from_parts(d["levels"][0]["A_parts"][dist.rank])]
    Pd = [la.ParallelCSRMatrix.from_parts(d["levels"][l]["P_parts"][dist.rank]) for l in range(1, R + 1)]
    for l in range(1, R + 1):
        Ac = la.RAP(ops[-1], Pd[l - 1])
        Ac.fix_zero_rows()
        ops.append(Ac)
    H._keep_smoothers = []
    for l in range(R):
        H.SetOperator(l, ops[l])
        if l > 0:
            H.SetProlongator(l, Pd[l - 1])
        S = la.HypreSmootherWrapper(ops[l], t, params or {})
        H.SetSmoothers(l, S, S)
        H._keep_smoothers.append(S)
    # replicated tail: level R gathered onto every GPU, the levels below built there redundantly (device RAP)
    subA, _starts = ops[R].Replicate()
    subP = [la.ParallelCSRMatrix.from_scipy(P) for P in d["P_global"][R:]]
    Hsub = la.buildHierarchyFromDeRhamSequence(subA, subP, smoother, params or {}, reorder=0)
    hc = C.c_void_p()
    _lib.check(lib.elag_hip_coarse_create_from_hierarchy(C.byref(hc), Hsub.h, 1))
    cs = la.CoarseSolver.__new__(la.CoarseSolver)
    cs._lib, cs.A, cs.h, cs.iterative_mode = lib, subA, hc, False
    H.SetOperator(R, subA)
    # P from the first replicated level: this rank's rows, GLOBAL coarse columns, no comm package
    s = d["levels"][R - 1]["starts"]
    Ploc = sp.csr_matrix(d["P_global"][R - 1])[int(s[dist.rank]):int(s[dist.rank + 1])].tocsr()
    H.SetProlongator(R, la.ParallelCSRMatrix.from_scipy(Ploc))
    H.SetCoarseSolver(cs)
    H.Finalize(use_graph=False)
    H._keep += [Hsub, subA, *subP, *ops, *Pd]
    la.synchronize()
    shapes = []
    for l in range(nl):
        Ag = d["A_global"][l]
        Pg = d["P_global"][l] if l < nl - 1 else None
        shapes.append((Ag.shape[0], Ag.nnz, Pg.shape[1] if Pg is not None else 0, Pg.nnz if Pg is not None else 0))
    n_local = ops[0].Height()
    H._dist_info = d
    return H, n_local, shapes, time.time() - t0
