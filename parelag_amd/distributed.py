"""One process per GPU: control plane on torch.distributed (gloo), data plane on RCCL inside libelag_hip.so.

The reference runs one MPI rank per sub-domain and exchanges halos inside hypre.  Here ``init`` creates
the RCCL communicator the library uses for the neighbour exchange (rank 0 makes the unique id, the host broadcasts
it), and ``build_hierarchy`` hands every rank its rows of every level's operators.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import socket
import sys
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


@contextlib.contextmanager
def stdout_to_stderr():
    """fd-level redirect of stdout to stderr (gloo announces its connections on the C++ stdout; a benchmark's stdout carries
    exactly one JSON line)"""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def init(rank: int, world: int) -> Dist:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        # a launcher (torch.distributed.run, bench.py --gpus N) always provides the port; only a lone rank gets here without
        # one, and a lone rank may take any free port
        if world > 1:
            raise RuntimeError("distributed.init: MASTER_PORT is not set (start the ranks with torch.distributed.run or bench.py --gpus N)")
        os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if int(os.environ.get("NNODES", os.environ.get("GROUP_WORLD_SIZE", "1")) or 1) <= 1:
        # one node: RCCL's bootstrap sockets on the loopback interface (the container's hostname may not resolve)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        # ... and gloo's: without it gloo looks its interface up through the host name, which may not resolve (or may take minutes not to)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
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
        with stdout_to_stderr():
            if world == 1:
                # a lone rank needs no rendezvous: an in-process store (the TCP store looks host names up, which on a box whose name does
                # not resolve costs a warning at best and minutes at worst)
                td.init_process_group("gloo", store=td.HashStore(), rank=0, world_size=1)
            else:
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


def _build_from_pieces(dist, A_part, P_parts, Ploc, P_tail, level_rows, P_shapes, smoother, params, R, nl):
    """The device side of ``build_hierarchy``: this rank's ParCSR pieces of the fine operator and of the interpolators of the
    distributed head (levels 0..R-1), its rows of the interpolator into the first replicated level (global coarse columns) and the
    complete interpolators below.  Every Galerkin product is formed on the device."""
    t = la.HYPRE_SMOOTHER_TYPES[smoother] if isinstance(smoother, str) else int(smoother)
    lib = _lib.init()
    # distributed head: levels 0..R-1; their Galerkin operators are formed on the device across the ranks
    # (elag_hip_parcsr_rap with distributed operands + FixZeroRows), inputs are the fine operator and the true P's
    H = la.Hierarchy(R + 1)
    as_dev = lambda p: p if isinstance(p, la.ParallelCSRMatrix) else la.ParallelCSRMatrix.from_parts(p)      # (pieces as dicts, or made on the device already)
    ops = [as_dev(A_part)]
    Pd = [as_dev(p) for p in P_parts]
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
    subP = [la.ParallelCSRMatrix.from_scipy(P) for P in P_tail]
    Hsub = la.buildHierarchyFromDeRhamSequence(subA, subP, smoother, params or {}, reorder=0)
    hc = C.c_void_p()
    _lib.check(lib.elag_hip_coarse_create_from_hierarchy(C.byref(hc), Hsub.h, 1))
    cs = la.CoarseSolver.__new__(la.CoarseSolver)
    cs._lib, cs.A, cs.h, cs.iterative_mode = lib, subA, hc, False
    H.SetOperator(R, subA)
    # P from the first replicated level: this rank's rows, GLOBAL coarse columns, no comm package
    H.SetProlongator(R, la.ParallelCSRMatrix.from_scipy(Ploc))
    H.SetCoarseSolver(cs)
    H.Finalize(use_graph=False)
    H._keep += [Hsub, subA, *subP, *ops, *Pd]
    la.synchronize()
    # global level shapes: distributed levels from the summed local entry counts, replicated levels from the complete copies
    shapes = []
    for l in range(nl):
        pc, pn = P_shapes[l] if l < nl - 1 else (0, 0)
        if l < R:
            v = C.c_double(float(ops[l].nnz))
            _lib.check(lib.elag_hip_comm_allreduce(C.byref(v), 0))          # the library's communicator (RCCL / in-process), sum
            rows, nnz = int(level_rows[l]), int(round(v.value))
        else:
            op = Hsub.GetOperator(l - R)
            rows, nnz = op.Height(), op.nnz
        shapes.append((rows, nnz, int(pc), int(pn)))
    return H, ops[0].Height(), shapes


def build_hierarchy(dist: Dist, A, Ps, seq, smoother="L1 Jacobi", params=None, replicate_from=1):
    """Distributed 0-form hierarchy.  Levels < ``replicate_from`` are row-partitioned over the ranks (halo exchange per
    operator apply, hypre's ParCSR model); the levels from ``replicate_from`` down are far too small to be worth a halo
    exchange and live, complete, on every GPU as an ordinary single-GPU hierarchy (fused levels, dense tail, hipGraph).
    The restriction into the first replicated level produces per-rank partial sums (its P is stored with global coarse
    columns) that one RCCL all-reduce completes, and the prolongation back needs no exchange at all -- with
    ``replicate_from = 1`` a V-cycle costs three RCCL rounds.  Returns (H, n_local, global level shapes, setup s)."""
    import scipy.sparse as sp
    t0 = time.time()
    nl = len(Ps) + 1
    R = max(1, min(int(replicate_from), nl - 1))
    # host side: numbering and per-rank pieces of the fine operator and of the interpolators of the distributed head only;
    # every Galerkin product is formed on the device
    # (the numbering and the operators in it: what a ParElag rank is handed by its mesh partition.  No per-rank pieces are cut on the host --
    # split_levels = 0 -- : this rank's rows go to the device with global columns and the library makes the ParCSR pieces)
    d = partition.distributed_h1_hierarchy(A, Ps, seq, dist.world, galerkin=False, split_levels=0)
    r = dist.rank
    st = [d["levels"][l]["starts"] for l in range(nl)]
    rows = lambda M, l: sp.csr_matrix(M)[int(st[l][r]):int(st[l][r + 1])].tocsr()
    A_rows = rows(d["A_global"][0], 0)
    P_rows = [rows(d["P_global"][l - 1], l - 1) for l in range(1, R + 1)]
    Ploc = rows(d["P_global"][R - 1], R - 1)
    partition_s = time.time() - t0
    t1 = time.time()

    def piece(M, rs, cs):
        return la.ParallelCSRMatrix.from_global_rows(M, int(rs[r]), int(rs[-1]), int(cs[r]), int(cs[r + 1] - cs[r]), int(cs[-1]))
    A_part = piece(A_rows, st[0], st[0])
    P_parts = [piece(P_rows[l - 1], st[l - 1], st[l]) for l in range(1, R + 1)]
    H, n_local, shapes = _build_from_pieces(
        dist, A_part, P_parts, Ploc, d["P_global"][R:],
        [int(st[l][-1]) for l in range(R)], [(P.shape[1], P.nnz) for P in d["P_global"]], smoother, params, R, nl)
    H._dist_info = d
    H._partition_s = partition_s                       # host: block numbering of every level, the operators permuted into it (the generator's side)
    return H, n_local, shapes, time.time() - t1


def build_hierarchy_sharded(dist: Dist, comm, shards, smoother="L1 Jacobi", params=None, replicate_from=1):
    """``build_hierarchy`` from the shards of ``meshgen_sharded.sharded_h1_problem``: no rank ever holds a global operator of a
    distributed level -- the ParCSR pieces and their comm packages come from the rank's own rows and one all-gather of the ghost
    lists per operator (the role of ``hypre_MatvecCommPkgCreate``, call sites in SURVEY 2.3); only the interpolators of the
    replicated tail (levels with at most ~2^17 dofs) are gathered.  ``comm``: the control-plane communicator the shards were made
    with.  Returns (H, n_local, global level shapes, setup s)."""
    import scipy.sparse as sp
    from . import meshgen_sharded as msh
    t0 = time.time()
    starts, P_rows = shards["starts"], shards["P_rows"]
    nl = len(P_rows) + 1
    R = max(1, min(int(replicate_from), nl - 1))
    # the ParCSR pieces: the rank's rows go to the device as they are (global columns); the split into diag / offd blocks and the comm package
    # are the library's (elag_hip_parcsr_create_from_global_csr) -- round 3 split them with scipy on the host: 0.45 of the 0.50 s of this
    # setup at cube456 x5 (profiles/r04_dist_setup_probe.txt)
    r = dist.rank

    def piece(rows, rs, cs):
        return la.ParallelCSRMatrix.from_global_rows(rows, int(rs[r]), int(rs[-1]), int(cs[r]), int(cs[r + 1] - cs[r]), int(cs[-1]))
    A_part = piece(shards["A_rows"], starts[0], starts[0])
    P_parts = [piece(P_rows[l - 1], starts[l - 1], starts[l]) for l in range(1, R + 1)]
    tail = msh.agree(comm, lambda: ([P_rows[l] for l in range(R, nl - 1)], [P.nnz for P in P_rows]))
    P_tail = [sp.vstack([t[0][i] for t in tail]).tocsr() for i in range(nl - 1 - R)]
    P_shapes = [(int(starts[l + 1][-1]), sum(t[1][l] for t in tail)) for l in range(nl - 1)]
    H, n_local, shapes = _build_from_pieces(dist, A_part, P_parts, P_rows[R - 1], P_tail, [int(starts[l][-1]) for l in range(R)], P_shapes,
                                            smoother, params, R, nl)
    H._dist_info = dict(levels=[dict(starts=st) for st in starts])
    return H, n_local, shapes, time.time() - t0


def hex_darcy_problem(dist: Dist, n, h=(20.0, 10.0, 2.0), kinv=None, parts=None):
    """BASELINE configs[4] on several ranks: the mixed Darcy system on the Cartesian grid of ``examples/MultigridTestSPE10.cpp:194-208``
    with ParElag's own coarse spaces, the coarsening SHARDED over the ranks.  The grid is cut into row blocks of whole coarsest
    agglomerates (``partition.hex_element_owners``); on every level a rank runs ``DeRhamSequence::Coarsen`` (``jformStart = 2``,
    device kernels) for the agglomerates it owns -- agglomerated facets on an interface are computed by both neighbours from the
    same numbers, bit for bit -- and the pieces are all-gathered for the row-block builders (``derham_dist.coarsen_allgather``;
    the reference's counterpart is the shared-entity ownership of ``src/amge/DeRhamSequence.cpp:1820,1971,2024``).
    Returns (dar, dd): the ``dar`` dict of ``meshgen.darcy_problem`` (+ ``levels``, ``elem_owners``, ``amge`` record) and the
    row-block decomposition ``partition.distributed_darcy_levels`` made of it."""
    from . import derham_dist
    from .derham_level import seq_level_from_hex
    t0 = time.time()
    lv = seq_level_from_hex(tuple(n), h=h, jstart=2, kinv=kinv)
    eown = partition.hex_element_owners(n, parts, dist.world)
    gen_s = time.time() - t0
    levels, Ps, recs = [lv], [], []
    t0 = time.time()
    for l, part in enumerate(parts):
        t1 = time.time()
        # agglomerate a of level l is element a of level l + 1: its owner is that element's rank
        res = derham_dist.coarsen_allgather(levels[-1], np.asarray(part, dtype=np.int64), dist.rank, dist.world, owner=eown[l + 1])
        c = res["coarse"]
        recs.append({"agglomerates": int(c.n[0]), "agglomerated_facets": int(c.n[1]), "dofs2": [levels[-1].ndof(2), c.ndof(2)],
                     "dofs3": [levels[-1].ndof(3), c.ndof(3)], "mine": int(res["piece"].agglomerates.size), "wall_s": round(time.time() - t1, 4)})
        Ps.append((res["P"][2].tocsr(), res["P"][3].tocsr()))
        levels.append(c)
    total_s = time.time() - t0
    M = lv.mass(2)
    W = lv.mass(3)
    B = (W @ lv.D[2]).tocsr(); B.sort_indices()
    Bt = B.T.tocsr(); Bt.sort_indices()
    dar = dict(M=M, B=B, Bt=Bt, Ps=Ps, seq=None, rhs_p=W @ lv.const3, kinv=kinv, levels=levels, elem_owners=eown,
               amge={"what": "DeRhamSequence::Coarsen (jformStart=2) on the device, sharded over %d rank(s) by agglomerate ownership: hex %dx%dx%d, "
                             "2x2x2 box agglomerates" % ((dist.world,) + tuple(n)),
                     "levels": recs, "coarsen_total_s": round(total_s, 3), "generate_s": round(gen_s, 2)})
    dd = partition.distributed_darcy_levels(dar, levels, eown, dist.world)
    return dar, dd


def build_darcy_hierarchy(dist: Dist, dar, block_smoother="Block Jacobi", smoother="L1 Jacobi", params=None, use_negative_s=True, dd=None):
    """Distributed blocked (Darcy) hierarchy in the same layout as ``build_hierarchy``: the fine level is row-partitioned
    (faces and elements owned per rank, one merged operator = one halo exchange per product, block smoother on this
    rank's sub-vectors); the levels below are complete on every GPU as an ordinary single-GPU blocked hierarchy behind
    one all-reduce of the restricted residual.  Returns (H, n_local, global level shapes, setup s); ``H._dist_info``
    carries the numbering (``order`` / ``starts`` per space)."""
    import scipy.sparse as sp
    t0 = time.time()
    lib = _lib.init()
    r = dist.rank
    dd = partition.distributed_darcy(dar, dist.world) if dd is None else dd
    (s2, s3) = dd["starts"][0]
    off = [0, int(s2[r + 1] - s2[r]), int(s2[r + 1] - s2[r] + s3[r + 1] - s3[r])]
    op = la.MfemBlockOperator(off)
    for k, parts in dd["blocks"].items():
        op.SetBlock(*k, la.ParallelCSRMatrix.from_parts(parts[r]))
    t = la.HYPRE_SMOOTHER_TYPES[smoother] if isinstance(smoother, str) else int(smoother)
    # fine-level block smoother on the distributed blocks
    S = la.SchurComplement(op, "DIAGONAL", 1.0, use_negative_s)
    i0, i1 = la.HypreSmootherWrapper(op.GetBlock(0, 0), t, params or {}), la.HypreSmootherWrapper(S, t, params or {})
    inv = [i0, i0, i0, i1] if block_smoother == "Block LDU" else [i0, i1]
    bs = la.BlockSolver(op, inv, block_smoother)
    # level-1 blocks: distributed Galerkin products, gathered onto every rank (each space keeps its block numbering)
    P0 = [la.ParallelCSRMatrix.from_parts(p[r]) for p in dd["P"][0]]
    rep = {}
    for (i, j), _parts in dd["blocks"].items():
        C_ij = la.RAP(P0[i], op.GetBlock(i, j), P0[j])
        rep[(i, j)], _st = C_ij.Replicate()
        del C_ij
    n0c, n1c = rep[(0, 0)].Height(), rep[(1, 0)].Height()
    sub_op = la.MfemBlockOperator([0, n0c, n0c + n1c])
    for k, M in rep.items():
        sub_op.SetBlock(*k, M)
    (o2c, o3c) = dd["order"][1]
    sub_Ps = []
    for l, (P2, P3) in enumerate(dar["Ps"][1:], start=1):
        if l == 1:
            P2, P3 = sp.csr_matrix(P2)[o2c], sp.csr_matrix(P3)[o3c]      # rows in level-1 block numbering, natural columns
        sub_Ps.append([la.ParallelCSRMatrix.from_scipy(partition.canon(P2)), la.ParallelCSRMatrix.from_scipy(partition.canon(P3))])
    Hsub = la.buildBlockedHierarchyFromDeRhamSequence(sub_op, sub_Ps, block_smoother=block_smoother, a00_inverse=smoother,
                                                     a00_params=params, a11_inverse=smoother, a11_params=params,
                                                     use_negative_s=use_negative_s)
    hc = C.c_void_p()
    _lib.check(lib.elag_hip_coarse_create_from_hierarchy(C.byref(hc), Hsub.h, 1))
    cs = la.CoarseSolver.__new__(la.CoarseSolver)
    subA = sub_op.Monolithic()
    cs._lib, cs.A, cs.h, cs.iterative_mode = lib, subA, hc, False
    # P from level 1: this rank's rows [faces; elements], GLOBAL level-1 columns [faces; elements], no comm package
    (o2, o3) = dd["order"][0]
    P2g = sp.csr_matrix(dar["Ps"][0][0])[o2[int(s2[r]):int(s2[r + 1])]][:, o2c]
    P3g = sp.csr_matrix(dar["Ps"][0][1])[o3[int(s3[r]):int(s3[r + 1])]][:, o3c]
    Ploc = la.ParallelCSRMatrix.from_scipy(partition.canon(sp.block_diag([P2g, P3g]).tocsr()))
    H = la.Hierarchy(2)
    A0 = op.Monolithic()
    H.SetOperator(0, A0)
    H.SetSmoothers(0, bs, bs)
    H.SetOperator(1, subA)
    H.SetProlongator(1, Ploc)
    H.SetCoarseSolver(cs)
    H.Finalize(use_graph=False)
    H._keep += [Hsub, sub_op, subA, rep, sub_Ps, op, S, i0, i1, bs, P0, Ploc, A0, cs]
    la.synchronize()
    shapes = []
    shapes.append((dar["M"].shape[0] + dar["B"].shape[0], dar["M"].nnz + 2 * dar["B"].nnz,
                   sum(P.shape[1] for P in dar["Ps"][0]), sum(P.nnz for P in dar["Ps"][0])))
    for l in range(len(dar["Ps"])):
        opl = Hsub.GetOperator(l)
        pn = sum(P.shape[1] for P in dar["Ps"][l + 1]) if l + 1 < len(dar["Ps"]) else 0
        pz = sum(P.nnz for P in dar["Ps"][l + 1]) if l + 1 < len(dar["Ps"]) else 0
        shapes.append((opl.Height(), opl.nnz, pn, pz))
    H._dist_info = dd
    H._block_op = op
    return H, off[-1], shapes, time.time() - t0


def _two_layer_hierarchy(A_d, P0_d, Ploc_sp, sub_Ps_sp, smoother, params=None):
    """A scalar hierarchy in the layout of ``build_hierarchy`` on a distributed fine operator: level 0 row-partitioned with its own
    relaxation; the Galerkin operator ``P0^T A P0`` formed across the ranks, gathered, and the levels below built complete on
    every GPU (device RAP chain) behind one all-reduce of the restricted residual.  ``P0_d``: the interpolator's rows of this rank
    with the distributed coarse numbering (for the Galerkin product); ``Ploc_sp``: the same rows with GLOBAL coarse columns (for
    the cycle); ``sub_Ps_sp``: interpolators of the replicated levels.  Returns (H, keepalive)."""
    lib = _lib.init()
    t = la.HYPRE_SMOOTHER_TYPES[smoother] if isinstance(smoother, str) else int(smoother)
    Ac = la.RAP(P0_d, A_d, P0_d)
    Ac.fix_zero_rows()
    subA, _st = Ac.Replicate()
    del Ac
    subP = [la.ParallelCSRMatrix.from_scipy(partition.canon(P)) for P in sub_Ps_sp]
    Hsub = la.buildHierarchyFromDeRhamSequence(subA, subP, smoother, params or {}, reorder=0)       # (no level below: a dense solve)
    hc = C.c_void_p()
    _lib.check(lib.elag_hip_coarse_create_from_hierarchy(C.byref(hc), Hsub.h, 1))
    cs = la.CoarseSolver.__new__(la.CoarseSolver)
    cs._lib, cs.A, cs.h, cs.iterative_mode = lib, subA, hc, False
    S0 = la.HypreSmootherWrapper(A_d, t, params or {})
    Ploc = la.ParallelCSRMatrix.from_scipy(partition.canon(Ploc_sp))
    H = la.Hierarchy(2)
    H.SetOperator(0, A_d)
    H.SetSmoothers(0, S0, S0)
    H.SetOperator(1, subA)
    H.SetProlongator(1, Ploc)
    H.SetCoarseSolver(cs)
    H.Finalize(use_graph=False)
    H._keep += [Hsub, subA, subP, S0, Ploc, cs, A_d, P0_d]
    return H


def build_darcy_ldu_amge(dist: Dist, dar, H_blocked, smoother="L1 Jacobi", params=None):
    """The reference's SPE10 solver graph (``examples/example_parameterlists/spe10_example_parameters.xml:95-127``) on N ranks: Block LDU
    whose three ``A00`` inverses are AMGe V-cycles on the velocity mass matrix ``M`` (2-form interpolators) and whose ``S`` inverse is an
    AMGe V-cycle on ``B diag(M)^-1 B^T`` (3-form interpolators), negated -- every one of them a distributed hierarchy in the layout of
    ``build_hierarchy`` (fine level row-partitioned, the rest replicated behind one all-reduce).  ``H_blocked`` is the hierarchy of
    ``build_darcy_hierarchy`` on the same decomposition (its fine-level block operator and numbering are reused).  Returns the
    preconditioner (a ``Block2x2LDUInverseOperator`` with ``iterative_mode = False``)."""
    import scipy.sparse as sp
    dd, op, r = H_blocked._dist_info, H_blocked._block_op, dist.rank
    (s2, s3) = dd["starts"][0]
    (o2, o3) = dd["order"][0]
    (o2c, o3c) = dd["order"][1]
    P0 = [la.ParallelCSRMatrix.from_parts(p[r]) for p in dd["P"][0]]
    Ps = dar["Ps"]
    # form k (0: velocity / 2-form, 1: pressure / 3-form): rows of this rank with global level-1 columns, and the replicated levels'
    # interpolators (level 1 rows in the level-1 block numbering, natural columns below)
    def pieces(k, o, s, oc):
        Pg = sp.csr_matrix(Ps[0][k])[o[int(s[r]):int(s[r + 1])]][:, oc]
        sub = [sp.csr_matrix(P[k])[oc] if l == 1 else sp.csr_matrix(P[k]) for l, P in enumerate(Ps[1:], start=1)]
        return Pg, sub
    P2g, sub2 = pieces(0, o2, s2, o2c)
    P3g, sub3 = pieces(1, o3, s3, o3c)
    Md = op.GetBlock(0, 0)
    Spos = la.SchurComplement(op, "DIAGONAL", 1.0, True)            # -(0 - B diag(M)^-1 B^T) = B diag(M)^-1 B^T, distributed
    HM = _two_layer_hierarchy(Md, P0[0], P2g, sub2, smoother, params)
    HS = _two_layer_hierarchy(Spos, P0[1], P3g, sub3, smoother, params)
    iM, iS = la.HierarchySolver(HM), la.HierarchySolver(HS, scale=-1.0)
    ldu = la.Block2x2LDUInverseOperator(op, iM, iM, iM, iS)
    ldu.iterative_mode = False
    ldu._keep_ldu = [HM, HS, iM, iS, Spos, Md, P0]
    return ldu
