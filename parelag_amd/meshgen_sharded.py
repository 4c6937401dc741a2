"""Sharded generator of the refined-cube456 0-form problem: every rank builds only its own sub-domain (INPUT generator).

The reference never holds a global mesh on one rank: ``mfem::ParMesh`` gives every MPI rank its sub-domain
(``examples/MultigridTest0Form.cpp:187``), the ranks refine in place and ``SharingMap`` matches the shared entities
(``src/structures/SharingMap.hpp:266-287``; the master of a shared entity is the lowest rank that holds it).
``meshgen.h1_problem`` builds the whole problem in one process, which stops being possible where the multi-GPU runs get
interesting (cube456 x6: 20 M dofs, 35 GB of host arrays and minutes of numpy per rank, on every rank).  Here

* the 456 base tets are dealt out by the same recursive coordinate bisection as ``partition.element_partitions``;
* every rank refines its own sub-mesh.  ``TetMesh.refine`` numbers the new vertices by the sorted edge list and the
  local vertex numbers are an order-preserving restriction of the global ones, so the local sub-mesh *is* the restriction
  of the global refined mesh (same children, same Bey diagonals) -- nothing has to be communicated to refine;
* the faces a rank shares with other ranks are carried along as a triangle surface (a red-refined tet face is four
  red-refined triangles); their vertices are the only shared dofs.  One all-gather per level matches them by their
  coordinates (midpoints of identical operands are identical floating-point numbers on every rank), the lowest rank
  becomes the owner, and each rank numbers its owned vertices contiguously in local order -- which is exactly
  ``partition.block_numbering`` of the global problem, so the shards are the rows ``partition.split_rows`` cuts out of
  the serial generator's matrices (tests/test_meshgen_sharded.py compares them);
* the fine operator is sub-assembled per rank (Neumann on the interface) and the rows of shared vertices are sent to
  their owners -- ``SharingMap::Assemble`` for a generator; the interpolators need no exchange at all (both parents of a
  fine vertex are local to every rank that holds it).

Every exchange is an "agreement": a rank that fails inside its local work says so in the exchange and all ranks raise
together (a lone exception would leave the others waiting in the next collective).

Nothing in here is on the product path: it only manufactures inputs.
"""
from __future__ import annotations

import threading

import numpy as np
import scipy.sparse as sp

from .meshgen import TetMesh, interp_P0, load_cube456
from .partition import rcb_partition


# ---- communicators (control plane only: pickled numpy arrays) -----------------------------------------------------

class SingleComm:
    rank, world = 0, 1

    def allgather(self, obj):
        return [obj]

    def gather0(self, obj):
        return [obj]


class TorchComm:
    """all-gather of Python objects over a torch.distributed process group (gloo)"""

    def __init__(self, td, rank, world):
        self._td, self.rank, self.world = td, rank, world

    def allgather(self, obj):
        out = [None] * self.world
        self._td.all_gather_object(out, obj)
        return out

    def gather0(self, obj):
        """gather onto rank 0 (None elsewhere)"""
        out = [None] * self.world if self.rank == 0 else None
        self._td.gather_object(obj, out, dst=0)
        return out


class ThreadComm:
    """N ranks as threads of one process (tests, and the in-process communicator of the multi-rank GPU suite)"""

    class Group:
        def __init__(self, world):
            self.world, self.slots, self.barrier = world, [None] * world, threading.Barrier(world)

    def __init__(self, group, rank):
        self._g, self.rank, self.world = group, rank, group.world

    def allgather(self, obj):
        g = self._g
        g.slots[self.rank] = obj
        g.barrier.wait(timeout=600)
        out = list(g.slots)
        g.barrier.wait(timeout=600)
        return out

    def gather0(self, obj):
        out = self.allgather(obj)
        return out if self.rank == 0 else None


def agree(comm, fn):
    """run ``fn()`` on this rank, all-gather the results; if any rank raised, every rank raises"""
    try:
        mine = ("ok", fn())
    except Exception as exc:  # noqa: BLE001 - the text travels to the other ranks
        import traceback
        mine = ("err", "rank %d: %s" % (comm.rank, "".join(traceback.format_exception_only(type(exc), exc)).strip()))
    res = comm.allgather(mine)
    bad = [r[1] for r in res if r[0] == "err"]
    if bad:
        raise RuntimeError("sharded generator: " + " | ".join(bad))
    return [r[1] for r in res]


# ---- local pieces -------------------------------------------------------------------------------------------------

def _local_base(verts, tets, rank, world):
    """this rank's base tets as a TetMesh in local numbering + its faces shared with other ranks (local ids, sorted)"""
    base = TetMesh(verts, tets)
    epart = rcb_partition(verts[tets].mean(axis=1), world)
    mine = np.nonzero(epart == rank)[0]
    if len(mine) == 0:
        raise ValueError("more ranks than base elements")
    lv = np.unique(tets[mine])                                   # ascending: local order = global order
    mesh = TetMesh(verts[lv].copy(), np.searchsorted(lv, tets[mine]))
    base.build_faces()
    tf = base.tet_faces
    lo = np.full(base.nf, world, dtype=np.int64); hi = np.full(base.nf, -1, dtype=np.int64)
    np.minimum.at(lo, tf.reshape(-1), np.repeat(epart, 4))
    np.maximum.at(hi, tf.reshape(-1), np.repeat(epart, 4))
    myf = np.unique(tf[mine])
    iface = np.searchsorted(lv, base.faces[myf[lo[myf] != hi[myf]]])
    return mesh, iface.reshape(-1, 3)


def _refine_surface(mesh: TetMesh, tri: np.ndarray) -> np.ndarray:
    """the four children of every triangle (sorted vertex triples of ``mesh``) in the numbering of ``mesh.refine()``"""
    if len(tri) == 0:
        return tri
    a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
    V = mesh.nv
    mab = V + mesh._edge_ids(np.stack([a, b], 1))
    mac = V + mesh._edge_ids(np.stack([a, c], 1))
    mbc = V + mesh._edge_ids(np.stack([b, c], 1))
    ch = np.concatenate([np.stack([a, mab, mac], 1), np.stack([mab, b, mbc], 1), np.stack([mac, mbc, c], 1), np.stack([mab, mbc, mac], 1)])
    return np.sort(ch, axis=1)


def _number_level(comm, mesh: TetMesh, iface: np.ndarray, numbering: str = "natural"):
    """block numbering of one level: (gid: local vertex -> global dof, owned: bool per local vertex, owner, starts, row_order: the owned
    local vertices in the order of their global numbers)"""
    def local():
        shared = np.unique(iface)
        return mesh.nv, shared, mesh.vertices[shared] + 0.0          # (+0.0: no negative zeros in the keys)
    msgs = agree(comm, local)
    W, r = comm.world, comm.rank
    counts = [len(m[1]) for m in msgs]
    K = np.ascontiguousarray(np.concatenate([m[2].reshape(-1, 3) for m in msgs])).view(np.int64)
    Rk = np.repeat(np.arange(W), counts)
    Ix = np.concatenate([np.arange(c) for c in counts]) if sum(counts) else np.empty(0, np.int64)
    order = np.lexsort((Rk, K[:, 2], K[:, 1], K[:, 0]))             # equal keys adjacent, lowest rank first
    Ks = K[order]
    new = np.ones(len(order), dtype=bool)
    new[1:] = np.any(Ks[1:] != Ks[:-1], axis=1)
    first = np.nonzero(new)[0]
    grp = np.cumsum(new) - 1
    size = np.diff(np.append(first, len(order)))
    own_rank = np.empty(len(order), np.int64); own_pos = np.empty(len(order), np.int64); gsize = np.empty(len(order), np.int64)
    own_rank[order] = Rk[order][first][grp]
    own_pos[order] = Ix[order][first][grp]
    gsize[order] = size[grp]
    off = np.concatenate([[0], np.cumsum(counts)])
    if np.any(gsize < 2):                                            # (every rank sees every list: all ranks raise together)
        raise RuntimeError("sharded generator: an interface vertex has no partner on another rank")
    # who owns what is now known everywhere; every rank orders its OWN vertices and publishes where its shared ones landed
    mine = slice(off[r], off[r + 1])
    shared, no = msgs[r][1], own_rank[mine] != r
    owned = np.ones(mesh.nv, dtype=bool)
    owned[shared[no]] = False
    if numbering == "tree":
        # by the refinement tree: a vertex goes with the first tet that holds it (children of tet i are 8 i .. 8 i + 7, so the tet number IS
        # a position in the tree) -- neighbours in the mesh end up close in memory at every scale, which is what the SpMV's gathers want
        key = np.full(mesh.nv, mesh.nt, dtype=np.int64)
        np.minimum.at(key, mesh.tets.reshape(-1), np.repeat(np.arange(mesh.nt, dtype=np.int64), 4))
    elif numbering == "natural":
        key = np.arange(mesh.nv, dtype=np.int64)                 # local order = the serial generator's order restricted to this rank
    else:
        raise ValueError("order must be 'natural' or 'tree'")
    own_ids = np.nonzero(owned)[0]
    row_order = own_ids[np.argsort(key[own_ids], kind="stable")]    # this rank's rows, in the order of their global numbers
    local_rank = np.full(mesh.nv, -1, dtype=np.int64)
    local_rank[row_order] = np.arange(len(row_order))
    placed = agree(comm, lambda: (len(row_order), local_rank[shared]))
    starts = np.concatenate([[0], np.cumsum([p[0] for p in placed])]).astype(np.int64)
    gid = np.where(owned, starts[r] + local_rank, -1)
    orr, opp = own_rank[mine][no], own_pos[mine][no]
    g = np.empty(len(orr), dtype=np.int64)
    for q in np.unique(orr):
        sel = orr == q
        g[sel] = starts[q] + placed[q][1][opp[sel]]
    if len(g) and g.min() < 0:
        raise RuntimeError("sharded generator: an owner did not number one of its shared vertices")
    gid[shared[no]] = g
    owner = np.full(mesh.nv, r, dtype=np.int64)
    owner[shared[no]] = orr
    return gid, owned, owner, starts, row_order


def _owned_rows(M: sp.csr_matrix, row_order: np.ndarray, gid_cols: np.ndarray, ncols_global: int) -> sp.csr_matrix:
    """rows of the owned local dofs in the order of their global numbers, columns in global numbering"""
    R = M[row_order].tocsr()
    out = sp.csr_matrix((R.data, gid_cols[R.indices].astype(np.int32 if ncols_global < 2**31 else np.int64), R.indptr), shape=(R.shape[0], ncols_global))
    out.has_sorted_indices = False
    out.sort_indices()
    return out


def sharded_h1_problem(comm, nref: int, nlevels: int | None = None, base=None, order: str = "natural"):
    """This rank's rows of ``meshgen.h1_problem(nref, nlevels)``.  ``order="natural"``: in the block numbering of
    ``partition.distributed_h1_hierarchy(..., comm.world)`` (a rank's dofs in the serial generator's order); ``order="tree"``: a rank's dofs
    ordered by the refinement tree instead (a symmetric permutation of the same problem inside every rank's block: same partition, same
    owners) -- the numbering a bandwidth-bound SpMV wants, and the one ``bench.py`` uses for its GPU-filling workload.

    Returns dict(A_rows, P_rows=[...], starts=[...], n_local_vertices, generate_local_s): ``A_rows`` = owned rows of the
    fine operator (global columns), ``P_rows[l]`` = owned rows of the interpolator level l+1 -> l, ``starts[l]`` = the row
    partition of level l.  ``base = (vertices, tets)`` replaces cube456 (tests)."""
    verts, tets = base if base is not None else load_cube456()
    state = {}

    def refine_all():
        mesh, iface = _local_base(np.asarray(verts, dtype=np.float64), np.asarray(tets, dtype=np.int64), comm.rank, comm.world)
        meshes, ifaces = [mesh], [iface]
        for _ in range(nref):
            meshes[-1].build_edges()
            ifaces.append(_refine_surface(meshes[-1], ifaces[-1]))
            meshes.append(meshes[-1].refine())
        meshes, ifaces = meshes[::-1], ifaces[::-1]
        if nlevels is not None:
            meshes, ifaces = meshes[:nlevels], ifaces[:nlevels]
        state["meshes"], state["ifaces"] = meshes, ifaces
        return len(meshes)
    nl = agree(comm, refine_all)[0]
    meshes, ifaces = state["meshes"], state["ifaces"]
    num = [_number_level(comm, meshes[l], ifaces[l], order) for l in range(nl)]
    starts = [n[3] for n in num]

    # fine operator: Neumann sub-assembly, essential rows / columns zeroed with the pattern kept (meshgen.eliminate_rowcol; the unit
    # diagonal of an essential row comes from its owner alone), rows of shared vertices travel to their owners
    def local_operator():
        fine = meshes[0]
        gid, owned, owner, st, _rows = num[0]
        S = fine.M0_P1_stiffness()
        mk = fine.boundary_marker_unit_cube()
        rows = np.repeat(np.arange(fine.nv), np.diff(S.indptr))
        cols = S.indices
        data = S.data.copy()
        kill = mk[rows] | mk[cols]
        data[kill] = 0.0
        data[kill & (rows == cols) & owned[rows]] = 1.0
        grow, gcol = gid[rows], gid[cols]
        keep = owned[rows]
        state["own"] = (grow[keep], gcol[keep], data[keep])
        out = {}
        away = ~keep
        if away.any():
            dest = owner[rows[away]]
            gr, gc, dv = grow[away], gcol[away], data[away]
            for q in np.unique(dest):
                sel = dest == q
                out[int(q)] = (gr[sel], gc[sel], dv[sel])
        return out
    sent = agree(comm, local_operator)

    def assemble():
        r = comm.rank
        st = starts[0]
        parts = [state.pop("own")] + [m[r] for m in sent if r in m]
        gr = np.concatenate([p[0] for p in parts]) - st[r]
        gc = np.concatenate([p[1] for p in parts])
        dv = np.concatenate([p[2] for p in parts])
        n_own, N = int(st[r + 1] - st[r]), int(st[-1])
        A = sp.coo_matrix((dv, (gr, gc)), shape=(n_own, N)).tocsr()
        A.sum_duplicates(); A.sort_indices()
        P_rows = []
        for l in range(nl - 1):
            f, c = meshes[l], meshes[l + 1]
            P = interp_P0(c, f)
            P.data[c.boundary_marker_unit_cube()[P.indices]] = 0.0           # meshgen.eliminate_cols
            P_rows.append(_owned_rows(P, num[l][4], num[l + 1][0], int(starts[l + 1][-1])))
        state["A"], state["P"] = A, P_rows
        return True
    agree(comm, assemble)
    return dict(A_rows=state["A"], P_rows=state["P"], starts=starts, n_local_vertices=[m.nv for m in meshes], rank=comm.rank, world=comm.world,
                gid=[n[0] for n in num], owned=[n[1] for n in num])


def parcsr_piece(comm, rows: sp.csr_matrix, row_starts, col_starts):
    """This rank's ParCSR piece (the dict ``partition.split_rows`` makes: diag / offd / col_map_offd and the comm package) from
    its rows in global column numbering; the send lists come from one all-gather of what every rank asks for."""
    r, W = comm.rank, comm.world
    col_starts = np.asarray(col_starts, dtype=np.int64)
    pieces = {}

    def split():
        loc = sp.csr_matrix(rows)
        loc.sort_indices()
        r0, c0, c1 = int(row_starts[r]), int(col_starts[r]), int(col_starts[r + 1])
        nr = loc.shape[0]
        cols = loc.indices.astype(np.int64)
        is_diag = (cols >= c0) & (cols < c1)
        rr = np.repeat(np.arange(nr), np.diff(loc.indptr))
        diag = sp.csr_matrix((loc.data[is_diag], (rr[is_diag], cols[is_diag] - c0)), shape=(nr, c1 - c0))
        gcols = cols[~is_diag]
        col_map = np.unique(gcols)
        offd = sp.csr_matrix((loc.data[~is_diag], (rr[~is_diag], np.searchsorted(col_map, gcols))), shape=(nr, len(col_map)))
        diag.sort_indices(); offd.sort_indices()
        owner = np.searchsorted(col_starts, col_map, side="right") - 1
        recv_procs = np.unique(owner)
        recv_starts = np.concatenate([[0], np.cumsum([(owner == q).sum() for q in recv_procs])]).astype(np.int32)
        pieces["p"] = dict(diag=diag, offd=offd, col_map_offd=col_map.astype(np.int64), first_row=r0, first_col=c0,
                           global_shape=(int(row_starts[-1]), int(col_starts[-1])), recv_procs=recv_procs.astype(np.int32), recv_starts=recv_starts)
        return {int(q): col_map[owner == q] for q in recv_procs}
    wants = agree(comm, split)
    p = pieces["p"]
    c0 = int(col_starts[r])
    procs, sstarts, elmts = [], [0], []
    for q in range(W):
        if q != r and r in wants[q] and len(wants[q][r]):
            procs.append(q)
            elmts.append(wants[q][r] - c0)
            sstarts.append(sstarts[-1] + len(wants[q][r]))
    p["send_procs"] = np.asarray(procs, dtype=np.int32)
    p["send_starts"] = np.asarray(sstarts, dtype=np.int32)
    p["send_elmts"] = (np.concatenate(elmts) if elmts else np.empty(0)).astype(np.int32)
    return p
