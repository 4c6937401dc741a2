"""Row-block domain decomposition of the benchmark operators into per-rank ParCSR pieces (host logic).

In ParElag the pieces come from ``mfem::ParMesh`` + ``SharingMap`` (``src/structures/SharingMap.cpp:975-1011``) and the
comm package from ``hypre_MatvecCommPkgCreate`` (call sites ``src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:
168-172``, ``SharingMap.cpp:110-111``).  Here every rank can see the whole synthetic problem, so the same lists are
computed directly: ``diag`` / ``offd`` split by the owner's column range, ``col_map_offd`` ascending, ghosts grouped by
owner (``recv_procs`` / ``recv_vec_starts``) and, symmetrically, the local indices each neighbour asked for
(``send_procs`` / ``send_map_starts`` / ``send_map_elmts``).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def canon(M) -> sp.csr_matrix:
    """CSR with truly ascending column indices and no duplicates (do not trust cached flags after fancy indexing)."""
    M = sp.csr_matrix(M)
    M.has_sorted_indices = False
    M.sort_indices()
    M.sum_duplicates()
    return M


def rcb_partition(points: np.ndarray, nparts: int) -> np.ndarray:
    """Recursive coordinate bisection of a point cloud into ``nparts`` (any integer >= 1) balanced parts."""
    part = np.zeros(len(points), dtype=np.int32)

    def rec(idx, first, count):
        if count == 1:
            part[idx] = first
            return
        left = count // 2
        p = points[idx]
        axis = int(np.argmax(p.max(axis=0) - p.min(axis=0)))
        order = np.argsort(p[:, axis], kind="stable")
        cut = int(round(len(idx) * left / count))
        rec(idx[order[:cut]], first, left)
        rec(idx[order[cut:]], first + left, count - left)
    rec(np.arange(len(points)), 0, nparts)
    return part


def element_partitions(seq, nparts: int):
    """Partition the coarsest (base) tets by RCB of their centroids; children inherit their parent's rank, so every
    derefinement agglomerate stays rank-local as ParElag requires (``CoarsenLocalPartitioning`` is per rank:
    ``src/topology/Topology.cpp:686-740``).  Returns one element->rank array per level (finest first)."""
    base = seq.meshes[-1]
    cent = base.vertices[base.tets].mean(axis=1)
    p = rcb_partition(cent, nparts)
    out = [p]
    for _ in range(len(seq.meshes) - 1):
        out.append(np.repeat(out[-1], 8))
    return out[::-1]


def vertex_owners(mesh, elem_part: np.ndarray) -> np.ndarray:
    """A vertex belongs to the lowest rank among the elements that contain it (the SharingMap master rule)."""
    owner = np.full(mesh.nv, np.iinfo(np.int32).max, dtype=np.int32)
    np.minimum.at(owner, mesh.tets.reshape(-1), np.repeat(elem_part, 4))
    return owner


def block_numbering(owner: np.ndarray, nparts: int):
    """Renumber dofs so that each rank owns a contiguous block (hypre's row partition).
    Returns (order: new->old, rank_of_old: old->new, starts: nparts+1)."""
    order = np.argsort(owner, kind="stable")
    new_of_old = np.empty(len(owner), dtype=np.int64)
    new_of_old[order] = np.arange(len(owner))
    starts = np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=nparts))]).astype(np.int64)
    return order, new_of_old, starts


def split_rows(A, row_starts, col_starts):
    """Split a global CSR (already in block numbering) into one ParCSR piece per rank, with comm packages.
    Returns a list of dicts understood by ``ParallelCSRMatrix.from_parts`` and by the oracle's ``OParCSR``."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    nparts = len(row_starts) - 1
    col_starts = np.asarray(col_starts, dtype=np.int64)
    parts = []
    for r in range(nparts):
        r0, r1 = int(row_starts[r]), int(row_starts[r + 1])
        c0, c1 = int(col_starts[r]), int(col_starts[r + 1])
        loc = A[r0:r1].tocsr()
        loc.sort_indices()
        cols = loc.indices
        is_diag = (cols >= c0) & (cols < c1)
        rows = np.repeat(np.arange(r1 - r0), np.diff(loc.indptr))
        diag = sp.csr_matrix((loc.data[is_diag], (rows[is_diag], cols[is_diag] - c0)), shape=(r1 - r0, c1 - c0))
        gcols = cols[~is_diag]
        col_map = np.unique(gcols)
        offd = sp.csr_matrix((loc.data[~is_diag], (rows[~is_diag], np.searchsorted(col_map, gcols))),
                             shape=(r1 - r0, len(col_map)))
        diag.sort_indices(); offd.sort_indices()
        # ghosts grouped by owner, ascending (col_map is ascending and owners own contiguous ranges)
        owner = np.searchsorted(col_starts, col_map, side="right") - 1
        recv_procs = np.unique(owner)
        recv_starts = np.concatenate([[0], np.cumsum([(owner == q).sum() for q in recv_procs])]).astype(np.int32)
        parts.append(dict(diag=diag, offd=offd, col_map_offd=col_map.astype(np.int64), first_row=r0, first_col=c0,
                          global_shape=A.shape, recv_procs=recv_procs.astype(np.int32), recv_starts=recv_starts,
                          _ghost_owner=owner))
    # send lists: what every neighbour asked for, in the neighbour's ghost order
    for r in range(nparts):
        c0 = int(col_starts[r])
        procs, starts, elmts = [], [0], []
        for q in range(nparts):
            if q == r:
                continue
            pq = parts[q]
            want = pq["col_map_offd"][pq["_ghost_owner"] == r]
            if len(want):
                procs.append(q)
                elmts.append(want - c0)
                starts.append(starts[-1] + len(want))
        parts[r]["send_procs"] = np.asarray(procs, dtype=np.int32)
        parts[r]["send_starts"] = np.asarray(starts, dtype=np.int32)
        parts[r]["send_elmts"] = (np.concatenate(elmts) if elmts else np.empty(0)).astype(np.int32)
    for p in parts:
        del p["_ghost_owner"]
    return parts


def distributed_h1_hierarchy(A, Ps, seq, nparts: int, galerkin: bool = True, split_levels: int | None = None):
    """Block-number every level of the 0-form hierarchy for ``nparts`` ranks and split all operators.

    ``galerkin``: also form the coarse operators in the new numbering with scipy (tests compare the device's distributed RAP
    with them); ``distributed.build_hierarchy`` only needs the fine operator and the interpolators and passes False.
    ``split_levels``: number of leading levels whose operators / interpolators are split into per-rank pieces (default: all).
    Returns dict(levels=[dict(A_parts, P_parts, starts, order)], A_global=[...], P_global=[...]) with P_parts[l] mapping
    level l -> l-1."""
    eparts = element_partitions(seq, nparts)
    nl = len(Ps) + 1
    ns = nl if split_levels is None else max(0, min(int(split_levels), nl))          # (0: numbering and permuted operators only, no per-rank pieces)
    orders, news, starts = [], [], []
    for l in range(nl):
        own = vertex_owners(seq.meshes[l], eparts[l])
        o, n, s = block_numbering(own, nparts)
        orders.append(o); news.append(n); starts.append(s)
    A_glob = [canon(sp.csr_matrix(A)[orders[0]][:, orders[0]])]
    P_glob = []
    for l, P in enumerate(Ps):
        Pp = canon(sp.csr_matrix(P)[orders[l]][:, orders[l + 1]])
        P_glob.append(Pp)
        if not galerkin:
            continue
        Ac = canon(Pp.T @ A_glob[-1] @ Pp)
        # hypre_ParCSRMatrixFixZeroRows: essential-BC rows of the Galerkin operator are empty -> unit diagonal
        zero = np.asarray(abs(Ac).sum(axis=1)).ravel() == 0
        Ac = canon(Ac + sp.diags(zero.astype(float)))
        A_glob.append(Ac)
    levels = []
    for l in range(nl):
        lev = dict(starts=starts[l], order=orders[l])
        if l < len(A_glob) and l < ns:
            lev["A_parts"] = split_rows(A_glob[l], starts[l], starts[l])
        if 0 < l <= ns:
            lev["P_parts"] = split_rows(P_glob[l - 1], starts[l - 1], starts[l])
        levels.append(lev)
    return dict(levels=levels, A_global=A_glob, P_global=P_glob)


def sharing_map_h1(mesh, elem_part: np.ndarray, nparts: int):
    """dof -> trueDof map of the 0-form dofs (``SharingMap::entity_trueEntity``, src/structures/SharingMap.hpp:266-287)
    for an element partition: rank r's *local* dofs are the vertices of its elements (shared vertices appear on every
    adjacent rank), its *true* dofs the vertices it owns (lowest adjacent rank).  Returns dict with
    ``parts`` (e_tE as per-rank ParCSR pieces: local dofs x true dofs, entries 1), ``local_vertices[r]`` (natural ids of
    rank r's local dofs), ``entity_starts``, ``true_starts``, ``order`` (true-dof block numbering, new -> natural) and
    the stacked selection matrix ``E`` (all local dofs x true dofs)."""
    own = vertex_owners(mesh, elem_part)
    order, new_of_old, true_starts = block_numbering(own, nparts)
    loc, rows, cols = [], [], []
    off = 0
    entity_starts = [0]
    for r in range(nparts):
        verts = np.unique(mesh.tets[elem_part == r].reshape(-1))
        verts = verts[np.argsort(new_of_old[verts], kind="stable")]      # local dofs in ascending true-dof order
        loc.append(verts)
        rows.append(off + np.arange(len(verts)))
        cols.append(new_of_old[verts])
        off += len(verts)
        entity_starts.append(off)
    E = sp.csr_matrix((np.ones(off), (np.concatenate(rows), np.concatenate(cols))), shape=(off, mesh.nv))
    entity_starts = np.asarray(entity_starts, dtype=np.int64)
    return dict(parts=split_rows(E, entity_starts, true_starts), local_vertices=loc, entity_starts=entity_starts,
                true_starts=true_starts, order=order, E=canon(E))


def local_stiffness(mesh, elem_part: np.ndarray, rank: int, local_vertices: np.ndarray) -> sp.csr_matrix:
    """P1 stiffness sub-assembled over rank ``rank``'s elements only, in its local dof numbering (the SerialCSRMatrix a
    ParElag rank holds before ``Assemble``)."""
    from .meshgen import TetMesh
    sub = TetMesh(mesh.vertices, mesh.tets[elem_part == rank])
    S = sub.M0_P1_stiffness()
    return canon(S[local_vertices][:, local_vertices])


def entity_owners(mesh, elem_part: np.ndarray, form: int) -> np.ndarray:
    """Owner rank of every dof of a lowest-order form (0 vertices, 1 edges, 2 faces, 3 elements): the lowest rank among
    the elements that contain the entity (the SharingMap master rule)."""
    if form == 3:
        return np.asarray(elem_part, dtype=np.int32).copy()
    if form == 0:
        return vertex_owners(mesh, elem_part)
    mesh.build_faces()
    conn = mesh.tet_edges if form == 1 else mesh.tet_faces
    n = mesh.ne if form == 1 else mesh.nf
    owner = np.full(n, np.iinfo(np.int32).max, dtype=np.int32)
    np.minimum.at(owner, conn.reshape(-1), np.repeat(elem_part, conn.shape[1]))
    return owner


def chain_element_owners(parts, coarsest_owner: np.ndarray):
    """Element -> rank on every level of an agglomeration chain from the ownership of the coarsest level's elements: an element
    belongs to the rank of its agglomerate (``parts[l]``: elements of level l -> elements of level l + 1), so every agglomerate
    is rank-local on every level, as ParElag requires of a rank's partitioning (``src/topology/Topology.cpp:686-740``).
    Returns one array per level, finest first (``len(parts) + 1`` of them)."""
    out = [np.asarray(coarsest_owner, dtype=np.int32)]
    for part in reversed(parts):
        out.append(out[-1][np.asarray(part, dtype=np.int64)])
    return out[::-1]


def hex_element_owners(n, parts, nparts: int):
    """Row-block decomposition of the Cartesian grid of ``examples/MultigridTestSPE10.cpp:194-208`` (``n[0] x n[1] x n[2]`` cells cut
    into 2 x 2 x 2 boxes level after level, ``amge_level.box_partition``): the boxes of the COARSEST level are divided among the
    ranks by recursive coordinate bisection of their (i, j, k) positions -- slabs across the longest direction first -- and every
    finer cell follows the box it lies in.  Returns the element -> rank arrays of all levels, finest first."""
    m = tuple(int(v) for v in n)
    for _ in parts:
        m = tuple(-(-v // 2) for v in m)
    I, J, K = np.meshgrid(np.arange(m[0]), np.arange(m[1]), np.arange(m[2]), indexing="ij")
    cent = np.stack([I.ravel(), J.ravel(), K.ravel()], axis=1).astype(np.float64) + 0.5
    if cent.shape[0] < nparts:
        raise ValueError("hex_element_owners: %d coarsest boxes cannot feed %d ranks (use fewer coarsenings or fewer ranks)" % (cent.shape[0], nparts))
    return chain_element_owners(parts, rcb_partition(cent, nparts))


def _darcy_tables(level):
    """(element -> facet ptr, idx, dofs per facet of form 2, element-interior dofs of form 2 per element, dofs of form 3 per
    element) of a ``SeqLevel`` (general sequence) or an ``AmgeLevel`` (two-form fast path): both number the facet dofs of form 2
    first, facet by facet, then the element-interior ones (``DofHandlerALG::computeOffset``, ``src/amge/DofHandler.cpp:1058-1186``)."""
    if hasattr(level, "ef_ptr"):
        return level.ef_ptr, level.ef_idx, np.diff(level.f_dof2_ptr), np.diff(level.e_idof2_ptr), np.diff(level.e_dof3_ptr)
    return level.B[0][0], level.B[0][1], np.diff(level.idof[2][1]), np.diff(level.idof[2][0]), np.diff(level.idof[3][0])


def darcy_dof_owners(level, elem_owner: np.ndarray):
    """Owner rank of every dof of forms 2 and 3 of a level (any element type, any number of dofs per entity): a dof living on
    a facet belongs to the lowest rank among the facet's elements (the ``SharingMap`` master rule,
    ``src/structures/SharingMap.cpp:975-1011``), a dof interior to an element to the element's rank."""
    ef_ptr, ef_idx, nf2, ne2, ne3 = _darcy_tables(level)
    elem_owner = np.asarray(elem_owner, dtype=np.int32)
    nF = nf2.size
    fown = np.full(nF, np.iinfo(np.int32).max, dtype=np.int32)
    np.minimum.at(fown, ef_idx, np.repeat(elem_owner, np.diff(ef_ptr)))
    if nF and fown.max() == np.iinfo(np.int32).max:
        raise ValueError("darcy_dof_owners: a facet belongs to no element")
    own2 = np.concatenate([np.repeat(fown, nf2), np.repeat(elem_owner, ne2)]).astype(np.int32)
    own3 = np.repeat(elem_owner, ne3).astype(np.int32)
    return own2, own3


def distributed_darcy_from_owners(M, B, Bt, Ps, owners, nparts: int):
    """The row-block pieces of the mixed system and its interpolators for given dof owners (``owners[l] = (own2, own3)`` per
    level, ``len(Ps) + 1`` levels): block numbering per space and level, every block and interpolator block split into ParCSR
    pieces with their comm packages.  Same return layout as ``distributed_darcy``."""
    order, starts = [], []
    for own2, own3 in owners:
        o2, _n2, s2 = block_numbering(own2, nparts)
        o3, _n3, s3 = block_numbering(own3, nparts)
        order.append((o2, o3)); starts.append((s2, s3))
    o2, o3 = order[0]
    s2, s3 = starts[0]
    glob = {(0, 0): canon(sp.csr_matrix(M)[o2][:, o2]), (0, 1): canon(sp.csr_matrix(Bt)[o2][:, o3]),
            (1, 0): canon(sp.csr_matrix(B)[o3][:, o2])}
    blocks = {(0, 0): split_rows(glob[(0, 0)], s2, s2), (0, 1): split_rows(glob[(0, 1)], s2, s3),
              (1, 0): split_rows(glob[(1, 0)], s3, s2)}
    P = []
    for l, (P2, P3) in enumerate(Ps):
        (f2, f3), (c2, c3) = order[l], order[l + 1]
        P.append([split_rows(canon(sp.csr_matrix(P2)[f2][:, c2]), starts[l][0], starts[l + 1][0]),
                  split_rows(canon(sp.csr_matrix(P3)[f3][:, c3]), starts[l][1], starts[l + 1][1])])
    return dict(blocks=blocks, P=P, order=order, starts=starts, global_blocks=glob)


def distributed_darcy_levels(dar, levels, elem_owners, nparts: int):
    """``distributed_darcy`` for a hierarchy given by its levels' own tables (``SeqLevel`` / ``AmgeLevel``: Cartesian hex grids,
    AMGe coarse spaces with several dofs per agglomerated entity) instead of the generator's tet meshes."""
    owners = [darcy_dof_owners(L, eo) for L, eo in zip(levels, elem_owners)]
    for l, (own2, own3) in enumerate(owners):
        n2 = dar["M"].shape[0] if l == 0 else dar["Ps"][l - 1][0].shape[1]
        n3 = dar["B"].shape[0] if l == 0 else dar["Ps"][l - 1][1].shape[1]
        if own2.size != n2 or own3.size != n3:
            raise ValueError("distributed_darcy_levels: level %d has %d / %d dofs but the operators have %d / %d" % (l, own2.size, own3.size, n2, n3))
    return distributed_darcy_from_owners(dar["M"], dar["B"], dar["Bt"], dar["Ps"], owners, nparts)


def distributed_darcy(dar, nparts: int):
    """Row-block decomposition of the mixed Darcy system of ``meshgen.darcy_problem`` for ``nparts`` ranks: faces (2-form)
    and elements (3-form) are block-numbered per level, every block and every interpolator block is split into ParCSR
    pieces.  Returns dict(blocks={(i, j): parts}, P=[[parts of P2_l, parts of P3_l]], order=[(faces, tets) per level],
    starts=[(faces, tets) per level])."""
    seq = dar["seq"]
    eparts = element_partitions(seq, nparts)
    owners = [(entity_owners(mesh, eparts[l], 2), entity_owners(mesh, eparts[l], 3)) for l, mesh in enumerate(seq.meshes)]
    return distributed_darcy_from_owners(dar["M"], dar["B"], dar["Bt"], dar["Ps"], owners[:len(dar["Ps"]) + 1], nparts)


def distributed_hdiv(A, Ps, Ds, seq, nparts: int):
    """Row-block decomposition of the 2-form problem of ``meshgen.hdiv_problem`` (faces owned by the lowest adjacent
    rank, edges likewise for the auxiliary space of the Hiptmair smoother): A (faces x faces), every P_l and every
    D_l (faces x edges) as per-rank ParCSR pieces.  Returns dict(A=parts, P=[parts], D=[parts], order2=[...], starts2=[...])."""
    eparts = element_partitions(seq, nparts)
    o2, s2, o1, s1 = [], [], [], []
    for l, mesh in enumerate(seq.meshes):
        a, _n, b = block_numbering(entity_owners(mesh, eparts[l], 2), nparts)
        o2.append(a); s2.append(b)
        a, _n, b = block_numbering(entity_owners(mesh, eparts[l], 1), nparts)
        o1.append(a); s1.append(b)
    Ag = canon(sp.csr_matrix(A)[o2[0]][:, o2[0]])
    out = dict(A=split_rows(Ag, s2[0], s2[0]), P=[], D=[], order2=o2, starts2=s2, A_global=Ag)
    for l, P in enumerate(Ps):
        out["P"].append(split_rows(canon(sp.csr_matrix(P)[o2[l]][:, o2[l + 1]]), s2[l], s2[l + 1]))
    for l, D in enumerate(Ds[:len(Ps)]):
        out["D"].append(split_rows(canon(sp.csr_matrix(D)[o2[l]][:, o1[l]]), s2[l], s1[l]))
    return out
