"""``DeRhamSequence::Coarsen`` on more than one rank (one process per GPU).

The reference distributes the mesh: every rank coarsens its own elements, an agglomerated entity shared by two ranks is
computed by its owner and sent to the other (``src/amge/DeRhamSequence.cpp:1820, 1930, 1971, 2024``: ``AE_TrueAE.IsShared``,
``Synchronize``), and the true-dof numbering comes from the ``SharingMap`` of every codimension.  Here the split is made for a
GPU with much memory and cheap arithmetic:

* the **integer topology** of the level (``coarsen_topology``: a few int arrays per codimension) is replicated -- every rank
  builds it for the whole level, so the numbering of the coarse entities is global by construction;
* the **numeric stages** are sharded by agglomerate ownership.  A rank restricts the level to the closure of its own
  agglomerates (their elements, and the facets / ridges / peaks of those), keeping the global relative order of entities and
  dofs, and runs the unchanged single-rank ``coarsen`` on that sub-level with the restricted global topology.  An
  agglomerated facet / ridge / peak on the interface is therefore computed by every rank that touches it, from the same
  numbers in the same order: the ranks agree on its coarse basis (to the last bit when their launches use the same workgroup
  size, to rounding otherwise) without a message -- redundant interface work instead of owner-computes-and-sends;
* one **exchange of integers** fixes the global coarse dof numbering: the number of coarse dofs of every coarse entity
  (``DofHandlerALG``'s prefix sums) -- a MAX all-reduce over arrays of zeros and counts (``exchange_counts``: RCCL or gloo
  through ``torch.distributed``, or an in-process stand-in for tests).

What a rank ends with (``RankPiece``): its rows of every ``P_j`` (fine dofs of its closure, global coarse columns), the coarse
``D_j`` rows, coarse element / facet / ridge matrices, projected targets and cochain-projector pieces of its coarse entities
-- in global numbers, which is what ``distributed.py`` / ``SharingMap`` assembly consumes.  ``merge_pieces`` puts the pieces of
all ranks together (tests, or a gather on one rank) and must reproduce the single-rank result.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .amge import expand_ranges
from .amge_level import ptr_from_counts
from .derham_level import SeqLevel


def agglomerate_owner(n_agglomerates: int, nranks: int) -> np.ndarray:
    """Default ownership: contiguous blocks of agglomerate numbers (what a row-block partition of the coarse elements gives)."""
    return (np.arange(n_agglomerates, dtype=np.int64) * nranks) // max(n_agglomerates, 1)


def _distinct(ids, n):
    """The distinct values of ``ids`` (all in 0 .. n-1), ascending: a marker array instead of a sort."""
    mark = np.zeros(n, dtype=bool)
    mark[ids] = True
    return np.flatnonzero(mark).astype(np.int64)


def _restrict_rows(ptr, idx, rows, col_map, sgn=None):
    q, _own = expand_ranges(ptr[rows], np.diff(ptr)[rows])
    cols = col_map[idx[q]]
    if cols.size and cols.min() < 0:
        raise ValueError("restriction: a selected entity refers to an entity outside the closure")
    out = (ptr_from_counts(np.diff(ptr)[rows]), cols.astype(np.int64))
    return out + ((sgn[q],) if sgn is not None else ())


def _dof_list(idof, base_of, sel, cb):
    """Global dof numbers of a form on the selected entities, in the level's own order (highest codimension first)."""
    parts = []
    for c in range(cb, -1, -1):
        ip = idof[c]
        d, _ = expand_ranges(base_of(c) + ip[sel[c]], np.diff(ip)[sel[c]])
        parts.append(d)
    return np.concatenate(parts).astype(np.int64) if parts else np.zeros(0, dtype=np.int64)


@dataclass
class Restriction:
    level: SeqLevel                 # the sub-level
    topo: dict                      # the global agglomerated topology restricted to it
    sel: list                       # sel[c]: global fine entity numbers kept (ascending), c = 0 .. 3
    csel: list                      # csel[c]: global coarse entity numbers kept (ascending)
    dofs: list                      # dofs[j]: global fine dof numbers of form j in the sub-level's order


def restrict_level(lv: SeqLevel, topo: dict, agglomerates: np.ndarray) -> Restriction:
    """The sub-level spanned by the given agglomerates of ``topo`` and the matching part of ``topo``."""
    agglomerates = np.unique(np.asarray(agglomerates, dtype=np.int64))
    js = lv.jstart
    if agglomerates.size == topo["n"][0]:                       # every agglomerate (one rank): the level itself, numbering unchanged
        ident = lambda n: np.arange(n, dtype=np.int64)
        return Restriction(level=lv, topo=topo, sel=[ident(lv.n[c]) for c in range(4)], csel=[ident(topo["n"][c]) for c in range(4)],
                           dofs=[None if j < js else ident(lv.ndof(j)) for j in range(4)])
    # ---- fine and coarse entities of the closure, ascending ----
    ap, ai, _ = topo["A"][0]
    q, _ = expand_ranges(ap[agglomerates], np.diff(ap)[agglomerates])
    sel = [_distinct(ai[q], lv.n[0])]
    csel = [agglomerates]
    for c in range(3):
        p, i, _s = lv.B[c]
        q, _ = expand_ranges(p[sel[c]], np.diff(p)[sel[c]])
        sel.append(_distinct(i[q], lv.n[c + 1]))
        p, i, _s = topo["B"][c]
        q, _ = expand_ranges(p[csel[c]], np.diff(p)[csel[c]])
        csel.append(_distinct(i[q], topo["n"][c + 1]))
    g2l = []
    for c in range(4):
        m = np.full(lv.n[c], -1, dtype=np.int64); m[sel[c]] = np.arange(sel[c].size); g2l.append(m)
    # ---- the sub-level ----
    B = [_restrict_rows(*lv.B[c][:2], sel[c], g2l[c + 1], lv.B[c][2]) for c in range(3)]
    idof, M, dofs = [None] * 4, [None] * 4, [None] * 4
    for j in range(js, 4):
        cb = lv.cb(j)
        idof[j] = [ptr_from_counts(np.diff(lv.idof[j][c])[sel[c]]) for c in range(cb + 1)]
        dofs[j] = _dof_list(lv.idof[j], lambda c, j=j: lv.base(j, c), sel, cb)
        M[j] = []
        for c in range(cb + 1):
            ptr, vals = lv.M[j][c]
            q, _ = expand_ranges(ptr[sel[c]], np.diff(ptr)[sel[c]])
            M[j].append((ptr_from_counts(np.diff(ptr)[sel[c]]), vals[q]))
    D = [None] * 3
    for j in range(js, 3):
        D[j] = lv.D[j].tocsr()[dofs[j + 1]][:, dofs[j]].tocsr()
        D[j].sort_indices()
    sub = SeqLevel(n=[int(s.size) for s in sel], B=B, facet_attr=lv.facet_attr[sel[1]], jstart=js, idof=idof, M=M, D=D,
                   T=[None if j < js else lv.T[j][dofs[j]] for j in range(4)], pv=[None if j < js else lv.pv[j][dofs[j]] for j in range(4)],
                   const3=lv.const3[dofs[3]], meta=dict(lv.meta))
    return Restriction(level=sub, topo=_restrict_topology(topo, csel, g2l), sel=sel, csel=csel, dofs=dofs)


def _coarse_closure(topo: dict, agglomerates: np.ndarray) -> list:
    csel = [np.unique(np.asarray(agglomerates, dtype=np.int64))]
    for c in range(3):
        p, i, _s = topo["B"][c]
        q, _ = expand_ranges(p[csel[c]], np.diff(p)[csel[c]])
        csel.append(_distinct(i[q], topo["n"][c + 1]))
    return csel


def _restrict_topology(topo: dict, csel: list, g2l: list) -> dict:
    """The agglomerated topology restricted to the coarse entities ``csel`` with the fine entities renumbered by ``g2l``."""
    cg2l = []
    for c in range(4):
        m = np.full(topo["n"][c], -1, dtype=np.int64); m[csel[c]] = np.arange(csel[c].size); cg2l.append(m)
    A = [_restrict_rows(*topo["A"][c][:2], csel[c], g2l[c], topo["A"][c][2]) for c in range(4)]
    Bc = [_restrict_rows(*topo["B"][c][:2], csel[c], cg2l[c + 1], topo["B"][c][2]) for c in range(3)]
    return dict(A=A, B=Bc, facet_attr=topo["facet_attr"][csel[1]], n=[int(s.size) for s in csel])


# ---- exchange of the coarse dof counts -----------------------------------------------------------------------------------------

def exchange_counts(local: np.ndarray, group=None) -> np.ndarray:
    """MAX all-reduce of an int64 array over the ranks (``torch.distributed``: gloo on host tensors, RCCL on device tensors).
    Entities a rank does not hold contribute zeros; ranks holding the same entity contribute the same count."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("exchange_counts: torch.distributed is not initialised (pass exchange= for an in-process run)")
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.int64))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t.cpu().numpy()


@dataclass
class RankPiece:
    """One rank's share of the coarsening, in global numbers."""
    rank: int
    agglomerates: np.ndarray
    fine_dofs: list                  # [j]: global fine dofs of the rows of P[j]
    coarse_dofs: list                # [j]: global coarse dofs of this rank's coarse entities (rows of D, T, ...; columns of P)
    P: list                          # [j]: csr (len(fine_dofs[j]) x n_coarse_global[j])
    D: list                          # [j]: csr (len(coarse_dofs[j+1]) x n_coarse_global[j])
    T: list                          # [j]: rows coarse_dofs[j] of the projected targets
    const3: np.ndarray
    csel: list                       # [c]: global coarse entities of this rank
    counts: dict                     # (j, c) -> global count array of coarse dofs per coarse entity
    blocks: dict                     # (j, c) -> (ptr, vals): coarse mass blocks of the entities csel[c]
    n_null: dict
    local: dict = field(default_factory=dict)     # the untouched single-rank result on the sub-level


def coarsen_rank(lv: SeqLevel, part: np.ndarray, rank: int, nranks: int, owner: np.ndarray | None = None, topo: dict | None = None,
                 exchange=exchange_counts, coarsen_fn=None, held: "RankPiece | None" = None, **kw) -> RankPiece:
    """This rank's part of ``Coarsen``: global topology, restriction, the single-rank numeric path on the sub-level, the
    exchange of the coarse dof counts, the renumbering to global dofs.  ``coarsen_fn`` defaults to the device path
    (``derham.coarsen``); ``exchange(local_counts) -> global_counts`` defaults to the ``torch.distributed`` all-reduce.

    ``held``: the piece this rank got from coarsening the level above.  ``lv`` then only needs the integer part of the level
    (``n``, ``B``, ``facet_attr``, ``jstart``, ``idof`` sizes are not read): the numeric sub-level is the coarse level the rank
    already holds (``held.local["coarse"]``, entities ``held.csel``, dofs ``held.coarse_dofs``), which is the closure of its
    agglomerates when ownership is nested (an agglomerate of this level owns only agglomerates of the level above that the
    same rank owns) -- the numeric data of a coarse level never exists in one place."""
    from . import derham
    coarsen_fn = derham.coarsen if coarsen_fn is None else coarsen_fn
    topo = derham.coarsen_topology(lv, part) if topo is None else topo
    owner = agglomerate_owner(topo["n"][0], nranks) if owner is None else np.asarray(owner)
    mine = np.nonzero(owner == rank)[0]
    js = lv.jstart
    keys = [(j, c) for j in range(js, 4) for c in range(lv.cb(j) + 1)]
    offs = np.concatenate([[0], np.cumsum([topo["n"][c] for _j, c in keys])])
    flat = np.zeros(int(offs[-1]), dtype=np.int64)
    NOT_HELD = np.int64(-(1 << 40))

    def exchange_checked(flat, heldmask):
        """one MAX all-reduce that carries the counts and, negated, their minimum over the ranks that hold an entity: every
        rank sees the same (max, min) pair per entity, so a disagreement raises on ALL ranks at once and nobody is left
        waiting in the next collective"""
        both = exchange(np.concatenate([flat, np.where(heldmask, -flat, NOT_HELD)]))
        gmax, gmin = both[:flat.size], -both[flat.size:]
        bad = np.nonzero((gmin < -NOT_HELD) & (gmax != gmin))[0]
        if bad.size:
            raise RuntimeError("coarsen_rank: ranks disagree on the number of coarse dofs of %d shared entit%s (first: flat index %d, "
                               "counts between %d and %d)" % (bad.size, "y" if bad.size == 1 else "ies", int(bad[0]), int(gmin[bad[0]]), int(gmax[bad[0]])))
        return gmax
    if mine.size == 0:
        # a rank without agglomerates (more ranks than agglomerates on a coarse level) still takes part in the exchange
        glob = exchange_checked(flat, np.zeros(flat.size, dtype=bool))
        counts = {key: glob[offs[k]:offs[k + 1]] for k, key in enumerate(keys)}
        nco = {j: int(sum(counts[(j, c)].sum() for c in range(lv.cb(j) + 1))) for j in range(js, 4)}
        e64 = np.zeros(0, dtype=np.int64)
        return RankPiece(rank=rank, agglomerates=mine, fine_dofs=[None if j < js else e64 for j in range(4)],
                         coarse_dofs=[None if j < js else e64 for j in range(4)],
                         P=[None if j < js else sp.csr_matrix((0, nco[j])) for j in range(4)],
                         D=[None if j < js else sp.csr_matrix((0, nco[j])) for j in range(3)],
                         T=[None if j < js else np.zeros((0, (lv.T[j] if held is None else held.T[j]).shape[1])) for j in range(4)], const3=np.zeros(0),
                         csel=[e64] * 4,
                         counts=counts, blocks={key: (np.zeros(1, dtype=np.int64), np.zeros(0)) for key in keys}, n_null={}, local={})
    if held is None:
        R = restrict_level(lv, topo, mine)
    else:
        csel = _coarse_closure(topo, mine)
        g2l = []
        for c in range(4):
            m = np.full(lv.n[c], -1, dtype=np.int64); m[held.csel[c]] = np.arange(held.csel[c].size); g2l.append(m)
        ap, ai, _ = topo["A"][0]
        q, _ = expand_ranges(ap[mine], np.diff(ap)[mine])
        if not np.array_equal(np.sort(ai[q]), held.csel[0]):
            raise ValueError("coarsen_rank: ownership is not nested -- the elements of this rank's agglomerates are not the coarse "
                             "elements it holds from the level above")
        R = Restriction(level=held.local["coarse"], topo=_restrict_topology(topo, csel, g2l), sel=held.csel, csel=csel, dofs=held.coarse_dofs)
    res = coarsen_fn(R.level, None, topo=R.topo, **kw)
    cl = res["coarse"]
    # ---- global numbering of the coarse dofs: counts per coarse entity, exchanged ----
    heldmask = np.zeros(flat.size, dtype=bool)
    for k, (j, c) in enumerate(keys):
        flat[offs[k] + R.csel[c]] = np.diff(cl.idof[j][c])
        heldmask[offs[k] + R.csel[c]] = True
    glob = exchange_checked(flat, heldmask)
    counts = {key: glob[offs[k]:offs[k + 1]] for k, key in enumerate(keys)}
    cidof = {j: [ptr_from_counts(counts[(j, c)]) for c in range(lv.cb(j) + 1)] for j in range(js, 4)}
    ncoarse = {j: int(sum(cidof[j][c][-1] for c in range(lv.cb(j) + 1))) for j in range(js, 4)}

    def cbase(j, c):
        return int(sum(cidof[j][c2][-1] for c2 in range(lv.cb(j), c, -1)))
    cdofs = [None] * 4
    for j in range(js, 4):
        cdofs[j] = _dof_list(cidof[j], lambda c, j=j: cbase(j, c), R.csel, lv.cb(j))
        assert cdofs[j].size == cl.ndof(j)
    # ---- results in global numbers ----
    P, Dg, T = [None] * 4, [None] * 3, [None] * 4
    for j in range(js, 4):
        Pl = res["P"][j].tocsr()
        P[j] = sp.csr_matrix((Pl.data.copy(), cdofs[j][Pl.indices], Pl.indptr.copy()), shape=(Pl.shape[0], ncoarse[j]))   # (sorted in place below)
        P[j].sort_indices()
        T[j] = cl.T[j]
        if j < 3:
            Dl = cl.D[j].tocsr()
            Dg[j] = sp.csr_matrix((Dl.data.copy(), cdofs[j][Dl.indices], Dl.indptr.copy()), shape=(Dl.shape[0], ncoarse[j]))
            Dg[j].sort_indices()
    blocks = {(j, c): cl.M[j][c] for j, c in keys}
    return RankPiece(rank=rank, agglomerates=mine, fine_dofs=R.dofs, coarse_dofs=cdofs, P=P, D=Dg, T=T, const3=cl.const3, csel=R.csel,
                     counts=counts, blocks=blocks, n_null=res["n_null"], local=res)


def integer_level(topo: dict, jstart: int) -> SeqLevel:
    """The integer part of the coarse level an agglomerated topology defines (entity counts, boundary tables, facet attributes):
    all that ``coarsen_topology`` reads, and all of a coarse level that is replicated on every rank."""
    return SeqLevel(n=list(topo["n"]), B=topo["B"], facet_attr=topo["facet_attr"], jstart=jstart, idof=[None] * 4, M=[None] * 4, D=[None] * 3,
                    T=[None] * 4, pv=[None] * 4, const3=np.zeros(0), meta=dict(kind="integer"))


def coarsen_allgather(lv: SeqLevel, part: np.ndarray, rank: int, nranks: int, owner: np.ndarray | None = None, group=None,
                      coarsen_fn=None, **kw) -> dict:
    """``Coarsen`` over the ranks of ``torch.distributed`` with the result complete on every rank: each rank computes its
    piece, the pieces are all-gathered (they are a 1 / nranks share of the result plus the interface) and merged.  This is
    the form the row-block builders of ``distributed.py`` take today (they split global operators); a consumer that keeps
    the pieces distributed calls ``coarsen_rank`` directly."""
    import dataclasses
    import torch.distributed as dist
    from . import derham
    topo = derham.coarsen_topology(lv, part)
    piece = coarsen_rank(lv, part, rank, nranks, owner=owner, topo=topo, exchange=lambda a: exchange_counts(a, group), coarsen_fn=coarsen_fn, **kw)
    pieces = [None] * nranks
    dist.all_gather_object(pieces, dataclasses.replace(piece, local={}), group=group)
    out = merge_pieces(lv, topo, pieces)
    out["piece"] = piece
    return out


def merge_pieces(lv: SeqLevel, topo: dict, pieces: list, atol: float = 0.0) -> dict:
    """All ranks' pieces as one global result (``P``, coarse ``SeqLevel``): what a gather on one rank would assemble.  Rows /
    entities held by several ranks are taken from the lowest rank; with ``atol`` the copies are compared first."""
    js = lv.jstart
    counts = pieces[0].counts
    idof = [None] * 4
    for j in range(js, 4):
        idof[j] = [ptr_from_counts(counts[(j, c)]) for c in range(lv.cb(j) + 1)]
    nco = {j: int(sum(idof[j][c][-1] for c in range(lv.cb(j) + 1))) for j in range(js, 4)}
    P, D, T, M = {}, [None] * 3, [None] * 4, [None] * 4
    const3 = np.zeros(nco[3])

    def stack_rows(nrows, ncols, items):
        """items: (global rows, csr) per rank; first writer wins.  The rows are copied straight into the CSR arrays of the result."""
        taken = np.zeros(nrows, dtype=bool)
        cnt = np.zeros(nrows, dtype=np.int64)
        chosen = []
        for rows, mat in items:
            mat = mat.tocsr()
            new = ~taken[rows]
            if atol and np.any(~new) and chosen:
                cur = assemble(nrows, ncols, cnt, chosen)
                d = abs(cur[rows[~new]] - mat[np.nonzero(~new)[0]])
                if d.nnz and d.max() > atol:
                    raise RuntimeError(f"merge_pieces: ranks disagree on shared rows by {d.max():.3e}")
            loc = np.flatnonzero(new)
            cnt[rows[loc]] = np.diff(mat.indptr)[loc]
            chosen.append((rows[loc], mat, loc))
            taken[rows] = True
        if not np.all(taken):
            raise RuntimeError("merge_pieces: some rows are held by no rank")
        return assemble(nrows, ncols, cnt, chosen)

    def assemble(nrows, ncols, cnt, chosen):
        indptr = ptr_from_counts(cnt)
        indices, data = np.empty(int(indptr[-1]), dtype=np.int64), np.empty(int(indptr[-1]))
        for grows, mat, loc in chosen:
            n_r = np.diff(mat.indptr)[loc].astype(np.int64)
            src, _ = expand_ranges(mat.indptr[loc].astype(np.int64), n_r)
            dst, _ = expand_ranges(indptr[grows], n_r)
            indices[dst] = mat.indices[src]; data[dst] = mat.data[src]
        out = sp.csr_matrix((data, indices, indptr), shape=(nrows, ncols))
        out.sort_indices()                                          # (a pass over the pattern; the pieces' rows come sorted)
        return out
    for j in range(js, 4):
        P[j] = stack_rows(lv.ndof(j), nco[j], [(p.fine_dofs[j], p.P[j]) for p in pieces])
        T[j] = np.zeros((nco[j], lv.T[j].shape[1]))
        for p in reversed(pieces):
            T[j][p.coarse_dofs[j]] = p.T[j]
        if j < 3:
            D[j] = stack_rows(nco[j + 1], nco[j], [(p.coarse_dofs[j + 1], p.D[j]) for p in pieces])
        M[j] = []
        for c in range(lv.cb(j) + 1):
            # block sizes: dofs of the entity's sub-entities + its own -- known only to a rank that holds it
            size = np.zeros(topo["n"][c], dtype=np.int64)
            for p in pieces:
                size[p.csel[c]] = np.diff(p.blocks[(j, c)][0])
            ptr = ptr_from_counts(size)
            vals = np.zeros(int(ptr[-1]))
            for p in reversed(pieces):
                q, _ = expand_ranges(ptr[p.csel[c]], size[p.csel[c]])
                vals[q] = p.blocks[(j, c)][1]
            M[j].append((ptr, vals))
    for p in reversed(pieces):
        const3[p.coarse_dofs[3]] = p.const3
    coarse = SeqLevel(n=topo["n"], B=topo["B"], facet_attr=topo["facet_attr"], jstart=js, idof=idof, M=M, D=D, T=T, pv=[None] * 4,
                      const3=const3, meta=dict(kind="alg"))
    for j in range(js, 4):
        vv = np.zeros(coarse.ndof(j)); ctr = coarse.cb(j)
        p_ = coarse.idof[j][ctr]
        vv[coarse.base(j, ctr) + p_[:-1][np.diff(p_) > 0]] = 1.0
        coarse.pv[j] = vv
    return dict(P=P, coarse=coarse, topo=topo, n_null={j: None for j in range(js, 4)})
