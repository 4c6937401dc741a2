"""Host side of the full AMGe coarse-space construction: ``DeRhamSequence::Coarsen`` (``src/amge/DeRhamSequence.cpp:572-692``)
for all forms ``jformStart_ .. 3`` of a 3-D sequence.

Same split as ``amge.py`` (the two-form special case): the *integer tables* -- agglomerated topology of every codimension
(``AgglomeratedTopology::CoarsenLocalPartitioning``, ``src/topology/Topology.cpp:686-828``), dof agglomeration
(``src/amge/DOFAgglomeration.cpp:33-314``), coarse dof numbering (``DofHandlerALG``) -- are built here with vectorised numpy;
every floating-point operation of the per-agglomerate loops runs on the device: ``ComputeCoarseTracesWithTargets``
(``:1818-1926``) through ``elag_hip_coarsen_traces``; ``hFacetExtension`` (``:2364-2556``) and ``hRidgePeakExtension``
(``:2762-3033``) through ``elag_hip_coarsen_extension`` in mode 0 / 1; the product ``P_{j+1} D_c`` (``:2696``) and the
projection of the targets through the device SpGEMM / SpMV.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from .amge import (EPS, _Assembly, _ExtIn, _ExtOut, _Keep, _TracesIn, _TracesOut, _assembly, _blocks_to_csr, _blocks_to_csr_sets, _join_cols,
                   _pack_cols, _ptr, _spmm_device, _take_leading, _take_leading_rows, expand_ranges)
from .amge_level import ptr_from_counts
from .derham_level import SeqLevel, _Tables, _concat_rows, _product


class _Lookup:
    """position of (row, id) pairs inside ragged rows (first occurrence); native host code (``csrc/host_tables.hip``)"""

    def __init__(self, ptr, idx, nid):
        self.ptr = np.ascontiguousarray(ptr, dtype=np.int64)
        self.idx = np.ascontiguousarray(idx, dtype=np.int64)
        self.nid = int(nid)

    def find(self, row, ids):
        row = np.ascontiguousarray(row, dtype=np.int64)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        pos = np.empty(row.size, dtype=np.int64)
        _lib.check(_lib.load().elag_hip_host_lookup_find(len(self.ptr) - 1, _ptr(self.ptr), _ptr(self.idx), self.nid, row.size, _ptr(row), _ptr(ids), _ptr(pos)))
        return pos


def _gather_dense(M: sp.csr_matrix, row_ptr, rows, col_lookup: _Lookup, ncol):
    """Per entity the dense sub-block M[rows of the entity, columns of the entity] packed col-major (nrows_e x ncol_e)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    ncol = np.ascontiguousarray(np.broadcast_to(ncol, (len(row_ptr) - 1,)), dtype=np.int64)
    out_ptr = ptr_from_counts(np.diff(row_ptr) * ncol)
    out = np.zeros(int(out_ptr[-1]))
    if M.nnz > np.iinfo(np.int32).max:
        raise ValueError("_gather_dense: matrix too large for 32-bit CSR indices")
    indptr, indices = np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32)
    data = np.ascontiguousarray(M.data, dtype=np.float64)
    _lib.check(_lib.load().elag_hip_host_gather_dense(len(row_ptr) - 1, _ptr(row_ptr), _ptr(rows), _ptr(col_lookup.ptr), _ptr(col_lookup.idx), _ptr(ncol),
                                                  M.shape[0], M.shape[1], _ptr(indptr), _ptr(indices), _ptr(data), _ptr(out_ptr), _ptr(out)))
    return out


# ---- topology ----------------------------------------------------------------------------------------------------------------

def _deagglomerate(A, bad):
    """``DeAgglomerateBadAgglomeratedEntities`` (``Topology.cpp:1151-1214``): every member of a bad agglomerated entity becomes
    an agglomerated entity of its own, in place (the later ones move up), orientations kept."""
    ptr, idx, sgn = A
    first = np.zeros(idx.size + 1, dtype=bool)
    first[ptr] = True
    row = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    first[:-1] |= bad[row]
    return np.nonzero(first)[0].astype(np.int64), idx, sgn


def coarsen_topology(lv: SeqLevel, part: np.ndarray, check: bool = False) -> dict:
    """``CoarsenLocalPartitioning`` for all codimensions.  The minimal intersection sets
    (``src/structures/minimalIntersectionSet.cpp:43-132``) are the classes of the relation ``Z_ii = Z_jj = |Z_ij|``, i.e. of equal
    adjacency signatures (see below), numbered by their lowest member and oriented like it.  With ``check`` (the reference's
    ``check_topology`` argument, ``Topology.cpp:727-738, 778, 803``) the agglomerated entities of every codimension are tested as
    soon as they exist (``check_topology`` below) and the bad ones are de-agglomerated before the next codimension is built;
    the diagnostics are returned under ``"messages"``."""
    part = np.asarray(part, dtype=np.int64)
    nA = int(part.max()) + 1
    order = np.argsort(part, kind="stable")
    A = [(ptr_from_counts(np.bincount(part, minlength=nA)), order.astype(np.int64), np.ones(order.size))]
    messages, nfixed = [], [0, 0, 0]

    def fix(Ac, c):
        _b, bad, msg = _check_codim(lv, Ac, c)
        messages.extend(msg)
        nfixed[c] = int(bad.sum())
        return _deagglomerate(Ac, bad) if bad.any() else Ac
    if check:
        A[0] = fix(A[0], 0)
    Bc, attr_c = [], None
    for c in range(3):
        ap, ai, asg = A[c]
        fp, fi, fs = _product(ap, ai, lv.B[c][0], lv.B[c][1], asg, lv.B[c][2])          # AE_fc
        nsub = lv.n[c + 1]
        # minimal intersection sets from the adjacency signatures of the sub-entities on agglomerate boundaries: their
        # agglomerates with the signs relative to the first one -- or the absolute signs and the attribute for boundary facets
        # with an attribute, whose term in Z cannot change sign.  Equal signatures <=> Z_ii = Z_jj = |Z_ij|, without forming Z
        # (which couples all the sub-entities of an agglomerate with each other: 10^8 entries on a 10^6-cell mesh).  Native host
        # code (csrc/host_tables.hip).
        fpc, fic, fsc = np.ascontiguousarray(fp, dtype=np.int64), np.ascontiguousarray(fi, dtype=np.int64), np.ascontiguousarray(fs, dtype=np.float64)
        attr32 = np.ascontiguousarray(lv.facet_attr, dtype=np.int32) if c == 0 else None
        mis, ori, nset = np.empty(nsub, dtype=np.int64), np.empty(nsub), np.zeros(1, dtype=np.int64)
        _lib.check(_lib.load().elag_hip_host_minimal_intersection_sets(len(fpc) - 1, _ptr(fpc), _ptr(fic), _ptr(fsc), nsub,
                                                                       None if attr32 is None else _ptr(attr32), _ptr(mis), _ptr(ori), _ptr(nset)))
        nmis = int(nset[0])
        member = mis >= 0
        mem = np.nonzero(member)[0]
        o2 = np.argsort(mis[mem], kind="stable")
        Anew = (ptr_from_counts(np.bincount(mis[mem], minlength=nmis)), mem[o2].astype(np.int64), ori[mem[o2]])
        if check and c + 1 <= 2:
            Afix = fix(Anew, c + 1)
            if Afix is not Anew:
                Anew = Afix
                mis[Anew[1]] = np.repeat(np.arange(len(Anew[0]) - 1, dtype=np.int64), np.diff(Anew[0]))
        A.append(Anew)
        sub_ptr = ptr_from_counts(member.astype(np.int64))                   # fc_AF: one entry per member row
        bp, bi, bs = _product(fp, fi, sub_ptr, mis[mem], fs, ori[mem])
        Bc.append((bp, bi.astype(np.int64), bs))
        if c == 0:
            attr_c = lv.facet_attr[A[1][1][A[1][0][:-1]]].astype(np.int32)
    out = dict(A=A, B=Bc, facet_attr=attr_c, n=[len(a[0]) - 1 for a in A])
    if check:
        out["messages"], out["n_deagglomerated"] = messages, nfixed
    return out


# ---- dof tables ---------------------------------------------------------------------------------------------------------------

def agglomerate_dofs(lv: SeqLevel, tb: _Tables, topo: dict, j: int) -> dict:
    """``DofAgglomeration`` of form j for every codimension: rows (ptr, dofs), interior counts, separator types."""
    cb = lv.cb(j)
    nd = lv.ndof(j)
    rows = {}
    for c in range(cb + 1):
        ep, ei = tb.edof(j, c)
        p, d, _ = _product(topo["A"][c][0], topo["A"][c][1], ep, ei)
        rows[c] = (p, d)
    sep = np.zeros(nd, dtype=np.int64)
    for c in range(1, cb + 1):
        sep[rows[c][1]] = c
    nint = {}
    for c in range(cb + 1):
        p, d = rows[c]
        a = np.repeat(np.arange(len(p) - 1, dtype=np.int64), np.diff(p))
        if cb > c:
            d = np.ascontiguousarray(d, dtype=np.int64).copy()
            pc = np.ascontiguousarray(p, dtype=np.int64)
            _lib.check(_lib.load().elag_hip_host_sort_rows_by_key(len(pc) - 1, _ptr(pc), _ptr(d), nd, _ptr(sep)))      # reorderRow
            rows[c] = (p, d)
            nint[c] = np.bincount(a, weights=(sep[d] == c), minlength=len(p) - 1).astype(np.int64)
        else:
            nint[c] = np.diff(p)
    return dict(rows=rows, nint=nint, sep=sep, look={c: _Lookup(rows[c][0], rows[c][1], nd) for c in range(cb + 1)})


class _CoarseDofs:
    def __init__(self, lv, topo, j):
        self.cb = lv.cb(j)
        self.cnt = {c: np.zeros(topo["n"][c], dtype=np.int64) for c in range(self.cb + 1)}
        self.n_rt = {c: np.zeros(topo["n"][c], dtype=np.int64) for c in range(self.cb + 1)}
        self.ptr, self.base = {}, {}

    def freeze(self, c):
        self.ptr[c] = ptr_from_counts(self.cnt[c])
        self.base[c] = int(sum(self.cnt[c2].sum() for c2 in range(self.cb, c, -1)))

    def ndof_upto(self, c):
        return int(sum(self.cnt[c2].sum() for c2 in range(self.cb, c - 1, -1)))


def _asm(keep, lv, tb, j, c, topo, look):
    """Assembly list of the codim-c mass of form j over the agglomerated entities of that codimension."""
    ap, ai, _ = topo["A"][c]
    nA = len(ap) - 1
    ent_of_item = np.repeat(np.arange(nA, dtype=np.int64), np.diff(ap))
    ep, ei = tb.edof(j, c)
    n_e = np.diff(ep)[ai]
    q, own = expand_ranges(ep[ai], n_e)
    loc = look.find(ent_of_item[own], ei[q])
    if not np.all(loc >= 0):
        raise ValueError("DofAgglomeration: an entity dof is missing from its agglomerate's dof list (inconsistent level tables)")
    ptr, vals = lv.M[j][c]
    return _assembly(keep, ent_of_item, nA, ptr[ai], n_e, loc, vals)


def _triplets(D: sp.csr_matrix, rows_ptr, rows, look_cols: _Lookup):
    """Local triplets of D restricted to (rows of the entity) x (its columns): threaded host code (``elag_hip_host_local_triplets``)."""
    if D.nnz > np.iinfo(np.int32).max:
        raise ValueError("_triplets: matrix too large for 32-bit CSR indices")
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    rows_ptr, rows = c64(rows_ptr), c64(rows)
    indptr, indices = np.ascontiguousarray(D.indptr, dtype=np.int32), np.ascontiguousarray(D.indices, dtype=np.int32)
    data = np.ascontiguousarray(D.data, dtype=np.float64)
    n = len(rows_ptr) - 1
    d_ptr = np.empty(n + 1, dtype=np.int64)
    args = (n, _ptr(rows_ptr), _ptr(rows), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(look_cols.ptr), _ptr(look_cols.idx), _ptr(d_ptr))
    L = _lib.load()
    _lib.check(L.elag_hip_host_local_triplets(*args, None, None, None))
    nd = int(d_ptr[-1])
    lr, lc, v = np.empty(nd, dtype=np.int64), np.empty(nd, dtype=np.int64), np.empty(nd)
    try:
        _lib.check(L.elag_hip_host_local_triplets(*args, _ptr(lr), _ptr(lc), _ptr(v)))
    except _lib.ElagHipError as e:
        raise ValueError(str(e)) from None
    return d_ptr, lr, lc, v


def _triplets_numpy(D: sp.csr_matrix, rows_ptr, rows, look_cols: _Lookup):
    """Whole-array statement of :func:`_triplets` (the tests compare the two)."""
    ent = np.repeat(np.arange(len(rows_ptr) - 1, dtype=np.int64), np.diff(rows_ptr))
    lr = np.arange(rows.size, dtype=np.int64) - rows_ptr[ent]
    q, own = expand_ranges(D.indptr[rows], np.diff(D.indptr)[rows])
    lc = look_cols.find(ent[own], D.indices[q].astype(np.int64))
    if not np.all(lc >= 0):
        raise ValueError("DistributeAgglomerateMatrix: a derivative couples an agglomerate with a dof outside it")
    e = ent[own]
    return ptr_from_counts(np.bincount(e, minlength=len(rows_ptr) - 1)), lr[own], lc, D.data[q]


# ---- Coarsen ------------------------------------------------------------------------------------------------------------------

def coarsen(lv: SeqLevel, part: np.ndarray, svd_tol: float = 1e-9, smallest_entry: float = EPS, timings: dict | None = None,
            check_topology: bool = False, topo: dict | None = None) -> dict:
    """``DeRhamSequence::Coarsen`` for forms ``lv.jstart .. 3``.  Returns dict(P, hat, coarse, topo, n_null).  ``check_topology``:
    build the agglomerated topology with the checks and repairs of ``CoarsenLocalPartitioning(..., check_topology = true)``.
    ``topo``: an agglomerated topology made elsewhere (``derham_dist``: the global one restricted to a rank's part)."""
    with _lib.host_pages_kept():          # (the table stage's allocator policy, for the duration of this call only)
        return _coarsen_impl(lv, part, svd_tol=svd_tol, smallest_entry=smallest_entry, timings=timings, check_topology=check_topology, topo=topo)


def _coarsen_impl(lv: SeqLevel, part: np.ndarray, svd_tol: float = 1e-9, smallest_entry: float = EPS, timings: dict | None = None,
            check_topology: bool = False, topo: dict | None = None) -> dict:

    L = _lib.init()
    tb = _Tables(lv)
    topo = coarsen_topology(lv, part, check=check_topology) if topo is None else topo
    js = lv.jstart
    agg = {j: agglomerate_dofs(lv, tb, topo, j) for j in range(js, 4)}
    cd = {j: _CoarseDofs(lv, topo, j) for j in range(js, 4)}
    D = {j: lv.D[j].tocsr() for j in range(js, 3)}
    Ppieces = {j: [] for j in range(js, 4)}
    Pcache = {j: [0, None] for j in range(js, 4)}
    keep_all = _Keep()                                   # arrays behind the cached assembly lists / triplets stay alive for the whole call
    asm_cache, trip_cache = {}, {}

    def asm(jj, cc):                                     # the (form, codim) mass of the agglomerated entities: used by up to three stages
        if (jj, cc) not in asm_cache:
            asm_cache[(jj, cc)] = _asm(keep_all, lv, tb, jj, cc, topo, agg[jj]["look"][cc])
        return asm_cache[(jj, cc)]

    def trip(jj, cc):                                    # local triplets of D_jj on the codim-cc agglomerated entities (as D and as D2)
        if (jj, cc) not in trip_cache:
            rp_, rd_ = agg[jj + 1]["rows"][cc]
            trip_cache[(jj, cc)] = _triplets(D[jj], rp_, rd_, agg[jj]["look"][cc])
        return trip_cache[(jj, cc)]
    hat = {j: {} for j in range(js, 4)}
    Mc = {j: {} for j in range(js, 4)}
    Dc_e = {j: ([], [], []) for j in range(js, 3)}
    P, Dc = {}, {}
    n_null = {j: 0 for j in range(js, 4)}
    tms = {} if timings is None else timings
    ms = C.c_double(0.0)

    for codim in range(4):
        j = 3 - codim
        if j < js:
            break
        ctr = codim
        nAt = topo["n"][ctr]
        rp, rd = agg[j]["rows"][ctr]
        nT = lv.T[j].shape[1]
        kmax = 1 + nT
        # ======== coarse traces ========
        if j == 0:                                                   # Compute0formCoarseTraces
            if not np.all(np.diff(rp) == 1):                            # the reference's message (DeRhamSequence.cpp:1553-1556)
                raise ValueError("DeRhamSequence::compute0formCoarseTraces: likely topology error, possibly disconnected edge.")
            ncoarse = np.ones(nAt, dtype=np.int64)
            o_p = np.ones(nAt); o_cm = np.ones(nAt); o_f = np.ones(nAt)
            kmax = 1
        else:
            keep = _Keep()
            ap, ai, asg = topo["A"][ctr]
            # PV vector: base value of the dof times the orientation of its entity in the agglomerate
            ent_of_dof_sgn = np.zeros(lv.ndof(j))
            ip = lv.idof[j][ctr]
            dd, own = expand_ranges(lv.base(j, ctr) + ip[ai], np.diff(ip)[ai])
            ent_of_dof_sgn[dd] = asg[own]
            tin = _TracesIn(nAt, nT, svd_tol, keep.i32(np.diff(rp)), asm(j, ctr),
                            keep.f64(lv.pv[j][rd] * ent_of_dof_sgn[rd]), keep.f64(_pack_cols(lv.T[j], rd, rp)))
            o_nc = np.zeros(nAt, np.int32); o_p = np.zeros(rd.size * kmax); o_cm = np.zeros(nAt * kmax * kmax); o_f = np.zeros(rd.size * kmax)
            _lib.check(L.elag_hip_coarsen_traces(C.byref(tin), C.byref(_TracesOut(_ptr(o_nc), _ptr(o_p), _ptr(o_cm), _ptr(o_f))), C.byref(ms)))
            tms[f"traces{j}_ms"] = ms.value
            ncoarse = o_nc.astype(np.int64)
        cd[j].cnt[ctr][:] = ncoarse
        cd[j].freeze(ctr)
        n_null[j] += int((ncoarse - 1).sum())
        cptr = cd[j].ptr[ctr]
        cidx = cd[j].base[ctr] + np.arange(int(cptr[-1]), dtype=np.int64)
        Ppieces[j].append(("tr", rp, rd, cptr, cidx, o_p, np.full(nAt, kmax)))
        hat[j][ctr] = (cptr, cidx, rp, rd, o_f, np.diff(rp), np.full(nAt, kmax))
        Mc[j][ctr] = (_take_leading(o_cm, np.full(nAt, kmax), ncoarse), ncoarse)

        # ======== extensions ========
        for step, cdom in enumerate(range(ctr - 1, -1, -1)):
            mode = 0 if step == 0 else 1
            null_dofs = 1 if step <= 1 else 0
            nA = topo["n"][cdom]
            ncols_sofar = cd[j].ndof_upto(cdom + 1)
            Pj = _assemble_P(Ppieces[j], lv.ndof(j), ncols_sofar, keep_zeros=True, cache=Pcache[j])
            Pp = P[j + 1]
            cdp = cd[j + 1]
            up, ud = agg[j]["rows"][cdom]; ni = agg[j]["nint"][cdom]; nu = np.diff(up)
            pp, pd = agg[j + 1]["rows"][cdom]; npi = agg[j + 1]["nint"][cdom]; np_ = np.diff(pp)
            # coarse dofs on the boundary (GetDofsOnBdr / hAE_AFCD): sub-entities from the highest codimension down
            tabs = []
            tbc = _CoarseConn(topo)
            for c2 in range(cd[j].cb, cdom, -1):
                sp_, si = tbc.conn(cdom, c2)
                cp2 = cd[j].ptr[c2]
                dcol, own = expand_ranges(cd[j].base[c2] + cp2[si], np.diff(cp2)[si])
                cnt = np.bincount(np.repeat(np.arange(nA, dtype=np.int64), np.diff(sp_))[own], minlength=nA)
                tabs.append((ptr_from_counts(cnt), dcol))
            cbp, cbd = _concat_rows(tabs, nA)
            nxt = np.diff(cbp)
            look_cbd = _Lookup(cbp, cbd, ncols_sofar)
            # boundary rows of every agglomerate
            bq, ownb = expand_ranges(up[:-1] + ni, nu - ni)
            b_ptr = ptr_from_counts(nu - ni)
            R_b = _gather_dense(Pj, b_ptr, ud[bq], look_cbd, nxt)
            keep = _Keep()
            d_ptr, d_row, d_col, d_val = trip(j, cdom)
            pint_ptr = cdp.ptr[cdom]; pint_base = cdp.base[cdom]
            iq, owni = expand_ranges(pp[:-1], npi)                     # interior p dofs
            ip_ptr = ptr_from_counts(npi)
            if mode == 0:
                if not np.array_equal(npi, np_):
                    raise ValueError("hFacetExtension: form j+1 has dofs on entities of higher codimension than its trace entities")
                nrt = cdp.cnt[cdom] - 1
                pv_cols = (ptr_from_counts(np.ones(nA, dtype=np.int64)), pint_base + pint_ptr[:-1])
                pv_p = _gather_dense(Pp, pp, pd, _Lookup(pv_cols[0], pv_cols[1], Pp.shape[1]), np.ones(nA, dtype=np.int64))
                null_start = pint_ptr[:-1] + 1
            else:
                nrt = cdp.cnt[cdom] - cdp.n_rt[cdom]
                pv_p = np.zeros(int(np_.sum()))
                null_start = pint_ptr[:-1] + cdp.n_rt[cdom]
            ncol_, own_n = expand_ranges(pint_base + null_start, nrt)
            nl_ptr = ptr_from_counts(nrt)
            cP = _gather_dense(Pp, ip_ptr, pd[iq], _Lookup(nl_ptr, ncol_, Pp.shape[1]), nrt)
            args = dict(mass_q=_Assembly(), d2=(None, None, None, None), PDc=None, npi=None, nq=None)
            if mode == 1:
                qp, qd = agg[j + 2]["rows"][cdom]
                d2_ptr, d2_row, d2_col, d2_val = trip(j + 1, cdom)
                r_, c_, v_ = Dc_e[j]
                Dc_now = sp.csr_matrix((np.concatenate(v_) if v_ else np.zeros(0), (np.concatenate(r_) if r_ else np.zeros(0, np.int64),
                                                                                     np.concatenate(c_) if c_ else np.zeros(0, np.int64))),
                                       shape=(Pp.shape[1], ncols_sofar))
                PDc = _spgemm_device(Pp, Dc_now)                                    # :2696-2697
                args = dict(mass_q=asm(j + 2, cdom),
                            d2=(keep.i64(d2_ptr), keep.i32(d2_row), keep.i32(d2_col), keep.f64(d2_val)),
                            PDc=keep.f64(_gather_dense(PDc, pp, pd, look_cbd, nxt)), npi=keep.i32(npi), nq=keep.i32(np.diff(qp)))
            ein = _ExtIn(nA, nT, mode, null_dofs, svd_tol, smallest_entry, keep.i32(nu), keep.i32(ni), keep.i32(np_), args["npi"], args["nq"],
                         keep.i32(nxt), keep.i32(nrt),
                         asm(j, cdom), asm(j + 1, cdom),
                         args["mass_q"], keep.i64(d_ptr), keep.i32(d_row), keep.i32(d_col), keep.f64(d_val), *args["d2"],
                         keep.f64(pv_p), keep.f64(R_b), keep.f64(cP), keep.f64(_pack_cols(lv.T[j], ud, up)), args["PDc"])
            kf = nrt + nT
            ncm = nxt + nrt + nT
            o_nn = np.zeros(nA, np.int32)
            o_ext = np.zeros(int((ni * nxt).sum())); o_dc = np.zeros(int(nxt.sum())); o_rt = np.zeros(int((ni * nrt).sum()))
            o_nul = np.zeros(int((ni * nT).sum())); o_fun = np.zeros(int((kf * ni).sum())); o_cme = np.zeros(int((ncm * ncm).sum()))
            eout = _ExtOut(_ptr(o_nn), _ptr(o_ext), _ptr(o_dc), _ptr(o_rt), _ptr(o_nul), _ptr(o_fun), _ptr(o_cme))
            _lib.check(L.elag_hip_coarsen_extension(C.byref(ein), C.byref(eout), C.byref(ms)))
            tms[f"extension{j}_{cdom}_ms"] = ms.value
            nn = o_nn.astype(np.int64)
            nint_c = nrt + nn
            cd[j].cnt[cdom][:] = nint_c
            cd[j].n_rt[cdom][:] = nrt
            cd[j].freeze(cdom)
            n_null[j] += int(nn.sum())
            ci_ptr = cd[j].ptr[cdom]
            ci_idx = cd[j].base[cdom] + np.arange(int(ci_ptr[-1]), dtype=np.int64)
            i_ptr = ptr_from_counts(ni)
            iu, _o = expand_ranges(up[:-1], ni)
            int_dofs = ud[iu]
            Ppieces[j].append(("ext", i_ptr, int_dofs, cbp, cbd, o_ext, nxt))
            Ppieces[j].append(("bub", i_ptr, int_dofs, ci_ptr, ci_idx, _join_cols(o_rt, ni, nrt, o_nul, nT, nn), nint_c))
            hat[j][cdom] = (ci_ptr, ci_idx, i_ptr, int_dofs, _take_leading_rows(o_fun, kf, ni, nint_c), ni, nint_c)
            Mc[j][cdom] = (_take_leading(o_cme, ncm, nxt + nint_c), nxt + nint_c)
            r_, c_, v_ = Dc_e[j]
            if mode == 0:
                sel = o_dc != 0.0
                pv_row = pint_base + pint_ptr[:-1]
                r_.append(np.repeat(pv_row, nxt)[sel]); c_.append(cbd[sel]); v_.append(o_dc[sel])
            rt_rows, _o = expand_ranges(pint_base + null_start, nrt)
            rt_cols, _o = expand_ranges(cd[j].base[cdom] + ci_ptr[:-1], nrt)
            r_.append(rt_rows); c_.append(rt_cols); v_.append(np.ones(rt_rows.size))
        # ======== finalise form j ========
        ndc = cd[j].ndof_upto(0)
        P[j] = _assemble_P(Ppieces[j], lv.ndof(j), ndc, keep_zeros=False, cache=Pcache[j])
        if j < 3:
            r_, c_, v_ = Dc_e[j]
            Dc[j] = sp.csr_matrix((np.concatenate(v_), (np.concatenate(r_), np.concatenate(c_))), shape=(P[j + 1].shape[1], ndc))
            Dc[j].sort_indices()

    # ---- projector pieces and the projected targets (CochainProjector::Project, CochainProjector.cpp:158-216) ----
    hats, Tc = {}, [None] * 4
    for j in range(js, 4):
        ndc = P[j].shape[1]
        hats[j] = {}
        for c, h in hat[j].items():
            cptr, cidx, rp, rd, vals, ld, slot_rows = h
            hats[j][c] = _blocks_to_csr(cptr, cidx, rp, rd, vals, ld, (ndc, lv.ndof(j)), col_major=False, slot_rows=slot_rows)
        v = np.zeros((ndc, lv.T[j].shape[1]))
        res = lv.T[j]
        for c in range(lv.cb(j), -1, -1):
            v = v + _spmm_device(hats[j][c], res)
            if c > 0:
                res = lv.T[j] - _spmm_device(P[j], v)
        Tc[j] = v
    const3 = _spmm_device(hats[3][0], lv.const3[:, None])[:, 0]

    idof_c, M_c = [None] * 4, [None] * 4
    for j in range(js, 4):
        idof_c[j] = [cd[j].ptr[c] for c in range(lv.cb(j) + 1)]
        M_c[j] = [(ptr_from_counts(Mc[j][c][1] * Mc[j][c][1]), Mc[j][c][0]) for c in range(lv.cb(j) + 1)]
    coarse = SeqLevel(n=topo["n"], B=topo["B"], facet_attr=topo["facet_attr"], jstart=js, idof=idof_c, M=M_c,
                      D=[Dc.get(j) for j in range(3)], T=Tc, pv=[None] * 4, const3=const3, meta=dict(kind="alg"))
    for j in range(js, 4):
        vv = np.zeros(coarse.ndof(j)); ctr = coarse.cb(j)
        p = coarse.idof[j][ctr]
        vv[coarse.base(j, ctr) + p[:-1][np.diff(p) > 0]] = 1.0
        coarse.pv[j] = vv
    return dict(P=P, hat=hats, coarse=coarse, topo=topo, n_null=n_null, dofs=cd)


def projector(res: dict, j: int) -> sp.csr_matrix:
    """``CochainProjector::ComputeProjector``: ``Pi_{c-1} = Pi_c + hat_{c-1} (I - P Pi_c)`` (test helper: host sparse algebra)."""
    cur = None
    for c in sorted(res["hat"][j].keys(), reverse=True):
        h = res["hat"][j][c]
        cur = h if cur is None else (cur + h - h @ (res["P"][j] @ cur))
    cur = cur.tocsr(); cur.sort_indices()
    return cur


class _CoarseConn:
    def __init__(self, topo):
        self.B = topo["B"]
        self._c = {}

    def conn(self, c, c2):
        if (c, c2) not in self._c:
            B = self.B
            if c2 == c + 1:
                self._c[(c, c2)] = (B[c][0], B[c][1])
            elif (c, c2) == (0, 3):
                p, i = self.conn(0, 2)
                self._c[(c, c2)] = _product(p, i, B[2][0], B[2][1])[:2]
            else:
                self._c[(c, c2)] = _product(B[c][0], B[c][1], B[c + 1][0], B[c + 1][1])[:2]
        return self._c[(c, c2)]


def _assemble_P(pieces, nrows, ncols, keep_zeros, cache=None):
    """Sum of the dense-block pieces of an interpolator as one CSR with ``ncols`` columns.  ``cache`` = [n pieces already
    summed, their sum]: only the new pieces are converted and added."""
    done, out = (cache[0], cache[1]) if cache else (0, None)
    fresh = [piece[1:] for piece in pieces[done:]]
    if fresh:                                               # the new pieces in one pass (their (row, column) sets are disjoint)
        M = _blocks_to_csr_sets(fresh, (nrows, ncols))
        if out is None:
            out = M
        else:
            if out.shape[1] != ncols:                       # earlier pieces only touch the first columns: widen without copying data
                out = sp.csr_matrix((out.data, out.indices, out.indptr), shape=(nrows, ncols))
            out = out + M
    if out is not None and out.shape[1] != ncols:
        out = sp.csr_matrix((out.data, out.indices, out.indptr), shape=(nrows, ncols))
    out = out.tocsr()
    out.sort_indices()
    if cache is not None:
        cache[0], cache[1] = len(pieces), out
    return out


def _spgemm_device(A: sp.csr_matrix, B: sp.csr_matrix) -> sp.csr_matrix:
    if A.nnz == 0 or B.nnz == 0:
        return sp.csr_matrix((A.shape[0], B.shape[1]))
    from . import linalg as la
    out = la.ParallelCSRMatrix.from_scipy(A.tocsr()).matmul(la.ParallelCSRMatrix.from_scipy(B.tocsr())).to_scipy()
    return out.tocsr()


# ---- the examples' problems on AMGe spaces --------------------------------------------------------------------------------

def boundary_dofs(lv: SeqLevel, j: int) -> np.ndarray:
    """Marker of the dofs of form j lying on boundary facets with an attribute (``DofHandler::MarkDofsOnSelectedBndr`` with
    every attribute selected, ``src/amge/DofHandler.cpp:105-137,812-852``)."""
    mk = np.zeros(lv.ndof(j), dtype=bool)
    if lv.cb(j) < 1:
        return mk
    ep, ei = _Tables(lv).edof(j, 1)
    bf = np.nonzero(lv.facet_attr > 0)[0]
    q, _ = expand_ranges(ep[bf], np.diff(ep)[bf])
    mk[ei[q]] = True
    return mk


def amge_problem(form: int, seq=None, level: SeqLevel | None = None, parts: list | None = None, what: str | None = None, coarsen_fn=None):
    """The operators of MultigridTest0Form / 1Form / 2Form on AMGe spaces built by the device coarsening: fine operator
    (``D0^T M1 D0`` resp. ``M_j + D_j^T M_{j+1} D_j``) with all boundary attributes essential, the interpolators ``P_form`` of
    every level with essential columns eliminated (``ComputeTrueP(jform, ess)``, ``src/amge/DeRhamSequence.cpp:1118-1153,1230-1253``)
    and, for the Hiptmair smoother, the derivative ``D_{form-1}`` of every level with essential columns eliminated
    (``HiptmairSmootherFactory.cpp:84-103``).  Either ``seq`` (the generator's refined cube456: derefinement agglomerates)
    or a finest ``level`` with its list of element partitionings.  ``coarsen_fn(level, part, l)`` replaces the single-rank
    ``coarsen`` (``derham_dist.coarsen_allgather`` on several ranks).  Returns (A, Ps, Ds or None, levels, record)."""
    import time
    from . import meshgen
    from .derham_level import seq_level_from_tets
    t0 = time.time()
    if level is None:
        lv = seq_level_from_tets(seq.meshes[0], jstart=0 if form <= 1 else 1)
        parts = [seq.partition(l) for l in range(len(seq.meshes) - 1)]
        what = "cube456 x%d tets, derefinement agglomerates" % (len(seq.meshes) - 1)
    else:
        lv = level
    gen_s = time.time() - t0
    levels, Ps, recs = [lv], [], []
    t0 = time.time()
    for part in parts:
        tm = {}
        t1 = time.time()
        res = coarsen(levels[-1], part, timings=tm) if coarsen_fn is None else coarsen_fn(levels[-1], part, len(Ps))
        c = res["coarse"]
        recs.append({"agglomerates": c.n[0], "entities": c.n, "dofs": [c.ndof(j) for j in range(c.jstart, 4)],
                     "null_dofs": {str(j): v for j, v in res["n_null"].items()},
                     "wall_s": round(time.time() - t1, 4), "kernel_ms": round(sum(tm.values()), 4)})
        Ps.append(res["P"][form])
        levels.append(c)
    total_s = time.time() - t0
    marks = [boundary_dofs(L_, form) for L_ in levels]
    if form == 0:
        A = (lv.D[0].T @ lv.mass(1) @ lv.D[0]).tocsr()
        Ds = None
    else:                                               # MultigridTest1Form / 2Form: M_j + D_j^T M_{j+1} D_j, Hiptmair through D_{j-1}
        A = (lv.mass(form) + lv.D[form].T @ lv.mass(form + 1) @ lv.D[form]).tocsr()
        Ds = [meshgen.eliminate_cols(L_.D[form - 1].tocsr(), boundary_dofs(L_, form - 1)) for L_ in levels]
    A.sort_indices()
    A = meshgen.eliminate_rowcol(A, marks[0])
    Ps = [meshgen.eliminate_cols(P.tocsr(), marks[l + 1]) for l, P in enumerate(Ps)]
    rec = {"what": "DeRhamSequence::Coarsen (jformStart=%d) on the device: %s" % (lv.jstart, what),
           "levels": recs, "coarsen_total_s": round(total_s, 3), "generate_s": round(gen_s, 2)}
    return A, Ps, Ds, levels, rec


# ---- the reference's class interface ------------------------------------------------------------------------------------------

class DeRhamSequence:
    """Mirror of ``parelag::DeRhamSequence`` (``src/amge/DeRhamSequence.hpp:64-330``) over the device coarsening: the same
    member names and argument meaning, so that a driver written against the reference
    (``examples/MultigridTest0Form.cpp:214-300``: ``SetjformStart`` / ``SetTargets`` / ``SetSVDTol`` / ``Coarsen`` /
    ``ComputeTrueP`` / ``ComputeMassOperator`` / ``GetDerivativeOperator``) reads the same here.  Matrices are scipy CSR on the
    host (the reference returns ``mfem::SparseMatrix``); one rank, so "true" dofs are the dofs."""

    def __init__(self, level: SeqLevel, finer: "DeRhamSequence | None" = None):
        self.level = level
        self.FinerSequence = finer
        self.CoarserSequence = None
        self.SVD_Tolerance = 1e-9                       # DeRhamSequence.cpp:56
        self.SmallestEntry = EPS                        # :57
        self._res = None

    # -- getters ------------------------------------------------------------------------------------------------------------
    def GetNumberOfForms(self):
        return 4

    def GetNumberOfDofs(self, jform):
        return self.level.ndof(jform)

    GetNumberOfTrueDofs = GetNumberOfDofs

    def GetD(self, jform):
        return self.level.D[jform]

    GetDerivativeOperator = GetD

    def GetP(self, jform):
        self._need_coarse("GetP")
        return self._res["P"][jform]

    def GetPi(self, jform):
        """The cochain projector as a matrix (``CochainProjector::GetProjectorMatrix``)."""
        self._need_coarse("GetPi")
        return projector(self._res, jform)

    def GetM(self, etype, jform):
        """Entity mass matrices of codimension ``etype`` as one "DG-like" block-diagonal matrix (``:207-213``)."""
        ptr, vals = self.level.M[jform][etype]
        sizes = np.sqrt(np.diff(ptr)).round().astype(np.int64)
        return sp.block_diag([vals[ptr[e]:ptr[e + 1]].reshape(sizes[e], sizes[e]) for e in range(sizes.size)], format="csr")

    def ComputeMassOperator(self, jform, elemMatrixScaling=None):
        """``ComputeMassOperator(jform[, elemMatrixScaling])`` (``DeRhamSequence.hpp:407-425``): the element blocks, each scaled
        by its entry of ``elemMatrixScaling`` when given, assembled over the dofs."""
        return self.level.mass(jform, scaling=elemMatrixScaling)

    def ComputeTrueMassOperator(self, jform, elemMatrixScaling=None):
        return self.level.mass(jform, scaling=elemMatrixScaling)

    ComputeTrueM = ComputeTrueMassOperator

    def GetNumTrueDofs(self, jform):
        return self.GetNumberOfDofs(jform)                       # one rank: every dof is a true dof

    GetNumberOfTrueDofs = GetNumTrueDofs

    def GetDerivativeOperator(self, jform):
        return self.GetD(jform)

    def ComputeTruePi(self, jform):
        return self.GetPi(jform)

    def GetL2ConstRepresentation(self):
        """Coefficients of the constant function in the last form (``L2_const_rep_``, ``:452-453``)."""
        return self.level.const3

    def ComputeTrueDerivativeOperator(self, jform):
        return self.level.D[jform]

    def ComputeTrueD(self, jform, ess_label=None):
        """``ComputeTrueD(jform, ess_label)`` (``:1230-1253``): columns of the essential dofs eliminated."""
        from . import meshgen
        D = self.level.D[jform].tocsr()
        return D if ess_label is None else meshgen.eliminate_cols(D, self._marker(jform, ess_label))

    def ComputeTrueP(self, jform, ess_label=None):
        """``ComputeTrueP(jform, ess_label)`` (``:1118-1153,1230-1253``): essential coarse columns eliminated."""
        from . import meshgen
        self._need_coarse("ComputeTrueP")
        P = self._res["P"][jform].tocsr()
        return P if ess_label is None else meshgen.eliminate_cols(P, self.CoarserSequence._marker(jform, ess_label))

    def _marker(self, jform, ess_label):
        """Dofs on boundary facets whose attribute is selected (``ess_label[attr-1] != 0``; ``MarkDofsOnSelectedBndr``)."""
        lab = np.asarray(ess_label)
        mk = np.zeros(self.level.ndof(jform), dtype=bool)
        if self.level.cb(jform) < 1:
            return mk
        ep, ei = _Tables(self.level).edof(jform, 1)
        attr = self.level.facet_attr
        sel = np.nonzero((attr > 0) & (lab[np.maximum(attr, 1) - 1] != 0))[0]
        q, _ = expand_ranges(ep[sel], np.diff(ep)[sel])
        mk[ei[q]] = True
        return mk

    # -- setters ------------------------------------------------------------------------------------------------------------
    def SetjformStart(self, start):
        if not 0 <= start <= 3:
            raise ValueError("SetjformStart: start must be one of the forms 0 .. 3")
        for j in range(start, 4):
            if self.level.M[j] is None:
                raise ValueError(f"SetjformStart({start}): this sequence was built without form {j}")
        self.level.jstart = int(start)

    def SetTargets(self, targets):
        """``SetTargets`` (``:63-96``): one (ndof x nTargets) array per form (deep copy); forms below ``jformStart`` may be None."""
        for j in range(self.level.jstart, 4):
            T = np.array(targets[j], dtype=np.float64, copy=True)
            if T.ndim != 2 or T.shape[0] != self.level.ndof(j):
                raise ValueError(f"SetTargets: targets of form {j} must have {self.level.ndof(j)} rows")
            self.level.T[j] = T

    def SetSVDTol(self, tol):
        self.SVD_Tolerance = float(tol)

    # -- math ---------------------------------------------------------------------------------------------------------------
    def Coarsen(self, partitioning, timings: dict | None = None, check_topology: bool = False) -> "DeRhamSequence":
        """``Coarsen()`` (``:572-692``).  The element partitioning is an argument here because this mirror has no separate
        ``AgglomeratedTopology`` object: the reference reads it from ``Topo_->CoarserTopology()``, which
        ``CoarsenLocalPartitioning(partitioning, check_topology, ...)`` built from the same array; the diagnostics of a checked
        coarsening are kept in ``self.TopologyMessages``."""
        self._res = coarsen(self.level, partitioning, svd_tol=self.SVD_Tolerance, smallest_entry=self.SmallestEntry, timings=timings,
                            check_topology=check_topology)
        self.TopologyMessages = self._res["topo"].get("messages", [])
        self.CoarserSequence = DeRhamSequence(self._res["coarse"], finer=self)
        self.CoarserSequence.SVD_Tolerance = self.SVD_Tolerance
        return self.CoarserSequence

    def CheckInvariants(self, tol=1e-9, loose_tol=1e-6):
        """``CheckInvariants`` (``:694-970``): CheckD (``D_{j+1} D_j = 0``), CheckDP (``D_f P_j = P_{j+1} D_c``),
        CheckCoarseMassMatrix (``M_c = P^T M P``), CheckPi (``Pi P = I``).  Raises like the reference's
        ``PARELAG_TEST_FOR_EXCEPTION``; returns the measured defects."""
        self._need_coarse("CheckInvariants")
        lv, c, P = self.level, self._res["coarse"], self._res["P"]
        out = {}
        for j in range(lv.jstart, 4):
            M, Mc = lv.mass(j), c.mass(j)
            out[f"CheckCoarseMassMatrix[{j}]"] = abs(P[j].T @ M @ P[j] - Mc).max() / max(abs(Mc).max(), 1e-300)
            out[f"CheckPi[{j}]"] = abs(projector(self._res, j) @ P[j] - sp.identity(P[j].shape[1])).max()
            if j < 3:
                out[f"CheckDP[{j}]"] = abs(lv.D[j] @ P[j] - P[j + 1] @ c.D[j]).max() / max(abs(lv.D[j]).max(), 1e-300)
            if j < 2:
                out[f"CheckD[{j}]"] = abs(lv.D[j + 1] @ lv.D[j]).max()
        bad = {k: v for k, v in out.items() if v > (loose_tol if k.startswith("CheckCoarseMass") else tol)}
        if bad:
            raise RuntimeError(f"DeRhamSequence::CheckInvariants(): {bad}")
        return out

    def ComputeSpaceInterpolationError(self, jform, fineVector, out: list | None = None):
        """``ComputeSpaceInterpolationError`` (``:972-1080``): for this sequence (when it has a finer one) and every coarser one,
        project the finest-level vectors down with the cochain projectors, interpolate back, and measure per vector
        ``|| v - P Pi v ||_M / || v ||_M`` and the graph-norm version with ``D``.  The reference prints one row per level
        (number of coarse dofs, then the ratios); the rows are returned here as (ndofs, l2_ratios, graph_ratios or None)."""
        out = [] if out is None else out
        V0 = np.asarray(fineVector, dtype=np.float64)
        if self.FinerSequence is not None:
            chain = [self]
            while chain[-1].FinerSequence is not None:
                chain.append(chain[-1].FinerSequence)
            finest = chain[-1]
            if V0.ndim != 2 or V0.shape[0] != finest.level.ndof(jform):
                raise ValueError("DeRhamSequence::ComputeSpaceInterpolationError(): fineVector size is wrong.")
            v = V0
            for seq in reversed(chain[1:]):                       # finest ... the sequence just above this one
                v = seq.GetPi(jform) @ v
            for seq in chain[1:]:
                v = seq.GetP(jform) @ v
            diff = V0 - v
            M = finest.level.mass(jform)
            l2d = np.einsum("ij,ij->j", diff, M @ diff)
            l2v = np.einsum("ij,ij->j", V0, M @ V0)
            graph = None
            if jform < 3:
                W, D = finest.level.mass(jform + 1), finest.level.D[jform]
                dd, dv = D @ diff, D @ V0
                l2dd = np.einsum("ij,ij->j", dd, W @ dd)
                l2dv = np.einsum("ij,ij->j", dv, W @ dv)
                l2dv = np.where(np.abs(l2v + l2dv) < 1e-9, 1.0, l2dv)          # DEFAULT_EQUALITY_TOL
                graph = np.sqrt((l2d + l2dd) / (l2v + l2dv))
            out.append((self.level.ndof(jform), np.sqrt(l2d / l2v), graph))
        if self.CoarserSequence is not None:
            self.CoarserSequence.ComputeSpaceInterpolationError(jform, V0, out)
        return out

    def _need_coarse(self, what):
        if self._res is None:
            raise RuntimeError(f"DeRhamSequence::{what}(): call Coarsen() first")


# ---- topology checks ------------------------------------------------------------------------------------------------------------

def _B_csr(lv: SeqLevel, c: int) -> sp.csr_matrix:
    return sp.csr_matrix((lv.B[c][2], lv.B[c][1], lv.B[c][0]), shape=(lv.n[c], lv.n[c + 1]))


def betti_numbers(lv: SeqLevel, topo, codim: int) -> np.ndarray:
    """``AgglomeratedTopologyCheck::computeBettiNumbersAgglomeratedEntities`` (``src/topology/AgglomeratedTopologyCheck.cpp:241-318``):
    for every agglomerated entity of codimension ``codim`` the Betti numbers of the sub-complex of the fine entities it is
    made of -- column 0 connected components, then (elements) tunnels and holes, (facets) holes.  The ranks of the local
    boundary operators come from the device (``elag_hip_batched_rank``)."""
    Ac = topo["A"][codim] if isinstance(topo, dict) else topo
    nlow = 3 - codim
    nA = len(Ac[0]) - 1
    if nlow == 0:
        return np.zeros((nA, 0), dtype=np.int64)
    rows = [(Ac[0], Ac[1])]
    for i in range(nlow):                                   # MultBoolean chain: all sub-entities, interior ones included
        rows.append(_product(rows[i][0], rows[i][1], lv.B[codim + i][0], lv.B[codim + i][1])[:2])
    dims = [np.diff(r[0]) for r in rows]
    ranks = []
    for i in range(nlow):
        look = _Lookup(rows[i + 1][0], rows[i + 1][1], lv.n[codim + i + 1])
        t_ptr, t_row, t_col, t_val = _triplets(_B_csr(lv, codim + i), rows[i][0], rows[i][1], look)
        L = _lib.init()
        keep = _Keep()
        out = np.zeros(nA, dtype=np.int32)
        _lib.check(L.elag_hip_batched_rank(nA, keep.i32(dims[i]), keep.i32(dims[i + 1]), keep.i64(t_ptr), keep.i32(t_row), keep.i32(t_col),
                                           keep.f64(t_val), 1e-9, _ptr(out)))
        ranks.append(out.astype(np.int64))
    ranks.append(np.zeros(nA, dtype=np.int64))
    betti = np.zeros((nA, nlow), dtype=np.int64)
    for i in range(nlow):
        betti[:, nlow - i - 1] = dims[i + 1] - ranks[i] - ranks[i + 1]
    return betti


def _bad_connectivity(lv: SeqLevel, topo: dict, codim: int) -> np.ndarray:
    """``additionalTopologyCheck`` (``:25-83``): on the boundary of an agglomerated element (facet) every boundary ridge (peak)
    must lie on exactly two boundary facets (ridges) -- the reference compares the sums only."""
    ap, ai, asg = topo["A"][codim] if isinstance(topo, dict) else topo
    fp, fi, _fs = _product(ap, ai, lv.B[codim][0], lv.B[codim][1], asg, lv.B[codim][2])      # boundary facets (interior ones cancel)
    ep, _ei, _ = _product(fp, fi, lv.B[codim + 1][0], lv.B[codim + 1][1])                      # boundary ridges
    per_facet = np.diff(lv.B[codim + 1][0])[fi]
    nnz = np.bincount(np.repeat(np.arange(len(fp) - 1), np.diff(fp)), weights=per_facet, minlength=len(fp) - 1)
    return np.abs(nnz - 2 * np.diff(ep)) > 1e-10


_ENTITY_NAMES = {0: "Element", 1: "Facet", 2: "Ridge"}


def _check_codim(lv: SeqLevel, topo, codim: int):
    """Betti numbers, bad flags and the reference's diagnostic lines for the agglomerated entities of one codimension
    (``topo``: the coarsened topology, or the (ptr, idx, sgn) table of that codimension while it is being built)."""
    b = betti_numbers(lv, topo, codim)
    bad = b[:, 0] != 1
    msg = [f"    {_ENTITY_NAMES[codim]} {a} is disconnected. The number of connected components is {b[a, 0]}" for a in np.nonzero(bad)[0]]
    for i in range(1, b.shape[1]):
        bad |= b[:, i] != 0
        what = "holes" if (codim == 1 or i == 2) else "tunnels"
        msg += [f"    {_ENTITY_NAMES[codim]} {a} has {b[a, i]} {what}." for a in np.nonzero(b[:, i] != 0)[0]]
    if codim <= 1:
        bc = _bad_connectivity(lv, topo, codim)
        bad |= bc
        msg += [f"    codim {codim} iAE {a} has bad connectivity (eg boundary edge adjacent to >2 boundary faces)." for a in np.nonzero(bc)[0]]
    return b, bad, msg


def show_me(topo: dict, weights: list | None = None) -> str:
    """``AgglomeratedTopology::ShowMe`` (``src/topology/Topology.cpp:310-391``) for a coarsened topology on one rank: entity counts
    (local and global columns coincide), the Euler characteristic, and -- when the fine entities' weights are given, one array
    per codimension 0 .. 2 -- the MIN / MAX / AVERAGE weight of the agglomerated entities (``WedgeMult``: sums over members)."""
    n = topo["n"]
    lines = []
    for name, c in (("N_elements", 0), ("N_facets  ", 1), ("N_ridges  ", 2), ("N_peaks   ", 3)):
        lines.append("  %s = %10d%10d" % (name, n[c], n[c]))
    euler = n[3] - n[2] + n[1] - n[0]
    lines.append("Euler Characteristic = %10d%10d" % (euler, euler))
    if weights is not None:
        lines.append("Entity   MIN  MAX AVERAGE (weight)")
        for name, c in (("Element", 0), ("Facet", 1), ("Ridge", 2)):
            ptr, idx, _ = topo["A"][c]
            w = np.add.reduceat(np.asarray(weights[c])[idx], ptr[:-1]) if idx.size else np.zeros(0)
            lines.append("%s %g %g %g" % (name, w.min(), w.max(), w.mean()))
    return "\n".join(lines) + "\n"


def check_topology(lv: SeqLevel, topo: dict) -> dict:
    """``MarkBadAgglomeratedEntities`` / ``ShowBadAgglomeratedEntities`` (``:85-240``) for agglomerated elements, facets and
    ridges: Betti numbers, the extra boundary-connectivity test for codimensions 0 and 1, the reference's diagnostic lines.
    Returns dict(betti={codim: array}, isbad={codim: bool array}, messages=[...])."""
    out = dict(betti={}, isbad={}, messages=[])
    for codim in range(3):
        out["betti"][codim], out["isbad"][codim], msg = _check_codim(lv, topo, codim)
        out["messages"] += msg
    return out
