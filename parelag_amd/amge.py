"""Host side of the AMGe coarse-space construction for the (H(div), L2) pair.

Mirrors ``DeRhamSequence::Coarsen`` (``src/amge/DeRhamSequence.cpp:572-692``) as it runs with ``jformStart_ = nDim-1``
(``examples/MultigridTestSPE10.cpp:441``).  The split follows the reference's own: *integer tables* -- the agglomerated
topology (``AgglomeratedTopology::CoarsenLocalPartitioning``, ``src/topology/Topology.cpp:686-828``) and the dof
agglomeration (``DofAgglomeration``, ``src/amge/DOFAgglomeration.cpp:33-314``) -- are built here with vectorised numpy;
*every floating-point operation* of the per-agglomerate loops (``ComputeCoarseTracesWithTargets`` ``:1818-1926``,
``hFacetExtension`` ``:2364-2556``) runs on the device through ``elag_hip_coarsen_traces`` /
``elag_hip_coarsen_extension`` (``csrc/amge.hip``); the projection of the targets (``CochainProjector::Project``,
``src/amge/CochainProjector.cpp:158-216``) uses the device SpMV.  There is no CPU fallback: without the HIP library the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import scipy.sparse as sp

from . import _lib
from .amge_level import AmgeLevel, ptr_from_counts

EPS = float(np.finfo(np.float64).eps)


def expand_ranges(starts, counts):
    """Concatenation of ``arange(starts[i], starts[i] + counts[i])``; also returns the owner index of every entry."""
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    tot = int(counts.sum())
    idx, owner = np.empty(tot, dtype=np.int64), np.empty(tot, dtype=np.int64)
    _lib.check(_lib.load().elag_hip_host_expand_ranges(counts.size, _ptr(starts), _ptr(counts), tot, _ptr(idx), _ptr(owner)))
    return idx, owner


# ---- integer tables --------------------------------------------------------------------------------------------------------

def coarsen_topology(lv: AmgeLevel, part: np.ndarray) -> dict:
    """Agglomerated elements / facets and the coarse element-facet table (``Topology.cpp:686-828``, codimensions 0 and 1,
    ``coarsefaces_algo = 0``).  The minimal intersection sets (``src/structures/minimalIntersectionSet.cpp:43-132``) are
    found by key instead of through the matrix ``Z``: two facets on agglomerate boundaries fall into the same set iff
    they separate the same pair of agglomerates, or lie on the domain boundary of the same agglomerate with the same
    boundary attribute and the same orientation (the attribute term of ``Z`` cannot cancel a sign), or -- without an
    attribute table -- on the domain boundary of the same agglomerate.  Sets are numbered by their lowest facet, whose
    orientation they take."""
    part = np.asarray(part, dtype=np.int64)
    nAE = int(part.max()) + 1
    order = np.argsort(part, kind="stable")
    ae_ptr = ptr_from_counts(np.bincount(part, minlength=nAE))
    from .derham_level import _product                                 # native first-touch products (csrc/host_tables.hip)
    fc_ptr, f, s = _product(ae_ptr, order, lv.ef_ptr, lv.ef_idx, np.ones(order.size), lv.ef_sgn)       # AE_fc, first-touch order
    # minimal intersection sets from the facets' adjacency signatures (native: csrc/host_tables.hip)
    fcp, fc, sc = np.ascontiguousarray(fc_ptr, dtype=np.int64), np.ascontiguousarray(f, dtype=np.int64), np.ascontiguousarray(s, dtype=np.float64)
    attr32 = np.ascontiguousarray(lv.facet_attr, dtype=np.int32)
    mis_of, ori, nset = np.empty(lv.nF, dtype=np.int64), np.empty(lv.nF), np.zeros(1, dtype=np.int64)
    _lib.check(_lib.load().elag_hip_host_minimal_intersection_sets(nAE, _ptr(fcp), _ptr(fc), _ptr(sc), lv.nF, _ptr(attr32), _ptr(mis_of), _ptr(ori),
                                                                   _ptr(nset)))
    nAF = int(nset[0])
    fid = np.nonzero(mis_of >= 0)[0]
    o2 = np.argsort(mis_of[fid], kind="stable")
    af_ptr = ptr_from_counts(np.bincount(mis_of[fid], minlength=nAF))
    af_idx = fid[o2]
    # coarse element-facet table: MultOrientation(AE_fc, fc_AF)
    member = mis_of >= 0
    B_ptr, bc, bs = _product(fc_ptr, f, ptr_from_counts(member.astype(np.int64)), mis_of[member], s, ori[member])
    return dict(nAE=nAE, nAF=nAF, ae_ptr=ae_ptr, ae_idx=order.astype(np.int64),
                af_ptr=af_ptr, af_idx=af_idx, af_sgn=ori[af_idx],
                B_ptr=B_ptr, B_idx=bc.astype(np.int64), B_sgn=bs,
                facet_attr=lv.facet_attr[af_idx[af_ptr[:-1]]].astype(np.int32))


class _Assembly(C.Structure):
    _fields_ = [("asm_ptr", C.c_void_p), ("blk_off", C.c_void_p), ("blk_n", C.c_void_p), ("loc_off", C.c_void_p),
                ("loc_idx", C.c_void_p), ("blocks", C.c_void_p), ("n_items", C.c_int64), ("n_loc", C.c_int64),
                ("n_block_values", C.c_int64)]


class _TracesIn(C.Structure):
    _fields_ = [("n_ent", C.c_int), ("nT", C.c_int), ("svd_tol", C.c_double), ("ndof", C.c_void_p), ("mass", _Assembly),
                ("pv", C.c_void_p), ("T", C.c_void_p)]


class _TracesOut(C.Structure):
    _fields_ = [("ncoarse", C.c_void_p), ("p_loc", C.c_void_p), ("cmass", C.c_void_p), ("functional", C.c_void_p)]


class _ExtIn(C.Structure):
    _fields_ = [("n_ent", C.c_int), ("nT", C.c_int), ("mode", C.c_int), ("null_dofs", C.c_int), ("svd_tol", C.c_double), ("smallest_entry", C.c_double),
                ("nu", C.c_void_p), ("ni", C.c_void_p), ("np", C.c_void_p), ("npi", C.c_void_p), ("nq", C.c_void_p),
                ("next", C.c_void_p), ("nrt", C.c_void_p),
                ("mass_u", _Assembly), ("mass_p", _Assembly), ("mass_q", _Assembly),
                ("d_ptr", C.c_void_p), ("d_row", C.c_void_p), ("d_col", C.c_void_p), ("d_val", C.c_void_p),
                ("d2_ptr", C.c_void_p), ("d2_row", C.c_void_p), ("d2_col", C.c_void_p), ("d2_val", C.c_void_p),
                ("pv_p", C.c_void_p), ("R_b", C.c_void_p), ("cP", C.c_void_p), ("T", C.c_void_p), ("PDc", C.c_void_p)]


class _ExtOut(C.Structure):
    _fields_ = [("n_null", C.c_void_p), ("ext", C.c_void_p), ("dc", C.c_void_p), ("rt", C.c_void_p), ("nul", C.c_void_p),
                ("functional", C.c_void_p), ("cmass", C.c_void_p)]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class _Keep:
    """Holds the numpy arrays behind the raw pointers of a ctypes struct alive."""
    def __init__(self):
        self.refs = []

    def i32(self, a):
        a = np.ascontiguousarray(a, dtype=np.int32); self.refs.append(a); return _ptr(a)

    def i64(self, a):
        a = np.ascontiguousarray(a, dtype=np.int64); self.refs.append(a); return _ptr(a)

    def f64(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64); self.refs.append(a); return _ptr(a)


def _assembly(keep: _Keep, ent_of_item, n_ent, blk_off, blk_n, loc_idx, blocks) -> _Assembly:
    """Items must be sorted by entity.  ``loc_idx`` is the concatenation of the items' local index lists."""
    blk_n = np.asarray(blk_n, dtype=np.int64)
    asm_ptr = ptr_from_counts(np.bincount(ent_of_item, minlength=n_ent))
    loc_off = ptr_from_counts(blk_n)[:-1]
    return _Assembly(keep.i64(asm_ptr), keep.i64(blk_off), keep.i32(blk_n), keep.i64(loc_off), keep.i32(loc_idx), keep.f64(blocks),
                     int(blk_n.size), int(np.asarray(loc_idx).size), int(np.asarray(blocks).size))


def _blocks_to_csr(rows_ptr, rows_idx, cols_ptr, cols_idx, vals, ld_cols, shape, col_major, slot_rows=None):
    """Sparse matrix whose entity blocks are dense: block e couples rows ``rows_idx[rows_ptr[e]:..]`` with columns
    ``cols_idx[cols_ptr[e]:..]``; its values are stored in ``vals`` as consecutive slots of ``nrows_e * ld_cols[e]``
    entries (col-major: column c at ``c * nrows_e``; row-major: row r at ``r * ld``...).  Exact zeros are dropped like
    mfem's ``SetSubMatrix``; straight to CSR with sorted columns in threaded host code (``csrc/host_amge_tables.hip``)."""
    keep = _Keep()
    n = len(rows_ptr) - 1
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    fam = (_BlockFamily * 1)()
    f = fam[0]
    f.n_ent, f.n_vals, f.col_major = n, vals.size, 1 if col_major else 0
    f.rows_ptr, f.rows_idx, f.cols_ptr, f.cols_idx = keep.i64(rows_ptr), keep.i64(rows_idx), keep.i64(cols_ptr), keep.i64(cols_idx)
    f.vals, f.ld = _ptr(vals), keep.i64(np.broadcast_to(ld_cols, (n,)))
    f.slot_rows = None if slot_rows is None else keep.i64(np.broadcast_to(slot_rows, (n,)))
    return _families_to_csr(fam, shape)


def _agglomerate_tables_numpy(ndof2, nAE, nAF, ae_ptr, els, e2_ptr, e2_idx, af_dofs, af_of_afdof, locf, af_d_ptr, B_ptr, AFs, cf_ptr, k2, o_pf,
                              ae_d3_ptr, ae_dofs3, D) -> dict:
    """Whole-array numpy statement of :func:`_agglomerate_tables` (the tests compare the two array by array)."""
    ae_of_item = np.repeat(np.arange(nAE, dtype=np.int64), np.diff(ae_ptr))
    ncf = np.diff(cf_ptr)
    sep = np.zeros(ndof2, dtype=np.int64)
    sep[af_dofs] = 1
    n2e = np.diff(e2_ptr)
    rd, ownr = expand_ranges(e2_ptr[els], n2e[els])              # rdofs (element-local copies) in agglomerate order
    rd_dof = e2_idx[rd]
    rd_ae = ae_of_item[ownr]
    big = 2 * ndof2
    key = rd_ae * big + sep[rd_dof] * ndof2 + rd_dof             # order of reorderRow: (agglomerate, separator type, dof)
    ukey, loc_of_rd = np.unique(key, return_inverse=True)
    ae_of_u = ukey // big
    ae_dofs2 = ukey % ndof2
    sep_u = (ukey % big) // ndof2
    ae_d2_ptr = ptr_from_counts(np.bincount(ae_of_u, minlength=nAE))
    ni = np.bincount(ae_of_u, weights=(sep_u == 0), minlength=nAE).astype(np.int64)
    nu = np.diff(ae_d2_ptr)
    loc2 = loc_of_rd.reshape(-1) - ae_d2_ptr[rd_ae]
    # coarse trace dofs on the boundary of an agglomerate: facets of the coarse element-facet row, dofs facet by facet
    ae_of_B = np.repeat(np.arange(nAE, dtype=np.int64), np.diff(B_ptr))
    cbd, ownb = expand_ranges(cf_ptr[AFs], ncf[AFs])
    cbd_ptr = ptr_from_counts(np.bincount(ae_of_B, weights=ncf[AFs], minlength=nAE).astype(np.int64))
    nxt = np.diff(cbd_ptr)
    # R_b: traces on the boundary dofs of the agglomerate, col-major (nb x next).  Trace values per (fine dof, local coarse
    # column of its AF) come from p_loc of the facet kernel.
    af_of_dof = np.full(ndof2, -1, dtype=np.int64); af_of_dof[af_dofs] = af_of_afdof
    pos_in_af = np.zeros(ndof2, dtype=np.int64); pos_in_af[af_dofs] = locf
    nb = nu - ni
    Rb_ptr = ptr_from_counts(nb * nxt)
    R_b = np.zeros(int(Rb_ptr[-1]))
    # for every coarse boundary column (ae, j): its AF F, its index c within the AF; rows = boundary dofs of ae lying on F
    col_ae = ae_of_B[ownb]; col_F = AFs[ownb]; col_c = cbd - cf_ptr[col_F]
    col_j = np.arange(cbd.size, dtype=np.int64) - cbd_ptr[col_ae]
    bsel = np.nonzero(sep_u == 1)[0]
    b_ae = ae_of_u[bsel]; b_dof = ae_dofs2[bsel]; b_loc = bsel - ae_d2_ptr[b_ae] - ni[b_ae]; b_F = af_of_dof[b_dof]
    # join (ae, F) of columns with (ae, F) of boundary dofs
    ckey = col_ae * nAF + col_F
    bkey = b_ae * nAF + b_F
    so = np.argsort(bkey, kind="stable")
    bkey_s = bkey[so]
    lo = np.searchsorted(bkey_s, ckey, side="left"); hi = np.searchsorted(bkey_s, ckey, side="right")
    pair_b, pair_col = expand_ranges(lo, hi - lo)
    pb = so[pair_b]
    src = af_d_ptr[b_F[pb]] * k2 + col_c[pair_col] * np.diff(af_d_ptr)[b_F[pb]] + pos_in_af[b_dof[pb]]
    dst = Rb_ptr[col_ae[pair_col]] + col_j[pair_col] * nb[col_ae[pair_col]] + b_loc[pb]
    R_b[dst] = o_pf[src]
    # local derivative triplets: rows = p dofs of the agglomerate (position in ae_dofs3), cols = u dofs (position in ae_dofs2)
    prow_ae = np.repeat(np.arange(nAE, dtype=np.int64), np.diff(ae_d3_ptr))
    cnt_row = np.diff(D.indptr)[ae_dofs3]
    dq, own_row = expand_ranges(D.indptr[ae_dofs3], cnt_row)
    d_ae = prow_ae[own_row]
    d_row = own_row - ae_d3_ptr[d_ae]
    dcol_glob = D.indices[dq].astype(np.int64)
    dkey = d_ae * big + sep[dcol_glob] * ndof2 + dcol_glob
    pos = np.searchsorted(ukey, dkey)
    if not np.array_equal(ukey[np.minimum(pos, ukey.size - 1)], dkey):
        raise ValueError("DistributeAgglomerateMatrix: D_2 couples an agglomerate with a dof outside it")
    d_col = pos - ae_d2_ptr[d_ae]
    d_ptr = ptr_from_counts(np.bincount(d_ae, minlength=nAE))
    return dict(ae_d2_ptr=ae_d2_ptr, ae_dofs2=ae_dofs2, ni=ni, loc2=loc2, cbd_ptr=cbd_ptr, cbd=cbd, R_b=R_b, d_ptr=d_ptr,
                d_row=d_row.astype(np.int32), d_col=d_col.astype(np.int32), d_val=D.data[dq])


def _agglomerate_tables(ndof2, nAE, nAF, ae_ptr, els, e2_ptr, e2_idx, af_dofs, af_of_afdof, locf, af_d_ptr, B_ptr, AFs, cf_ptr, k2, o_pf,
                        ae_d3_ptr, ae_dofs3, D) -> dict:
    """The index tables between the facet stage and the extension stage of ``Coarsen``, agglomerate by agglomerate in threaded
    host code (``csrc/host_amge_tables.hip``): DofAgglomeration of form 2 (``DOFAgglomeration.cpp:86-131``: dofs of every
    agglomerate, interior first, and the local index of every element dof), the coarse trace columns on its boundary with
    the traces restricted to its boundary dofs (``DeRhamSequence.cpp:2214-2290``), and the local triplets of ``D_2``
    (``DistributeAgglomerateMatrix``)."""
    Lh = _lib.load()

    def chk(status):                                            # same exception as the numpy statement
        try:
            _lib.check(status)
        except _lib.ElagHipError as e:
            raise ValueError(str(e)) from None

    c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    ae_ptr, els, e2_ptr, e2_idx, af_d_ptr, B_ptr, AFs, cf_ptr, ae_d3_ptr, ae_dofs3 = map(c64, (ae_ptr, els, e2_ptr, e2_idx, af_d_ptr, B_ptr, AFs, cf_ptr,
                                                                                                 ae_d3_ptr, ae_dofs3))
    sep = np.zeros(ndof2, dtype=np.int64); sep[af_dofs] = 1
    af_of_dof = np.full(ndof2, -1, dtype=np.int64); af_of_dof[af_dofs] = af_of_afdof
    pos_in_af = np.zeros(ndof2, dtype=np.int64); pos_in_af[af_dofs] = locf
    n_rd = int(np.diff(e2_ptr)[els].sum())
    ae_d2_ptr = np.empty(nAE + 1, dtype=np.int64); ni = np.empty(nAE, dtype=np.int64)
    dofs_buf = np.empty(n_rd, dtype=np.int64); loc2 = np.empty(n_rd, dtype=np.int64)
    chk(Lh.elag_hip_host_amge_dof_agglomeration(nAE, _ptr(ae_ptr), _ptr(els), _ptr(e2_ptr), _ptr(e2_idx), ndof2, _ptr(sep), _ptr(ae_d2_ptr),
                                                       _ptr(ni), _ptr(dofs_buf), _ptr(loc2)))
    ae_dofs2 = dofs_buf[:int(ae_d2_ptr[-1])].copy()
    cbd_ptr = np.empty(nAE + 1, dtype=np.int64); Rb_ptr = np.empty(nAE + 1, dtype=np.int64)
    o_pf = np.ascontiguousarray(o_pf, dtype=np.float64)
    args = (nAE, _ptr(B_ptr), _ptr(AFs), _ptr(cf_ptr), _ptr(af_of_dof), _ptr(pos_in_af), _ptr(af_d_ptr), int(k2), _ptr(o_pf), _ptr(ae_d2_ptr),
            _ptr(ae_dofs2), _ptr(ni), _ptr(cbd_ptr), _ptr(Rb_ptr))
    chk(Lh.elag_hip_host_amge_boundary_traces(*args, None, None))
    cbd = np.empty(int(cbd_ptr[-1]), dtype=np.int64); R_b = np.empty(int(Rb_ptr[-1]))
    chk(Lh.elag_hip_host_amge_boundary_traces(*args, _ptr(cbd), _ptr(R_b)))
    indptr = np.ascontiguousarray(D.indptr, dtype=np.int32); indices = np.ascontiguousarray(D.indices, dtype=np.int32)
    data = np.ascontiguousarray(D.data, dtype=np.float64)
    d_ptr = np.empty(nAE + 1, dtype=np.int64)
    dargs = (nAE, _ptr(ae_d3_ptr), _ptr(ae_dofs3), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(ae_d2_ptr), _ptr(ae_dofs2), _ptr(ni), _ptr(d_ptr))
    chk(Lh.elag_hip_host_amge_local_derivative(*dargs, None, None, None))
    nd = int(d_ptr[-1])
    d_row = np.empty(nd, dtype=np.int32); d_col = np.empty(nd, dtype=np.int32); d_val = np.empty(nd)
    chk(Lh.elag_hip_host_amge_local_derivative(*dargs, _ptr(d_row), _ptr(d_col), _ptr(d_val)))
    return dict(ae_d2_ptr=ae_d2_ptr, ae_dofs2=ae_dofs2, ni=ni, loc2=loc2, cbd_ptr=cbd_ptr, cbd=cbd, R_b=R_b, d_ptr=d_ptr, d_row=d_row, d_col=d_col,
                d_val=d_val)


class _BlockFamily(C.Structure):
    _fields_ = [("n_ent", C.c_int64), ("rows_ptr", C.c_void_p), ("rows_idx", C.c_void_p), ("cols_ptr", C.c_void_p), ("cols_idx", C.c_void_p),
                ("vals", C.c_void_p), ("n_vals", C.c_int64), ("ld", C.c_void_p), ("slot_rows", C.c_void_p), ("col_major", C.c_int)]


def _blocks_to_csr_sets(sets, shape):
    """One matrix from several families of col-major dense blocks ``(rows_ptr, rows_idx, cols_ptr, cols_idx, vals, ld_cols)`` (see
    :func:`_blocks_to_csr`) in one pass (``elag_hip_host_block_families_to_csr``); entries that meet are added family by family."""
    keep = _Keep()
    fam = (_BlockFamily * len(sets))()
    for f, (rows_ptr, rows_idx, cols_ptr, cols_idx, vals, ld_cols) in zip(fam, sets):
        n = len(rows_ptr) - 1
        vals = np.ascontiguousarray(vals, dtype=np.float64); keep.refs.append(vals)
        f.n_ent, f.n_vals, f.col_major, f.slot_rows = n, vals.size, 1, None
        f.rows_ptr, f.rows_idx, f.cols_ptr, f.cols_idx = keep.i64(rows_ptr), keep.i64(rows_idx), keep.i64(cols_ptr), keep.i64(cols_idx)
        f.vals, f.ld = _ptr(vals), keep.i64(np.broadcast_to(ld_cols, (n,)))
    return _families_to_csr(fam, shape)


def _families_to_csr(fam, shape):
    Lh = _lib.load()

    def chk(status):
        try:
            _lib.check(status)
        except _lib.ElagHipError as e:
            if "shorter than its slots" in str(e):
                raise ValueError("_blocks_to_csr: the value array is shorter than its slots") from None
            raise
    indptr = np.empty(shape[0] + 1, dtype=np.int64)
    nnz = np.zeros(1, dtype=np.int64)
    chk(Lh.elag_hip_host_block_families_to_csr(len(fam), C.byref(fam), int(shape[0]), int(shape[1]), _ptr(indptr), None, None, None))
    cap = int(indptr[-1])
    if cap >= 2**31:
        raise ValueError("_blocks_to_csr: more than 2^31 entries")
    indices, data = np.empty(cap, dtype=np.int32), np.empty(cap)
    chk(Lh.elag_hip_host_block_families_to_csr(len(fam), C.byref(fam), int(shape[0]), int(shape[1]), _ptr(indptr), _ptr(indices), _ptr(data), _ptr(nnz)))
    n = int(nnz[0])
    M = sp.csr_matrix((data[:n], indices[:n], indptr.astype(np.int32)), shape=shape)
    M.has_sorted_indices = True
    return M


# ---- Coarsen ----------------------------------------------------------------------------------------------------------------

def coarsen(lv: AmgeLevel, part: np.ndarray, svd_tol: float = 1e-9, smallest_entry: float = EPS, timings: dict | None = None) -> dict:
    """``DeRhamSequence::Coarsen`` for forms 2 and 3.  Returns dict(P2, P3, Pi2 parts, coarse (AmgeLevel), topo)."""
    with _lib.host_pages_kept():          # (the table stage's allocator policy, for the duration of this call only)
        return _coarsen_impl(lv, part, svd_tol=svd_tol, smallest_entry=smallest_entry, timings=timings)


def _coarsen_impl(lv: AmgeLevel, part: np.ndarray, svd_tol: float = 1e-9, smallest_entry: float = EPS, timings: dict | None = None) -> dict:

    L = _lib.init()
    t_mark = [time.perf_counter()]

    def mark(name):                                             # host wall time of the stage that just ended (seconds)
        if timings is not None:
            now = time.perf_counter(); timings["host_" + name + "_s"] = now - t_mark[0]; t_mark[0] = now

    topo = coarsen_topology(lv, part)
    mark("topology")
    nAE, nAF = topo["nAE"], topo["nAF"]
    nT2, nT3 = lv.T2.shape[1], lv.T3.shape[1]
    ae_of_elem = np.asarray(part, dtype=np.int64)
    els = topo["ae_idx"]                                        # elements in agglomerate order
    ae_of_item = ae_of_elem[els]

    # ===== form 3 on the agglomerated elements: dofs of an AE = dofs of its elements, element after element =====
    n3e = np.diff(lv.e_dof3_ptr)
    ae_dofs3, own3 = expand_ranges(lv.e_dof3_ptr[els], n3e[els])
    ae_d3_ptr = ptr_from_counts(np.bincount(ae_of_item, weights=n3e[els], minlength=nAE).astype(np.int64))
    # local index of every element dof inside its agglomerate = position in the concatenation
    loc3 = np.arange(ae_dofs3.size, dtype=np.int64) - ae_d3_ptr[ae_of_item[own3]]
    keep = _Keep()
    tin = _TracesIn(nAE, nT3, svd_tol, keep.i32(np.diff(ae_d3_ptr)),
                    _assembly(keep, ae_of_item, nAE, lv.M3e_ptr[els], n3e[els], loc3, lv.M3e),
                    keep.f64(lv.pv3[ae_dofs3]), keep.f64(_pack_cols(lv.T3, ae_dofs3, ae_d3_ptr)))
    k3 = 1 + nT3
    o_nc3 = np.zeros(nAE, np.int32); o_p3 = np.zeros(ae_dofs3.size * k3); o_cm3 = np.zeros(nAE * k3 * k3); o_f3 = np.zeros(ae_dofs3.size * k3)
    ms = C.c_double(0.0)
    _lib.check(L.elag_hip_coarsen_traces(C.byref(tin), C.byref(_TracesOut(_ptr(o_nc3), _ptr(o_p3), _ptr(o_cm3), _ptr(o_f3))), C.byref(ms)))
    if timings is not None:
        timings["traces3_ms"] = ms.value
    n3c = o_nc3.astype(np.int64)
    c3_ptr = ptr_from_counts(n3c)
    ndof3c = int(c3_ptr[-1])
    c3_idx = np.arange(ndof3c, dtype=np.int64)
    P3 = _blocks_to_csr(ae_d3_ptr, ae_dofs3, c3_ptr, c3_idx, o_p3, np.full(nAE, k3), (lv.ndof3, ndof3c), col_major=True)
    # functional slots are (1+nT) x ndof row-major: rows = coarse dofs
    Pi3 = _blocks_to_csr(c3_ptr, c3_idx, ae_d3_ptr, ae_dofs3, o_f3, np.diff(ae_d3_ptr), (ndof3c, lv.ndof3), col_major=False,
                         slot_rows=np.full(nAE, k3))
    M3c_blocks = _take_leading(o_cm3, np.full(nAE, k3), n3c)

    mark("form3")
    # ===== form 2 on the agglomerated facets =====
    n2f = np.diff(lv.f_dof2_ptr)
    fac = topo["af_idx"]
    af_of_item = np.repeat(np.arange(nAF, dtype=np.int64), np.diff(topo["af_ptr"]))
    af_dofs, ownf = expand_ranges(lv.f_dof2_ptr[fac], n2f[fac])
    af_d_ptr = ptr_from_counts(np.bincount(af_of_item, weights=n2f[fac], minlength=nAF).astype(np.int64))
    locf = np.arange(af_dofs.size, dtype=np.int64) - af_d_ptr[af_of_item[ownf]]
    pv_f = lv.pv2[af_dofs] * topo["af_sgn"][ownf]
    keep2 = _Keep()
    tin2 = _TracesIn(nAF, nT2, svd_tol, keep2.i32(np.diff(af_d_ptr)),
                     _assembly(keep2, af_of_item, nAF, lv.M2f_ptr[fac], n2f[fac], locf, lv.M2f),
                     keep2.f64(pv_f), keep2.f64(_pack_cols(lv.T2, af_dofs, af_d_ptr)))
    k2 = 1 + nT2
    o_ncf = np.zeros(nAF, np.int32); o_pf = np.zeros(af_dofs.size * k2); o_cmf = np.zeros(nAF * k2 * k2); o_ff = np.zeros(af_dofs.size * k2)
    _lib.check(L.elag_hip_coarsen_traces(C.byref(tin2), C.byref(_TracesOut(_ptr(o_ncf), _ptr(o_pf), _ptr(o_cmf), _ptr(o_ff))), C.byref(ms)))
    if timings is not None:
        timings["traces2_ms"] = ms.value
    ncf = o_ncf.astype(np.int64)
    cf_ptr = ptr_from_counts(ncf)
    ntrace = int(cf_ptr[-1])
    cf_idx = np.arange(ntrace, dtype=np.int64)
    M2fc_blocks = _take_leading(o_cmf, np.full(nAF, k2), ncf)

    mark("facets2")
    # ===== DofAgglomeration of form 2 on the agglomerated elements (DOFAgglomeration.cpp:86-131), boundary traces, local D =====
    e2_ptr, e2_idx = lv.elem_dof2_table()
    n2e = np.diff(e2_ptr)
    D = lv.D2.tocsr()
    tables = _agglomerate_tables_numpy if os.environ.get("ELAG_HIP_AMGE_NUMPY_TABLES") else _agglomerate_tables
    tb = tables(lv.ndof2, nAE, nAF, topo["ae_ptr"], els, e2_ptr, e2_idx, af_dofs, af_of_item[ownf], locf, af_d_ptr, topo["B_ptr"], topo["B_idx"],
                cf_ptr, k2, o_pf, ae_d3_ptr, ae_dofs3, D)
    ae_d2_ptr, ae_dofs2, ni, loc2 = tb["ae_d2_ptr"], tb["ae_dofs2"], tb["ni"], tb["loc2"]
    cbd_ptr, cbd, R_b = tb["cbd_ptr"], tb["cbd"], tb["R_b"]
    d_ptr, d_row, d_col, d_val = tb["d_ptr"], tb["d_row"], tb["d_col"], tb["d_val"]
    nu = np.diff(ae_d2_ptr)
    nxt = np.diff(cbd_ptr)
    mark("tables2")
    # pv_p and cP from P_3 (packed p_loc of the element kernel: col-major np x k3 per agglomerate)
    np_ = np.diff(ae_d3_ptr)
    pv_p, _ = expand_ranges(ae_d3_ptr[:-1] * k3, np_)
    pv_p = o_p3[pv_p]
    nrt = n3c - 1
    cP_ptr = ptr_from_counts(np_ * nrt)
    csrc, owncp = expand_ranges(ae_d3_ptr[:-1] * k3 + np_, np_ * nrt)
    cP = o_p3[csrc]
    keep3 = _Keep()
    ein = _ExtIn(nAE, nT2, 0, 1, svd_tol, smallest_entry, keep3.i32(nu), keep3.i32(ni), keep3.i32(np_), None, None, keep3.i32(nxt), keep3.i32(nrt),
                 _assembly(keep3, ae_of_item, nAE, lv.M2e_ptr[els], n2e[els], loc2, lv.M2e),
                 _assembly(keep3, ae_of_item, nAE, lv.M3e_ptr[els], n3e[els], loc3, lv.M3e), _Assembly(),
                 keep3.i64(d_ptr), keep3.i32(d_row), keep3.i32(d_col), keep3.f64(d_val), None, None, None, None,
                 keep3.f64(pv_p), keep3.f64(R_b), keep3.f64(cP), keep3.f64(_pack_cols(lv.T2, ae_dofs2, ae_d2_ptr)), None)
    kf = nrt + nT2
    ncm = nxt + nrt + nT2
    o_nn = np.zeros(nAE, np.int32)
    o_ext = np.zeros(int((ni * nxt).sum())); o_dc = np.zeros(int(nxt.sum())); o_rt = np.zeros(int((ni * nrt).sum()))
    o_nul = np.zeros(int((ni * nT2).sum())); o_fun = np.zeros(int((kf * ni).sum())); o_cm = np.zeros(int((ncm * ncm).sum()))
    eout = _ExtOut(_ptr(o_nn), _ptr(o_ext), _ptr(o_dc), _ptr(o_rt), _ptr(o_nul), _ptr(o_fun), _ptr(o_cm))
    _lib.check(L.elag_hip_coarsen_extension(C.byref(ein), C.byref(eout), C.byref(ms)))
    if timings is not None:
        timings["extension_ms"] = ms.value
    n_null = o_nn.astype(np.int64)
    nint_c = nrt + n_null
    ci_ptr = ptr_from_counts(nint_c)
    ndof2c = ntrace + int(ci_ptr[-1])

    mark("extension2")
    # ===== assemble P_2, the coarse derivative and the projector pieces =====
    int_ptr = ptr_from_counts(ni)
    int_dofs, own_i = expand_ranges(ae_d2_ptr[:-1], ni)
    int_dofs = ae_dofs2[int_dofs]                                   # interior fine dofs, agglomerate by agglomerate
    ci_idx = ntrace + np.arange(int(ci_ptr[-1]), dtype=np.int64)
    # P_2 = [traces on the agglomerated facets | extensions into the interiors | bubbles: RangeT columns, null-space columns]: four families of
    # dense blocks with disjoint (row, column) sets, assembled in one pass
    rt_rows, own_rt = expand_ranges(c3_ptr[:-1] + 1, nrt)
    rt_cols, _ = expand_ranges(ntrace + ci_ptr[:-1], nrt)
    nul_cols, _ = expand_ranges(ntrace + ci_ptr[:-1] + nrt, n_null)
    P2 = _blocks_to_csr_sets([(af_d_ptr, af_dofs, cf_ptr, cf_idx, o_pf, np.full(nAF, k2)),
                              (int_ptr, int_dofs, cbd_ptr, cbd, o_ext, nxt),
                              (int_ptr, int_dofs, ptr_from_counts(nrt), rt_cols, o_rt, nrt),
                              (int_ptr, int_dofs, ptr_from_counts(n_null), nul_cols, o_nul, np.full(nAE, nT2))], (lv.ndof2, ndof2c))
    # coarse D_2: row of the PV dof of the agglomerate, unit entries for the RangeT bubbles (:2408, :2441)
    dsel = o_dc != 0.0
    rows = np.r_[np.repeat(c3_ptr[:-1], nxt)[dsel], rt_rows]
    cols = np.r_[cbd[dsel], rt_cols]
    D2c = sp.csr_matrix((np.r_[o_dc[dsel], np.ones(rt_rows.size)], (rows, cols)), shape=(ndof3c, ndof2c))
    D2c.sort_indices()
    hat1 = _blocks_to_csr(cf_ptr, cf_idx, af_d_ptr, af_dofs, o_ff, np.diff(af_d_ptr), (ndof2c, lv.ndof2), col_major=False,
                          slot_rows=np.full(nAF, k2))
    hat0 = _blocks_to_csr(ci_ptr, ci_idx, int_ptr, int_dofs, _take_leading_rows(o_fun, kf, ni, nint_c), ni, (ndof2c, lv.ndof2), col_major=False)

    mark("assemble2")
    # ===== project the targets: CochainProjector::Project (codim 1 then 0) with device products =====
    T2c = _project(hat1, hat0, P2, lv.T2)
    T3c = _spmm_device(Pi3, np.c_[lv.T3, lv.const3])
    T3c, const3c = np.ascontiguousarray(T3c[:, :-1]), np.ascontiguousarray(T3c[:, -1])

    mark("project")
    M2ec_blocks = _take_leading(o_cm, ncm, nxt + nint_c)
    n2ec = nxt + nint_c
    coarse = AmgeLevel(
        nE=nAE, nF=nAF, ef_ptr=topo["B_ptr"], ef_idx=topo["B_idx"], ef_sgn=topo["B_sgn"], facet_attr=topo["facet_attr"],
        ndof3=ndof3c, e_dof3_ptr=c3_ptr, ndof2=ndof2c, f_dof2_ptr=cf_ptr, e_idof2_ptr=ci_ptr,
        M3e_ptr=ptr_from_counts(n3c * n3c), M3e=M3c_blocks,
        M2e_ptr=ptr_from_counts(n2ec * n2ec), M2e=M2ec_blocks,
        M2f_ptr=ptr_from_counts(ncf * ncf), M2f=M2fc_blocks,
        D2=D2c, T2=T2c, T3=T3c, const3=const3c,
        pv2=_first_dof_indicator(cf_ptr, ndof2c), pv3=_first_dof_indicator(c3_ptr, ndof3c), meta=dict(kind="alg"))
    mark("coarse_level")
    return dict(P2=P2, P3=P3, Pi3=Pi3, hat1=hat1, hat0=hat0, coarse=coarse, topo=topo, n_rt=nrt, n_null=n_null,
                agg=dict(ae_d2_ptr=ae_d2_ptr, ae_dofs2=ae_dofs2, ni=ni, ae_d3_ptr=ae_d3_ptr, ae_dofs3=ae_dofs3,
                         af_d_ptr=af_d_ptr, af_dofs=af_dofs))


def projector2(res: dict) -> sp.csr_matrix:
    """``CochainProjector::ComputeProjector`` (``CochainProjector.cpp:218-261``): ``Pi = hat1 + hat0 (I - P hat1)`` as a
    sparse matrix (device SpGEMM / add)."""
    from . import linalg as la
    h1 = la.ParallelCSRMatrix.from_scipy(res["hat1"]); h0 = la.ParallelCSRMatrix.from_scipy(res["hat0"]); P = la.ParallelCSRMatrix.from_scipy(res["P2"])
    hPPi = h0.matmul(P.matmul(h1))
    out = la.Add(1.0, la.Add(1.0, h1, 1.0, h0), -1.0, hPPi).to_scipy()
    out.sort_indices()
    return out


def coarsen_sequence(lv: AmgeLevel, parts: list, **kw) -> list:
    out = []
    for part in parts:
        res = coarsen(lv, part, **kw)
        out.append(res)
        lv = res["coarse"]
    return out


# ---- small helpers ------------------------------------------------------------------------------------------------------------

def _pack_cols(T, dofs, ptr):
    """Per entity the rows ``dofs[ptr[e]:ptr[e+1]]`` of ``T`` stored col-major, entities one after the other."""
    nT = T.shape[1]
    if nT == 0 or dofs.size == 0:
        return np.zeros(0)
    ptr, dofs = np.ascontiguousarray(ptr, dtype=np.int64), np.ascontiguousarray(dofs, dtype=np.int64)
    T = np.ascontiguousarray(T, dtype=np.float64)
    out = np.empty(dofs.size * nT)
    _lib.check(_lib.load().elag_hip_host_pack_cols(len(ptr) - 1, _ptr(ptr), _ptr(dofs), T.shape[0], nT, _ptr(T), _ptr(out)))
    return out


def _take_leading(vals, ld, k):
    """From consecutive row-major ``ld[e] x ld[e]`` slots keep the leading ``k[e] x k[e]`` blocks, packed."""
    ld = np.ascontiguousarray(ld, dtype=np.int64); k = np.ascontiguousarray(k, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    if vals.size < int((ld * ld).sum()):
        raise ValueError("_take_leading: the value array is shorter than its slots")
    out = np.empty(int((k * k).sum()))
    _lib.check(_lib.load().elag_hip_host_take_leading(k.size, _ptr(ld), _ptr(k), _ptr(vals), _ptr(out)))
    return out


def _take_leading_rows(vals, nrows_max, ncols, nrows):
    """From consecutive row-major ``nrows_max[e] x ncols[e]`` slots keep the first ``nrows[e]`` rows, packed."""
    slot = ptr_from_counts(nrows_max * ncols)
    idx, _ = expand_ranges(slot[:-1], nrows * ncols)
    return vals[idx]


def _join_cols(a, nrows, ka, b, kb_max, kb):
    """Per entity ``[A_e | B_e[:, :kb_e]]`` with ``A_e`` (nrows x ka) and ``B_e`` (nrows x kb_max) col-major, packed col-major."""
    pa = ptr_from_counts(nrows * ka); pb = ptr_from_counts(nrows * kb_max)
    ia, oa = expand_ranges(pa[:-1], nrows * ka)
    ib, ob = expand_ranges(pb[:-1], nrows * kb)
    out_ptr = ptr_from_counts(nrows * (ka + kb))
    out = np.empty(int(out_ptr[-1]))
    da = out_ptr[oa] + (ia - pa[oa])
    db = out_ptr[ob] + (nrows * ka)[ob] + (ib - pb[ob])
    out[da] = a[ia]; out[db] = b[ib]
    return out


def _first_dof_indicator(ptr, n):
    v = np.zeros(n)
    v[ptr[:-1][np.diff(ptr) > 0]] = 1.0
    return v


def _spmm_device(A: sp.csr_matrix, X: np.ndarray) -> np.ndarray:
    if A.nnz == 0 or X.shape[1] == 0:
        return np.zeros((A.shape[0], X.shape[1]))
    from . import linalg as la
    dA = la.ParallelCSRMatrix.from_scipy(A.tocsr())
    out = np.empty((A.shape[0], X.shape[1]))
    y = np.empty(A.shape[0])
    for c in range(X.shape[1]):
        dA.Mult(np.ascontiguousarray(X[:, c]), y)
        out[:, c] = y
    return out


def _project(hat1, hat0, P2, T):
    """``CochainProjector::Project`` (``CochainProjector.cpp:158-216``): facet functionals, residual, element functionals; the three
    products of a column run on device vectors (one upload of the column, one download of the result)."""
    out = np.zeros((hat1.shape[0], T.shape[1]))
    if T.shape[1] == 0:
        return out
    from . import linalg as la
    d1, dP, d0 = (la.ParallelCSRMatrix.from_scipy(M) if M.nnz else None for M in (hat1, P2, hat0))
    for c in range(T.shape[1]):
        t = la.DeviceVector.from_numpy(np.ascontiguousarray(T[:, c]))
        v1 = la.DeviceVector(hat1.shape[0], fill=0.0)
        if d1 is not None:
            d1.Mult(t, v1)                                        # v1 = hat1 T
        if dP is not None:
            dP.Mult(v1, t, alpha=-1.0, beta=1.0)                  # t = T - P v1
        if d0 is not None:
            d0.Mult(t, v1, alpha=1.0, beta=1.0)                   # v1 += hat0 (T - P v1)
        out[:, c] = v1.to_numpy()
    return out
