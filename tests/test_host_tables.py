"""The native host tables of the coarsening (``csrc/host_tables.hip``, ``elag_hip_host_*``; no device work) against plain
restatements of the reference's loops: ``mfem::Mult`` with first-touch column order and ``MultOrientation``
(``src/topology/TopologyTable.cpp:130-139``), ``GetSubMatrix`` / ``SetSubMatrix`` of all entities at once, ``reorderRow``
(``src/amge/DOFAgglomeration.cpp:33-131``); empty and ragged inputs, unsorted queries, arguments that must be refused."""
import numpy as np
import pytest
import scipy.sparse as sp

from parelag_amd import _lib, amge, derham
from parelag_amd.amge_level import ptr_from_counts
from parelag_amd.derham_level import _product, bool_mult


def ragged(rng, nrow, ncol, maxlen, signed=False):
    cnt = rng.integers(0, maxlen + 1, nrow)
    idx = np.concatenate([rng.choice(ncol, c, replace=False) for c in cnt]) if cnt.sum() else np.zeros(0, dtype=np.int64)
    out = (ptr_from_counts(cnt), idx.astype(np.int64))
    return out + ((rng.choice([-1.0, 1.0], idx.size),) if signed else ())


def mult_orientation(a, b):
    """Row loop with a dict: insertion order = first touch; cancelled sums leave the row."""
    (ap, ai, asg), (bp, bi, bsg) = a, b
    ptr, idx, sgn = [0], [], []
    for r in range(len(ap) - 1):
        acc = {}
        for k in range(ap[r], ap[r + 1]):
            for q in range(bp[ai[k]], bp[ai[k] + 1]):
                acc[int(bi[q])] = acc.get(int(bi[q]), 0.0) + asg[k] * bsg[q]
        for c, v in acc.items():
            if abs(v) > 1e-10:
                idx.append(c); sgn.append(np.sign(v))
        ptr.append(len(idx))
    return np.array(ptr), np.array(idx, dtype=np.int64), np.array(sgn)


@pytest.mark.parametrize("seed", range(4))
def test_table_products(seed):
    rng = np.random.default_rng(seed)
    a, b = ragged(rng, 40, 30, 6, signed=True), ragged(rng, 30, 25, 5, signed=True)
    p, i, s = _product(a[0], a[1], b[0], b[1], a[2], b[2])
    rp, ri, rs = mult_orientation(a, b)
    assert np.array_equal(p, rp) and np.array_equal(i, ri) and np.array_equal(s, rs)
    p, i, _ = _product(a[0], a[1], b[0], b[1])
    rp, ri = bool_mult(a[0], a[1], b[0], b[1])
    assert np.array_equal(p, rp) and np.array_equal(i, ri)


def test_table_product_edge_cases():
    empty = (np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64))
    p, i, s = _product(*empty, *empty)
    assert p.tolist() == [0] and i.size == 0 and s.size == 0
    a = (np.array([0, 0, 2]), np.array([0, 1]))                       # an empty row, then a row hitting an empty row of b
    b = (np.array([0, 0, 3]), np.array([4, 2, 4]))                    # a repeated column inside one row of b
    p, i, _ = _product(*a, *b)
    assert p.tolist() == [0, 0, 2] and i.tolist() == [4, 2]
    with pytest.raises(ValueError, match="outside the right table"):
        _product(np.array([0, 1]), np.array([5]), *b)
    lib = _lib.load()
    assert lib.elag_hip_host_table_product(1, None, None, None, 0, None, None, None, 0, None, None, None, None) != 0


def test_lookup_find_sorted_and_unsorted_queries():
    rng = np.random.default_rng(3)
    ptr, idx = ragged(rng, 50, 200, 12)
    L = derham._Lookup(ptr, idx, 199)
    row = rng.integers(0, 50, 500)
    ids = rng.integers(0, 200, 500)
    want = np.array([next((k - ptr[r] for k in range(ptr[r], ptr[r + 1]) if idx[k] == i), -1) for r, i in zip(row, ids)])
    assert np.array_equal(L.find(row, ids), want)                                          # unsorted rows: counting sort inside
    o = np.argsort(row, kind="stable")
    assert np.array_equal(L.find(row[o], ids[o]), want[o])
    assert L.find(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)).size == 0
    assert derham._Lookup(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int64), 0).find(np.zeros(0, np.int64), np.zeros(0, np.int64)).size == 0
    with pytest.raises(_lib.ElagHipError, match="outside the table"):
        L.find(np.array([50]), np.array([0]))


def test_gather_dense_blocks_to_csr_roundtrip():
    rng = np.random.default_rng(7)
    M = sp.random(60, 45, density=0.2, random_state=5, format="csr")
    rows_ptr, rows = ragged(rng, 12, 60, 7)
    cols_ptr, cols = ragged(rng, 12, 45, 6)
    ncol = np.diff(cols_ptr)
    blocks = derham._gather_dense(M, rows_ptr, rows, derham._Lookup(cols_ptr, cols, 44), ncol)
    off = ptr_from_counts(np.diff(rows_ptr) * ncol)
    Md = M.toarray()
    for e in range(12):
        r, c = rows[rows_ptr[e]:rows_ptr[e + 1]], cols[cols_ptr[e]:cols_ptr[e + 1]]
        assert np.array_equal(blocks[off[e]:off[e + 1]].reshape(len(c), len(r)).T, Md[np.ix_(r, c)])          # column-major
    # the blocks scattered back: entries covered by several blocks add up (AddSubMatrix), exact zeros are dropped
    back = amge._blocks_to_csr(rows_ptr, rows, cols_ptr, cols, blocks, ncol, M.shape, col_major=True)
    ref = np.zeros(M.shape)
    for e in range(12):
        r, c = rows[rows_ptr[e]:rows_ptr[e + 1]], cols[cols_ptr[e]:cols_ptr[e + 1]]
        ref[np.ix_(r, c)] += Md[np.ix_(r, c)]
    assert np.array_equal(back.toarray(), ref) and back.nnz == np.count_nonzero(ref)
    # row-major slots with a leading dimension larger than the block and fewer rows used than stored
    ld = ncol + 2
    slot_rows = np.diff(rows_ptr) + 1
    vals = np.zeros(int((slot_rows * ld).sum()))
    so = ptr_from_counts(slot_rows * ld)
    for e in range(12):
        r, c = rows[rows_ptr[e]:rows_ptr[e + 1]], cols[cols_ptr[e]:cols_ptr[e + 1]]
        blk = np.zeros((slot_rows[e], ld[e])); blk[:len(r), :len(c)] = Md[np.ix_(r, c)]
        vals[so[e]:so[e + 1]] = blk.ravel()
    back2 = amge._blocks_to_csr(rows_ptr, rows, cols_ptr, cols, vals, ld, M.shape, col_major=False, slot_rows=slot_rows)
    assert np.array_equal(back2.toarray(), ref)
    with pytest.raises(ValueError, match="shorter than its slots"):
        amge._blocks_to_csr(rows_ptr, rows, cols_ptr, cols, vals[:-1], ld, M.shape, col_major=False, slot_rows=slot_rows)


def test_small_helpers():
    idx, own = amge.expand_ranges(np.array([5, 0, 9]), np.array([2, 0, 3]))
    assert idx.tolist() == [5, 6, 9, 10, 11] and own.tolist() == [0, 0, 2, 2, 2]
    idx, own = amge.expand_ranges(np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64))
    assert idx.size == 0 and own.size == 0
    T = np.arange(12.0).reshape(6, 2)
    out = amge._pack_cols(T, np.array([4, 1, 5, 0]), np.array([0, 3, 4]))
    assert out.tolist() == [8, 2, 10, 9, 3, 11, 0, 1]                                    # entity 0: rows 4,1,5 column-major; entity 1: row 0
    vals = np.arange(9.0 + 4.0)
    assert amge._take_leading(vals, np.array([3, 2]), np.array([2, 1])).tolist() == [0, 1, 3, 4, 9]
    with pytest.raises(_lib.ElagHipError, match="larger than its slot"):
        amge._take_leading(vals, np.array([3, 2]), np.array([4, 1]))
    # reorderRow: every row by (key, id)
    ptr, idx = np.array([0, 4, 4, 7]), np.array([9, 2, 7, 4, 1, 8, 3])
    key = np.array([0, 1, 0, 0, 1, 0, 0, 0, 0, 1])
    _lib.check(_lib.load().elag_hip_host_sort_rows_by_key(3, amge._ptr(ptr), amge._ptr(idx), 10, amge._ptr(key)))
    assert idx.tolist() == [2, 7, 4, 9, 3, 8, 1]


def test_minimal_intersection_sets_against_the_matrix_form():
    """The native signature grouping against the oracle's restatement of ``findMinimalIntersectionSets`` on the matrix
    ``Z = AE_fc^T AE_fc`` (+ the boundary-attribute term), on random element -> facet tables of a partitioned grid."""
    from oracle import derham as od
    from parelag_amd.amge_level import box_partition
    from parelag_amd.derham_level import seq_level_from_hex
    rng = np.random.default_rng(11)
    lv = seq_level_from_hex((4, 3, 5), jstart=2)
    for trial in range(3):
        part = box_partition((4, 3, 5), (2, 2, 2)).astype(np.int64) if trial == 0 else rng.integers(0, 7, lv.n[0])
        _u, part = np.unique(part, return_inverse=True)
        t_prod, t_orc = derham.coarsen_topology(lv, part), od.coarsen_topology(lv, part)
        for c in range(4):
            for k in range(3):
                assert np.array_equal(t_prod["A"][c][k], t_orc["A"][c][k]), (trial, c, k)
        for c in range(3):
            for k in range(3):
                assert np.array_equal(t_prod["B"][c][k], t_orc["B"][c][k]), (trial, c, k)
    lib = _lib.load()
    mis, ori, n = np.empty(3, dtype=np.int64), np.empty(3), np.zeros(1, dtype=np.int64)
    fp, fi, fs = np.array([0, 2, 3]), np.array([0, 2, 2]), np.array([1.0, -1.0, 1.0])            # facet 1 on no boundary
    _lib.check(lib.elag_hip_host_minimal_intersection_sets(2, amge._ptr(fp), amge._ptr(fi), amge._ptr(fs), 3, None, amge._ptr(mis), amge._ptr(ori), amge._ptr(n)))
    assert mis.tolist() == [0, -1, 1] and ori.tolist() == [1.0, 0.0, 1.0] and n[0] == 2
    assert lib.elag_hip_host_minimal_intersection_sets(2, amge._ptr(fp), amge._ptr(np.array([0, 2, 5])), amge._ptr(fs), 3, None, amge._ptr(mis), amge._ptr(ori), amge._ptr(n)) != 0


def _agglomerate_table_inputs(n, boxes, seed):
    """The arguments ``amge.coarsen`` hands to its table stage on a hex grid, with made-up facet-stage outputs (a random number
    of coarse traces per agglomerated facet, random trace values): no device work."""
    from parelag_amd import amge_level
    from parelag_amd.amge import expand_ranges
    rng = np.random.default_rng(seed)
    lv = amge_level.level_from_hex(n)
    ijk = np.stack(np.unravel_index(np.arange(lv.nE), n[::-1]), axis=1)[:, ::-1]
    cell = [np.minimum(ijk[:, d] * boxes[d] // n[d], boxes[d] - 1) for d in range(3)]
    part = (cell[2] * boxes[1] + cell[1]) * boxes[0] + cell[0]
    topo = amge.coarsen_topology(lv, part)
    nAE, nAF = topo["nAE"], topo["nAF"]
    els = topo["ae_idx"]
    n3e = np.diff(lv.e_dof3_ptr)
    ae_dofs3, _ = expand_ranges(lv.e_dof3_ptr[els], n3e[els])
    ae_d3_ptr = ptr_from_counts(np.bincount(np.asarray(part)[els], weights=n3e[els], minlength=nAE).astype(np.int64))
    n2f = np.diff(lv.f_dof2_ptr); fac = topo["af_idx"]
    af_of_item = np.repeat(np.arange(nAF, dtype=np.int64), np.diff(topo["af_ptr"]))
    af_dofs, ownf = expand_ranges(lv.f_dof2_ptr[fac], n2f[fac])
    af_d_ptr = ptr_from_counts(np.bincount(af_of_item, weights=n2f[fac], minlength=nAF).astype(np.int64))
    locf = np.arange(af_dofs.size, dtype=np.int64) - af_d_ptr[af_of_item[ownf]]
    k2 = 3
    cf_ptr = ptr_from_counts(rng.integers(1, k2 + 1, nAF))
    o_pf = rng.standard_normal(af_dofs.size * k2)
    e2_ptr, e2_idx = lv.elem_dof2_table()
    return (lv.ndof2, nAE, nAF, topo["ae_ptr"], els, e2_ptr, e2_idx, af_dofs, af_of_item[ownf], locf, af_d_ptr, topo["B_ptr"], topo["B_idx"], cf_ptr, k2,
            o_pf, ae_d3_ptr, ae_dofs3, lv.D2.tocsr())


@pytest.mark.parametrize("n,boxes", [((6, 4, 4), (3, 2, 2)), ((5, 5, 3), (2, 2, 1)), ((4, 4, 4), (1, 1, 1)), ((4, 2, 2), (4, 2, 2))])
def test_agglomerate_tables_match_the_whole_array_statement(n, boxes):
    """Threaded per-agglomerate tables (``csrc/host_amge_tables.hip``) == the numpy composition, array by array and dtype by dtype:
    dofs of the agglomerates (interior first), local indices, boundary columns, restricted traces, local ``D`` triplets.  The last two
    cases are one agglomerate (no interior facets on its boundary but the domain's) and one element per agglomerate (no interior dofs)."""
    args = _agglomerate_table_inputs(n, boxes, seed=sum(n))
    got, ref = amge._agglomerate_tables(*args), amge._agglomerate_tables_numpy(*args)
    assert got.keys() == ref.keys()
    for k in ref:
        assert got[k].dtype == ref[k].dtype and got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), k


def test_agglomerate_tables_refuse_bad_input():
    args = list(_agglomerate_table_inputs((4, 4, 2), (2, 2, 1), seed=1))
    # D couples an agglomerate with a face outside it: an extra entry in the first row
    D = args[-1].tolil(); ndof2 = args[0]
    own = set(D.rows[0]); far = max(set(range(ndof2)) - own)
    D[0, far] = 1.0
    bad = args[:-1] + [D.tocsr()]
    for tables in (amge._agglomerate_tables, amge._agglomerate_tables_numpy):
        with pytest.raises(ValueError, match="outside it"):
            tables(*bad)
    bad = list(args); bad[6] = args[6].copy(); bad[6][0] = ndof2 + 5            # an element dof beyond the dof range
    with pytest.raises(ValueError, match="outside the dof range"):
        amge._agglomerate_tables(*bad)
    lib = _lib.load()
    assert lib.elag_hip_host_amge_dof_agglomeration(1, None, None, None, None, 0, None, None, None, None, None) != 0
    assert lib.elag_hip_host_amge_boundary_traces(1, None, None, None, None, None, None, 1, None, None, None, None, None, None, None, None) != 0
    assert lib.elag_hip_host_amge_local_derivative(1, None, None, None, None, None, None, None, None, None, None, None, None) != 0


def test_elem_dof_table_and_direct_csr_assembly():
    from parelag_amd import amge_level
    lv = amge_level.level_from_hex((5, 4, 3))
    lv.e_idof2_ptr = ptr_from_counts(np.arange(lv.nE) % 3)                         # pretend interior dofs, ragged
    for got, ref in zip(lv.elem_dof2_table(), lv.elem_dof2_table_numpy()):
        assert got.dtype == ref.dtype and np.array_equal(got, ref)
    # blocks -> CSR: overlapping blocks (entries that meet are added), exact zeros dropped, both storage orders, padded slots
    rng = np.random.default_rng(3)
    for col_major in (True, False):
        n_ent, nrow, ncol = 30, 40, 25
        rp, ri = ragged(rng, n_ent, nrow, 5); cp, ci = ragged(rng, n_ent, ncol, 4)
        nr, nc = np.diff(rp), np.diff(cp)
        ld = nc + rng.integers(0, 3, n_ent)
        sr = nr + rng.integers(0, 2, n_ent)
        slots = ptr_from_counts(sr * ld)
        vals = rng.integers(-2, 3, slots[-1]).astype(np.float64)                   # small integers: sums are exact whatever the order
        M = amge._blocks_to_csr(rp, ri, cp, ci, vals, ld, (nrow, ncol), col_major=col_major, slot_rows=sr)
        ref = np.zeros((nrow, ncol)); touched = np.zeros((nrow, ncol), dtype=bool)
        for e in range(n_ent):
            v = vals[slots[e]:slots[e + 1]]
            for r in range(nr[e]):
                for c in range(nc[e]):
                    x = v[c * nr[e] + r] if col_major else v[r * ld[e] + c]
                    if x != 0.0:
                        ref[ri[rp[e] + r], ci[cp[e] + c]] += x; touched[ri[rp[e] + r], ci[cp[e] + c]] = True
        assert M.has_sorted_indices and M.indices.dtype == np.int32
        assert np.array_equal(M.toarray(), ref)
        assert M.nnz == int(touched.sum())                                           # stored pattern = entries some block set (a sum may be 0)
        chk = M.copy(); chk.has_sorted_indices = False; chk.sort_indices()
        assert np.array_equal(chk.indices, M.indices)
    empty = amge._blocks_to_csr(np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0), np.zeros(0, np.int64),
                                (3, 2), col_major=True)
    assert empty.nnz == 0 and empty.shape == (3, 2)
    with pytest.raises(ValueError, match="shorter than its slots"):
        amge._blocks_to_csr(np.array([0, 2]), np.array([0, 1]), np.array([0, 2]), np.array([0, 1]), np.zeros(3), np.array([2]), (2, 2), col_major=True)
    with pytest.raises(_lib.ElagHipError, match="outside the matrix"):
        amge._blocks_to_csr(np.array([0, 1]), np.array([7]), np.array([0, 1]), np.array([0]), np.ones(1), np.array([1]), (2, 2), col_major=True)


def test_direct_csr_assembly_is_deterministic_across_threads():
    """Many overlapping blocks with non-integer values: entries that meet are added in block order whatever the thread schedule."""
    rng = np.random.default_rng(11)
    n_ent, nrow, ncol = 3000, 400, 300
    rp, ri = ragged(rng, n_ent, nrow, 4); cp, ci = ragged(rng, n_ent, ncol, 3)
    nr, nc = np.diff(rp), np.diff(cp)
    slots = ptr_from_counts(nr * nc)
    vals = rng.standard_normal(slots[-1])
    ref = {}
    for e in range(n_ent):
        v = vals[slots[e]:slots[e + 1]]
        for r in range(nr[e]):
            for c in range(nc[e]):
                k = (int(ri[rp[e] + r]), int(ci[cp[e] + c]))
                ref[k] = ref[k] + v[r * nc[e] + c] if k in ref else v[r * nc[e] + c]
    for _ in range(3):
        M = amge._blocks_to_csr(rp, ri, cp, ci, vals, nc, (nrow, ncol), col_major=False).tocoo()
        assert M.nnz == len(ref)
        assert all(ref[(int(r), int(c))] == v for r, c, v in zip(M.row, M.col, M.data))


def test_tables_do_not_depend_on_the_thread_count():
    """Sizes large enough that the table functions split their work: products (signed and pattern), minimal intersection sets, the
    coarse topology and the agglomerate tables with 1 thread == with 7 threads, entry for entry."""
    from parelag_amd import amge_level
    lib = _lib.load()
    rng = np.random.default_rng(5)
    a, b = ragged(rng, 3000, 800, 6, signed=True), ragged(rng, 800, 500, 5, signed=True)
    args = _agglomerate_table_inputs((16, 16, 16), (4, 4, 4), seed=2)
    lv = amge_level.level_from_hex((20, 20, 20))
    ijk = np.stack(np.unravel_index(np.arange(lv.nE), (20, 20, 20)), axis=1)
    part = (ijk[:, 0] // 4) * 25 + (ijk[:, 1] // 4) * 5 + ijk[:, 2] // 4

    def everything():
        out = list(_product(a[0], a[1], b[0], b[1], a[2], b[2])) + list(_product(a[0], a[1], b[0], b[1]))
        topo = amge.coarsen_topology(lv, part)
        out += [np.asarray(topo[k]) for k in sorted(topo)]
        tb = amge._agglomerate_tables(*args)
        return out + [tb[k] for k in sorted(tb)]
    try:
        assert lib.elag_hip_host_set_threads(1) == 1
        one = everything()
        assert lib.elag_hip_host_set_threads(7) == 7 and lib.elag_hip_host_set_min_work(64) == 64         # split even these small tables
        many = everything()
    finally:
        lib.elag_hip_host_set_threads(0); lib.elag_hip_host_set_min_work(0)
    assert len(one) == len(many)
    for x, y in zip(one, many):
        assert x.dtype == y.dtype and np.array_equal(x, y)
    rp, ri, rs = mult_orientation(a, b)                                   # and the row-loop statement of the product
    assert np.array_equal(one[0], rp) and np.array_equal(one[1], ri) and np.array_equal(one[2], rs)


def test_block_families_in_one_pass_equal_the_sum_of_the_families():
    rng = np.random.default_rng(8)
    nrow, ncol, sets, total = 60, 45, [], 0
    for n_ent in (25, 0, 40):
        rp, ri = ragged(rng, n_ent, nrow, 5); cp, ci = ragged(rng, n_ent, ncol, 4)
        ld = np.diff(cp) + rng.integers(0, 3, n_ent)
        vals = rng.integers(-2, 3, int((np.diff(rp) * ld).sum()) + 3).astype(np.float64)      # a longer value array is fine
        sets.append((rp, ri, cp, ci, vals, ld))
        total = total + amge._blocks_to_csr(rp, ri, cp, ci, vals, ld, (nrow, ncol), col_major=True).toarray()
    M = amge._blocks_to_csr_sets(sets, (nrow, ncol))
    assert M.has_sorted_indices and np.array_equal(M.toarray(), total)
    short = list(sets); short[0] = sets[0][:4] + (sets[0][4][:2], sets[0][5])
    with pytest.raises(ValueError, match="shorter than its slots"):
        amge._blocks_to_csr_sets(short, (nrow, ncol))


def test_local_triplets_match_the_whole_array_statement():
    rng = np.random.default_rng(21)
    n_ent, nrow, ncol = 300, 500, 400
    D = sp.random(nrow, ncol, density=0.02, random_state=4, format="csr")
    rp, rows = ragged(rng, n_ent, nrow, 6)
    # the columns of an entity: everything its rows touch, shuffled, plus a few more (one list is long enough for the hashed finder)
    cols = []
    for e in range(n_ent):
        c = np.unique(np.concatenate([D.indices[D.indptr[r]:D.indptr[r + 1]] for r in rows[rp[e]:rp[e + 1]]] + [rng.choice(ncol, 40 if e == 7 else 3)]))
        cols.append(rng.permutation(c))
    look = derham._Lookup(ptr_from_counts([len(c) for c in cols]), np.concatenate(cols), ncol)
    got, ref = derham._triplets(D, rp, rows, look), derham._triplets_numpy(D, rp, rows, look)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    look.idx[look.ptr[5]:look.ptr[6]] = (look.idx[look.ptr[5]:look.ptr[6]] + 1) % ncol                 # entity 5 loses its columns
    if rp[6] > rp[5] and np.diff(D.indptr)[rows[rp[5]:rp[6]]].sum():
        for fn in (derham._triplets, derham._triplets_numpy):
            with pytest.raises(ValueError, match="outside it"):
                fn(D, rp, rows, look)


def test_agglomerate_table_functions_on_empty_input():
    """no agglomerates, no blocks, no elements: sizes come back zero and nothing is touched"""
    import ctypes as C
    lib = _lib.load()
    z64 = np.zeros(1, dtype=np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    e64, ed = np.zeros(0, dtype=np.int64), np.zeros(0)
    out_ptr = np.full(1, -7, dtype=np.int64)
    assert lib.elag_hip_host_amge_dof_agglomeration(0, p(z64), p(e64), p(z64), p(e64), 0, p(e64), p(out_ptr), p(e64), p(e64), p(e64)) == 0 and out_ptr[0] == 0
    cb, rb = np.full(1, -7, dtype=np.int64), np.full(1, -7, dtype=np.int64)
    assert lib.elag_hip_host_amge_boundary_traces(0, p(z64), p(e64), p(z64), p(e64), p(e64), p(z64), 1, p(ed), p(z64), p(e64), p(e64), p(cb), p(rb), None, None) == 0
    assert cb[0] == 0 and rb[0] == 0
    i32 = np.zeros(1, dtype=np.int32)
    dp = np.full(1, -7, dtype=np.int64)
    assert lib.elag_hip_host_amge_local_derivative(0, p(z64), p(e64), p(i32), p(i32), p(ed), p(z64), p(e64), p(e64), p(dp), None, None, None) == 0 and dp[0] == 0
    assert lib.elag_hip_host_local_triplets(0, p(z64), p(e64), p(i32), p(i32), p(ed), p(z64), p(e64), p(dp), None, None, None) == 0 and dp[0] == 0
    tp = np.full(1, -7, dtype=np.int64)
    assert lib.elag_hip_host_amge_elem_dof_table(0, p(z64), p(e64), p(z64), p(z64), 0, p(tp), None) == 0 and tp[0] == 0
    indptr = np.full(4, -7, dtype=np.int64)
    assert lib.elag_hip_host_block_families_to_csr(0, None, 3, 2, p(indptr), None, None, None) == 0 and indptr.tolist() == [0, 0, 0, 0]
    M = amge._blocks_to_csr_sets([], (3, 2))
    assert M.shape == (3, 2) and M.nnz == 0


def test_entity_dof_table_equals_the_whole_array_statement():
    """elag_hip_host_entity_dof_table (round 5: the entity -> dof rows of DofHandler, layer by layer in connectivity order, then the entity's own interior
    dofs) against the numpy form it replaced -- ranges expanded, counted with bincount over repeat, rows concatenated -- on random tables: empty
    rows, empty layers, no layer at all, entities without dofs; and with one thread and several."""
    import numpy as np
    from parelag_amd import derham_level as dl, _lib

    def reference(nrow, layers, own_off, own_base):
        tabs = []
        for (p, i), ip, base in layers:
            d, own = dl._expand_ranges(base + np.asarray(ip)[i], np.diff(ip)[i])
            cnt = np.bincount(np.repeat(np.arange(len(p) - 1, dtype=np.int64), np.diff(p))[own], minlength=len(p) - 1)
            tabs.append((dl.ptr_from_counts(cnt), d))
        ip = np.asarray(own_off)
        d, _ = dl._expand_ranges(own_base + ip[:-1], np.diff(ip))
        tabs.append((ip.astype(np.int64), d))
        return dl._concat_rows(tabs, nrow)

    rng = np.random.default_rng(3)
    lib = _lib.load()
    for threads in (1, 6):
        lib.elag_hip_host_set_threads(threads)
        old = lib.elag_hip_host_set_min_work(64)
        try:
            for nrow in (0, 1, 7, 300, 20000):
                layers = []
                for _q in range(int(rng.integers(0, 4))):
                    n2 = int(rng.integers(1, 60))
                    p = np.concatenate([[0], np.cumsum(rng.integers(0, 4, nrow))]).astype(np.int64)
                    i = rng.integers(0, n2, int(p[-1])).astype(np.int64)
                    off = np.concatenate([[0], np.cumsum(rng.integers(0, 3, n2))]).astype(np.int64)
                    layers.append(((p, i), off, int(rng.integers(0, 1000))))
                own = np.concatenate([[0], np.cumsum(rng.integers(0, 3, nrow))]).astype(np.int64)
                got, ref = dl.entity_dof_table(nrow, layers, own, 17), reference(nrow, layers, own, 17)
                assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
        finally:
            lib.elag_hip_host_set_min_work(old)
            lib.elag_hip_host_set_threads(0)
    with pytest.raises(ValueError):
        dl.entity_dof_table(3, [((np.array([0, 1, 2, 3]), np.array([0, 5, 1])), np.array([0, 1, 2]), 0)], np.zeros(4, dtype=np.int64), 0)
