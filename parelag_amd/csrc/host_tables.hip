// Host-side integer tables of the AMGe coarsening (no device work): the ragged-table primitives every stage of
// parelag_amd/derham.py is made of, in native code because they are 99 % of the coarsening's wall time.  They follow the
// reference's own host loops:
//   elag_hip_host_table_product   mfem::Mult pattern product with first-touch column order, and its signed form MultOrientation
//                             (src/topology/TopologyTable.cpp:130-139; marker-array Gustavson rows)
//   elag_hip_host_lookup_find     the position of a dof inside its agglomerate's dof list (DofAgglomeration's dofMapper /
//                             mfem::SparseMatrix::GetSubMatrix column markers, src/amge/DOFAgglomeration.cpp:533-655)
//   elag_hip_host_gather_dense    GetSubMatrix of a CSR matrix for every entity at once, packed column-major
//   elag_hip_host_expand_ranges   concatenated index ranges with their owners
// No GPU is touched; the functions are callable before elag_hip_init.
#include "common.hpp"
#include "host_parallel.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <thread>
#include <vector>

using namespace elag;

extern "C" {

// out_ptr[nrow+1]; out_idx / out_sgn with room for sum_k |b row of a_idx[k]| entries.  a_sgn / b_sgn / out_sgn all null
// (pattern product) or all given (signed product: sums with |s| <= 1e-10 are dropped, the others reduced to their sign).
ELAG_HIP_API int elag_hip_host_table_product(int64_t nrow, const int64_t *a_ptr, const int64_t *a_idx, const double *a_sgn, int64_t nmid,
                                         const int64_t *b_ptr, const int64_t *b_idx, const double *b_sgn, int64_t ncol, int64_t *out_ptr,
                                         int64_t *out_idx, double *out_sgn, int64_t *out_nnz)
{
    return guarded([&] {
        ELAG_REQUIRE(nrow >= 0 && nmid >= 0 && ncol >= 0 && a_ptr && b_ptr && out_ptr && out_nnz, "table_product: null or negative argument");
        const bool sgn = a_sgn != nullptr;
        ELAG_REQUIRE(sgn == (b_sgn != nullptr) && sgn == (out_sgn != nullptr), "table_product: signs must be given for both tables and the output, or for none");
        const int64_t row_grain = grain_for(nrow, 4 * a_ptr[nrow]);
        int64_t n = 0;
        if (nrow / std::max<int64_t>(1, (int64_t)((double)row_grain * (double)host_min_work() / 131072.0)) <= 1 || host_threads() == 1) {
            // small table: one pass with a marker array over the columns (Gustavson's row loop)
            std::vector<int64_t> where((size_t)ncol, -1);          // position of a column in the current output row
            out_ptr[0] = 0;
            for (int64_t r = 0; r < nrow; r++) {
                const int64_t row0 = n;
                for (int64_t k = a_ptr[r]; k < a_ptr[r + 1]; k++) {
                    const int64_t m = a_idx[k];
                    ELAG_REQUIRE(m >= 0 && m < nmid, "table_product: entry %lld of the left table is outside the right table", (long long)m);
                    for (int64_t q = b_ptr[m]; q < b_ptr[m + 1]; q++) {
                        const int64_t c = b_idx[q];
                        ELAG_REQUIRE(c >= 0 && c < ncol, "table_product: column %lld outside 0..%lld", (long long)c, (long long)ncol);
                        if (where[(size_t)c] < row0) {              // first touch in this row
                            where[(size_t)c] = n;
                            out_idx[n] = c;
                            if (sgn) out_sgn[n] = 0.0;
                            n++;
                        }
                        if (sgn) out_sgn[where[(size_t)c]] += a_sgn[k] * b_sgn[q];
                    }
                }
                if (sgn) {                                          // cancelled entries leave the row, the others keep their order
                    int64_t w = row0;
                    for (int64_t t = row0; t < n; t++) {
                        const int64_t c = out_idx[t];
                        const double v = out_sgn[t];
                        where[(size_t)c] = -1;                      // positions move: no marker of this row survives it
                        if (std::fabs(v) > 1e-10) {
                            out_idx[w] = c;
                            out_sgn[w] = v > 0.0 ? 1.0 : -1.0;
                            w++;
                        }
                    }
                    n = w;
                }
                out_ptr[r + 1] = n;
            }
        } else {
            // two passes over the rows, both on threads: count the columns every row keeps, then write them at their final place.  A thread
            // holds the columns of its current row in a small open-addressing table (insertion order = first touch).
            auto rows = [&](int64_t rb, int64_t re, bool write) {
                std::vector<int64_t> hk, hp, cols;
                std::vector<double> acc;
                std::vector<size_t> used;
                size_t cap = 0; int shift = 0;
                for (int64_t r = rb; r < re; r++) {
                    int64_t cand = 0;
                    for (int64_t k = a_ptr[r]; k < a_ptr[r + 1]; k++) {
                        const int64_t m = a_idx[k];
                        if (m < 0 || m >= nmid) throw Oops("table_product: an entry of the left table is outside the right table");
                        cand += b_ptr[m + 1] - b_ptr[m];
                    }
                    if ((size_t)(2 * cand + 2) > cap) {
                        cap = 64; shift = 58;
                        while (cap < (size_t)(2 * cand + 2)) { cap <<= 1; shift--; }
                        hk.assign(cap, -1); hp.assign(cap, 0);
                    }
                    cols.clear(); acc.clear(); used.clear();
                    for (int64_t k = a_ptr[r]; k < a_ptr[r + 1]; k++) {
                        const int64_t m = a_idx[k];
                        for (int64_t q = b_ptr[m]; q < b_ptr[m + 1]; q++) {
                            const int64_t c = b_idx[q];
                            if (c < 0 || c >= ncol) throw Oops("table_product: a column is outside the column range");
                            size_t p = (size_t)(((uint64_t)c * 0x9E3779B97F4A7C15ull) >> shift);
                            while (hk[p] != -1 && hk[p] != c) p = (p + 1) & (cap - 1);
                            if (hk[p] == -1) {                      // first touch in this row
                                hk[p] = c; hp[p] = (int64_t)cols.size(); used.push_back(p);
                                cols.push_back(c);
                                if (sgn) acc.push_back(0.0);
                            }
                            if (sgn) acc[(size_t)hp[p]] += a_sgn[k] * b_sgn[q];
                        }
                    }
                    for (size_t p : used) hk[p] = -1;
                    int64_t w = write ? out_ptr[r] : 0;
                    for (size_t x = 0; x < cols.size(); x++) {
                        if (sgn && !(std::fabs(acc[x]) > 1e-10)) continue;       // cancelled entries leave the row, the others keep their order
                        if (write) { out_idx[w] = cols[x]; if (sgn) out_sgn[w] = acc[x] > 0.0 ? 1.0 : -1.0; }
                        w++;
                    }
                    if (!write) out_ptr[r + 1] = w;
                }
            };
            out_ptr[0] = 0;
            parallel_parts(nrow, [&](int, int64_t rb, int64_t re) { rows(rb, re, false); }, row_grain);
            for (int64_t r = 0; r < nrow; r++) out_ptr[r + 1] += out_ptr[r];
            n = out_ptr[nrow];
            parallel_parts(nrow, [&](int, int64_t rb, int64_t re) { rows(rb, re, true); }, row_grain);
        }
        *out_nnz = n;
    });
}

// pos[q] = position of q_id[q] inside row q_row[q] of the ragged table (ptr, idx), -1 when absent.
ELAG_HIP_API int elag_hip_host_lookup_find(int64_t nrow, const int64_t *ptr, const int64_t *idx, int64_t nid, int64_t nq, const int64_t *q_row,
                                       const int64_t *q_id, int64_t *pos)
{
    return guarded([&] {
        ELAG_REQUIRE(nrow >= 0 && nid >= 0 && nq >= 0 && ptr && (nq == 0 || (q_row && q_id && pos)), "lookup_find: null or negative argument");
        if (nq == 0) return;
        // queries are independent: a thread keeps the row of its last query at hand (queries usually come grouped by row)
        parallel_parts(nq, [&](int, int64_t qb, int64_t qe) {
            RowFinder f;
            for (int64_t q = qb; q < qe; q++) {
                const int64_t r = q_row[q];
                if (r < 0 || r >= nrow) throw Oops("lookup_find: a query row is outside the table");
                f.set(idx + ptr[r], ptr[r + 1] - ptr[r]);
                pos[q] = f.find(q_id[q]);
            }
        }, 16384);
    });
}

// For every entity e: the dense block M[rows of e, columns of e], packed column-major (nr_e x ncol_e) at out + out_ptr[e];
// rows of e = rows[row_ptr[e] .. row_ptr[e+1]), columns of e = col_idx[col_ptr[e] .. col_ptr[e+1]) (entries of M outside them
// are skipped, as mfem's GetSubMatrix does with unmarked columns).  `out` must be zero on entry.
ELAG_HIP_API int elag_hip_host_gather_dense(int64_t n_ent, const int64_t *row_ptr, const int64_t *rows, const int64_t *col_ptr, const int64_t *col_idx,
                                        const int64_t *ncol, int64_t m_rows, int64_t m_cols, const int32_t *m_indptr, const int32_t *m_indices,
                                        const double *m_data, const int64_t *out_ptr, double *out)
{
    return guarded([&] {
        ELAG_REQUIRE(n_ent >= 0 && row_ptr && col_ptr && ncol && m_indptr && out_ptr, "gather_dense: null or negative argument");
        parallel_parts(n_ent, [&](int, int64_t eb, int64_t ee) {
            RowFinder f;
            for (int64_t e = eb; e < ee; e++) {
                const int64_t nr = row_ptr[e + 1] - row_ptr[e];
                if (col_ptr[e + 1] - col_ptr[e] > ncol[e] || out_ptr[e + 1] - out_ptr[e] != nr * ncol[e]) throw Oops("gather_dense: a block has the wrong size");
                for (int64_t k = col_ptr[e]; k < col_ptr[e + 1]; k++)
                    if (col_idx[k] < 0 || col_idx[k] >= m_cols) throw Oops("gather_dense: a column is outside the matrix");
                f.set(col_idx + col_ptr[e], col_ptr[e + 1] - col_ptr[e]);
                double *blk = out + out_ptr[e];
                for (int64_t lr = 0; lr < nr; lr++) {
                    const int64_t g = rows[row_ptr[e] + lr];
                    if (g < 0 || g >= m_rows) throw Oops("gather_dense: a row is outside the matrix");
                    for (int32_t k = m_indptr[g]; k < m_indptr[g + 1]; k++) {
                        const int64_t at = f.find(m_indices[k]);
                        if (at >= 0) blk[at * nr + lr] = m_data[k];
                    }
                }
            }
        }, grain_for(n_ent, 8 * row_ptr[n_ent]));
    });
}

// idx = concatenation of [starts[i], starts[i] + counts[i]); owner[k] = i.  `total` must equal the sum of the counts.
ELAG_HIP_API int elag_hip_host_expand_ranges(int64_t n, const int64_t *starts, const int64_t *counts, int64_t total, int64_t *idx, int64_t *owner)
{
    return guarded([&] {
        ELAG_REQUIRE(n >= 0 && total >= 0 && (n == 0 || (starts && counts)) && (total == 0 || (idx && owner)), "expand_ranges: null or negative argument");
        std::vector<int64_t> off((size_t)n + 1, 0);
        for (int64_t i = 0; i < n; i++) {
            ELAG_REQUIRE(counts[i] >= 0 && off[(size_t)i] + counts[i] <= total, "expand_ranges: the counts do not add up to the given total");
            off[(size_t)i + 1] = off[(size_t)i] + counts[i];
        }
        ELAG_REQUIRE(off[(size_t)n] == total, "expand_ranges: the counts do not add up to the given total");
        parallel_parts(n, [&](int, int64_t b, int64_t e) {
            for (int64_t i = b; i < e; i++) {
                int64_t k = off[(size_t)i];
                for (int64_t j = 0; j < counts[i]; j++, k++) { idx[k] = starts[i] + j; owner[k] = i; }
            }
        }, grain_for(n, total, 262144));
    });
}

// entity_dof of one form and one codimension: row r = for every layer q (the codimensions 3 ... c + 1 that carry dofs of the form, in that order) the
// interior dofs of the entities conn_q(r, :) in connectivity order -- base[q] + dof_off[q][e] ... dof_off[q][e + 1] -- then the entity's own interior dofs
// own_base + own_off[r] ... own_off[r + 1].  out_idx == NULL: only the row pointers (out_ptr[nrow]: the size to allocate).
ELAG_HIP_API int elag_hip_host_entity_dof_table(int64_t nrow, int nlayers, const int64_t *const *conn_ptr, const int64_t *const *conn_idx,
                                                const int64_t *const *dof_off, const int64_t *base, const int64_t *own_off, int64_t own_base,
                                                int64_t *out_ptr, int64_t *out_idx)
{
    return guarded([&] {
        ELAG_REQUIRE(nrow >= 0 && nlayers >= 0 && out_ptr && own_off && (nlayers == 0 || (conn_ptr && conn_idx && dof_off && base)), "entity_dof_table: null or negative argument");
        for (int q = 0; q < nlayers; q++) ELAG_REQUIRE(conn_ptr[q] && dof_off[q] && (conn_ptr[q][nrow] == 0 || conn_idx[q]), "entity_dof_table: null layer %d", q);
        int64_t work = own_off[nrow] - own_off[0];
        for (int q = 0; q < nlayers; q++) work += 2 * conn_ptr[q][nrow];
        const int64_t grain = grain_for(nrow, 2 * work + nrow);
        if (!out_idx) {
            parallel_ranges(nrow, [&](int64_t b, int64_t e) {
                for (int64_t r = b; r < e; r++) {
                    int64_t cnt = own_off[r + 1] - own_off[r];
                    for (int q = 0; q < nlayers; q++)
                        for (int64_t k = conn_ptr[q][r]; k < conn_ptr[q][r + 1]; k++) { const int64_t en = conn_idx[q][k]; cnt += dof_off[q][en + 1] - dof_off[q][en]; }
                    out_ptr[r + 1] = cnt;
                }
            }, grain);
            out_ptr[0] = 0;
            for (int64_t r = 0; r < nrow; r++) out_ptr[r + 1] += out_ptr[r];
            return;
        }
        parallel_ranges(nrow, [&](int64_t b, int64_t e) {
            for (int64_t r = b; r < e; r++) {
                int64_t w = out_ptr[r];
                for (int q = 0; q < nlayers; q++)
                    for (int64_t k = conn_ptr[q][r]; k < conn_ptr[q][r + 1]; k++) {
                        const int64_t en = conn_idx[q][k];
                        for (int64_t d = dof_off[q][en]; d < dof_off[q][en + 1]; d++) out_idx[w++] = base[q] + d;
                    }
                for (int64_t d = own_off[r]; d < own_off[r + 1]; d++) out_idx[w++] = own_base + d;
                if (w != out_ptr[r + 1]) throw Error(ELAG_HIP_ERR_ARG, "entity_dof_table: the row pointers do not belong to these tables");
            }
        }, grain);
    });
}

// Triplets of a sparse matrix whose entity blocks are dense (mfem's SetSubMatrix / AddSubMatrix of every entity's local matrix,
// exact zeros dropped): block e couples rows_idx[rows_ptr[e]..] with cols_idx[cols_ptr[e]..]; its values sit in consecutive slots
// of (slot_rows ? slot_rows[e] : nrows_e) * ld[e] doubles -- column-major (column c at c * nrows_e) or row-major (row r at r * ld[e]).
// out_* need room for sum_e nrows_e * ncols_e entries.
ELAG_HIP_API int elag_hip_host_blocks_to_coo(int64_t n_ent, const int64_t *rows_ptr, const int64_t *rows_idx, const int64_t *cols_ptr,
                                             const int64_t *cols_idx, const double *vals, const int64_t *ld, const int64_t *slot_rows, int col_major,
                                             int64_t *out_r, int64_t *out_c, double *out_v, int64_t *out_nnz)
{
    return guarded([&] {
        ELAG_REQUIRE(n_ent >= 0 && rows_ptr && cols_ptr && ld && out_nnz, "blocks_to_coo: null or negative argument");
        int64_t slot = 0, n = 0;
        for (int64_t e = 0; e < n_ent; e++) {
            const int64_t nr = rows_ptr[e + 1] - rows_ptr[e], nc = cols_ptr[e + 1] - cols_ptr[e];
            ELAG_REQUIRE(nr >= 0 && nc >= 0 && ld[e] >= nc, "blocks_to_coo: block %lld does not fit its slot", (long long)e);
            const double *v = vals + slot;
            const int64_t *R = rows_idx + rows_ptr[e], *Cc = cols_idx + cols_ptr[e];
            if (col_major) {
                for (int64_t c = 0; c < nc; c++)
                    for (int64_t r = 0; r < nr; r++) {
                        const double x = v[c * nr + r];
                        if (x != 0.0) { out_r[n] = R[r]; out_c[n] = Cc[c]; out_v[n] = x; n++; }
                    }
            } else {
                for (int64_t r = 0; r < nr; r++)
                    for (int64_t c = 0; c < nc; c++) {
                        const double x = v[r * ld[e] + c];
                        if (x != 0.0) { out_r[n] = R[r]; out_c[n] = Cc[c]; out_v[n] = x; n++; }
                    }
            }
            slot += (slot_rows ? slot_rows[e] : nr) * ld[e];
        }
        *out_nnz = n;
    });
}

// Per entity the rows dofs[ptr[e] .. ptr[e+1]) of the row-major (ndof x nT) array T, stored column-major, entities one after
// the other (the local target matrices of ComputeCoarseTraces / the extensions).
ELAG_HIP_API int elag_hip_host_pack_cols(int64_t n_ent, const int64_t *ptr, const int64_t *dofs, int64_t ndof, int64_t nT, const double *T, double *out)
{
    return guarded([&] {
        ELAG_REQUIRE(n_ent >= 0 && ptr && nT >= 0 && ndof >= 0, "pack_cols: null or negative argument");
        parallel_parts(n_ent, [&](int, int64_t eb, int64_t ee) {
            for (int64_t e = eb; e < ee; e++) {
                const int64_t n = ptr[e + 1] - ptr[e];
                double *blk = out + ptr[e] * nT;
                for (int64_t i = 0; i < n; i++) {
                    const int64_t d = dofs[ptr[e] + i];
                    if (d < 0 || d >= ndof) throw Oops("pack_cols: a dof is outside the vector");
                    for (int64_t c = 0; c < nT; c++) blk[c * n + i] = T[d * nT + c];
                }
            }
        }, grain_for(n_ent, ptr[n_ent] * std::max<int64_t>(nT, 1)));
    });
}

// From consecutive row-major ld[e] x ld[e] slots keep the leading k[e] x k[e] blocks, packed.
ELAG_HIP_API int elag_hip_host_take_leading(int64_t n_ent, const int64_t *ld, const int64_t *k, const double *vals, double *out)
{
    return guarded([&] {
        ELAG_REQUIRE(n_ent >= 0 && (n_ent == 0 || (ld && k)), "take_leading: null or negative argument");
        int64_t slot = 0, w = 0;
        for (int64_t e = 0; e < n_ent; e++) {
            ELAG_REQUIRE(k[e] >= 0 && k[e] <= ld[e], "take_leading: block %lld larger than its slot", (long long)e);
            for (int64_t r = 0; r < k[e]; r++)
                for (int64_t c = 0; c < k[e]; c++) out[w++] = vals[slot + r * ld[e] + c];
            slot += ld[e] * ld[e];
        }
    });
}

// Every row of the ragged table sorted by (key[id], id): DofAgglomeration's reorderRow, which puts the dofs interior to an
// agglomerated entity first and those shared with entities of higher codimension behind them, each group ascending
// (src/amge/DOFAgglomeration.cpp:33-131).  In place.
ELAG_HIP_API int elag_hip_host_sort_rows_by_key(int64_t nrow, const int64_t *ptr, int64_t *idx, int64_t nid, const int64_t *key)
{
    return guarded([&] {
        ELAG_REQUIRE(nrow >= 0 && ptr && (nid == 0 || key), "sort_rows_by_key: null or negative argument");
        parallel_parts(nrow, [&](int, int64_t rb, int64_t re) {
            for (int64_t r = rb; r < re; r++) {
                for (int64_t k = ptr[r]; k < ptr[r + 1]; k++)
                    if (idx[k] < 0 || idx[k] >= nid) throw Oops("sort_rows_by_key: an id is outside the key range");
                std::sort(idx + ptr[r], idx + ptr[r + 1], [key](int64_t a, int64_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
            }
        }, grain_for(nrow, 8 * ptr[nrow]));
    });
}

// findMinimalIntersectionSets (src/structures/minimalIntersectionSet.cpp:43-132) on Z = AE_fc^T AE_fc (+ the boundary-attribute
// term for facets, Topology.cpp:757-765) without forming Z: two sub-entities on agglomerate boundaries belong to the same set iff
// they have the same adjacency signature -- the same agglomerates with the same signs relative to the first one (absolute signs
// and the same attribute for boundary facets with an attribute, whose term in Z cannot change sign).  Sets are numbered by their
// lowest member and oriented like it.  (fp, fi, fs): AE_fc, rows ascending; mis[e] = set of sub-entity e or -1; ori[e] = +-1 or 0.
ELAG_HIP_API int elag_hip_host_minimal_intersection_sets(int64_t nrow, const int64_t *fp, const int64_t *fi, const double *fs, int64_t nsub,
                                                         const int32_t *attr, int64_t *mis, double *ori, int64_t *n_sets)
{
    return guarded([&] {
        ELAG_REQUIRE(nrow >= 0 && nsub >= 0 && fp && n_sets && (nsub == 0 || (mis && ori)), "minimal_intersection_sets: null or negative argument");
        const int64_t nnz = fp[nrow];
        // transposed table (sub-entity -> rows, ascending): every thread owns a range of sub-entities and scans the whole table for them
        std::vector<int64_t> tp((size_t)nsub + 1, 0);
        parallel_parts(nsub, [&](int, int64_t sb, int64_t se) {
            for (int64_t q = 0; q < nnz; q++) {
                const int64_t e = fi[q];
                if (e < 0 || e >= nsub) throw Oops("minimal_intersection_sets: a sub-entity is outside the sub-entity range");
                if (e >= sb && e < se) tp[(size_t)e + 1]++;
            }
        }, 4096);
        for (int64_t e = 0; e < nsub; e++) tp[(size_t)e + 1] += tp[(size_t)e];
        std::vector<int64_t> ta((size_t)nnz), fill(tp.begin(), tp.end() - 1);
        std::vector<double> ts((size_t)nnz);
        parallel_parts(nsub, [&](int, int64_t sb, int64_t se) {
            for (int64_t a = 0; a < nrow; a++)                   // rows ascending: every transposed row comes out ascending
                for (int64_t q = fp[a]; q < fp[a + 1]; q++) {
                    const int64_t e = fi[q];
                    if (e < sb || e >= se) continue;
                    const int64_t w = fill[(size_t)e]++;
                    ta[(size_t)w] = a;
                    ts[(size_t)w] = fs[q];
                }
        }, 4096);
        auto same = [&](int64_t e, int64_t r) {
            const int64_t n = tp[(size_t)e + 1] - tp[(size_t)e];
            if (n != tp[(size_t)r + 1] - tp[(size_t)r]) return false;
            const int ae = attr ? attr[e] : 0, ar = attr ? attr[r] : 0;
            if (ae != ar) return false;
            const double se = ae > 0 ? 1.0 : ts[(size_t)tp[(size_t)e]], sr = ar > 0 ? 1.0 : ts[(size_t)tp[(size_t)r]];
            for (int64_t k = 0; k < n; k++)
                if (ta[(size_t)(tp[(size_t)e] + k)] != ta[(size_t)(tp[(size_t)r] + k)] ||
                    ts[(size_t)(tp[(size_t)e] + k)] * se != ts[(size_t)(tp[(size_t)r] + k)] * sr)
                    return false;
            return true;
        };
        // hash of every signature; the hash's top bits pick the thread that groups the sub-entity
        const int nparts = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, nsub / 4096));
        std::vector<uint64_t> hash((size_t)nsub);
        std::vector<int64_t> part_count((size_t)nparts * (size_t)host_threads(), 0);
        auto part_of = [&](uint64_t h) { return (int)(((h >> 33) * (uint64_t)nparts) >> 31); };
        parallel_parts(nsub, [&](int t, int64_t sb, int64_t se) {
            for (int64_t e = sb; e < se; e++) {
                const int64_t n = tp[(size_t)e + 1] - tp[(size_t)e];
                if (n == 0) { hash[(size_t)e] = 0; continue; }
                const int ae = attr ? attr[e] : 0;
                const double s0 = ae > 0 ? 1.0 : ts[(size_t)tp[(size_t)e]];
                uint64_t h = 1469598103934665603ull ^ (uint64_t)ae;
                for (int64_t k = 0; k < n; k++) {
                    const uint64_t v = (uint64_t)ta[(size_t)(tp[(size_t)e] + k)] * 2u + (ts[(size_t)(tp[(size_t)e] + k)] * s0 > 0 ? 1u : 0u);
                    h = (h ^ v) * 1099511628211ull;
                    h ^= h >> 29;
                }
                hash[(size_t)e] = h;
                part_count[(size_t)t * (size_t)nparts + (size_t)part_of(h)]++;
            }
        }, 4096);
        // representative (lowest member) of every sub-entity's set: each thread scans the sub-entities in ascending order and groups its share
        std::vector<int64_t> rep_of((size_t)nsub, -1);
        std::vector<std::thread> th;
        std::vector<std::string> err((size_t)nparts);
        for (int p = 0; p < nparts; p++) {
            auto job = [&, p] {
                try {
                    int64_t mine = 0;
                    for (int t = 0; t < host_threads(); t++) mine += part_count[(size_t)t * (size_t)nparts + (size_t)p];
                    size_t cap = 16;
                    while (cap < (size_t)mine * 2 + 16) cap <<= 1;
                    std::vector<int64_t> slot(cap, -1);
                    for (int64_t e = 0; e < nsub; e++) {
                        if (tp[(size_t)e + 1] == tp[(size_t)e]) continue;
                        const uint64_t h = hash[(size_t)e];
                        if (part_of(h) != p) continue;
                        size_t q = (size_t)h & (cap - 1);
                        int64_t g = -1;
                        while (slot[q] >= 0) {
                            if (same(e, slot[q])) { g = slot[q]; break; }
                            q = (q + 1) & (cap - 1);
                        }
                        if (g < 0) { g = e; slot[q] = e; }                                // e ascends: the representative is the lowest member
                        rep_of[(size_t)e] = g;
                    }
                } catch (const std::exception &ex) { err[(size_t)p] = ex.what(); }
            };
            if (nparts == 1) job(); else th.emplace_back(job);
        }
        for (auto &x : th) x.join();
        for (auto &m : err)
            if (!m.empty()) fail(ELAG_HIP_ERR_ARG, "%s", m.c_str());
        // sets are numbered by their lowest member
        std::vector<int64_t> set_of_rep((size_t)nsub, -1);
        int64_t nset = 0;
        for (int64_t e = 0; e < nsub; e++)
            if (rep_of[(size_t)e] == e) set_of_rep[(size_t)e] = nset++;
        parallel_ranges(nsub, [&](int64_t sb, int64_t se) {
            for (int64_t e = sb; e < se; e++) {
                const int64_t r = rep_of[(size_t)e];
                if (r < 0) { mis[e] = -1; ori[e] = 0.0; continue; }
                const int ae = attr ? attr[e] : 0;
                const double s0 = ae > 0 ? 1.0 : ts[(size_t)tp[(size_t)e]];
                mis[e] = set_of_rep[(size_t)r];
                ori[e] = ae > 0 ? 1.0 : s0 * ts[(size_t)tp[(size_t)r]];
            }
        });
        *n_sets = nset;
    });
}

}  // extern "C"
