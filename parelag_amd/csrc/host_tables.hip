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
        std::vector<int64_t> where((size_t)ncol, -1);          // position of a column in the current output row
        int64_t n = 0;
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
        bool sorted = true;
        for (int64_t q = 1; q < nq && sorted; q++) sorted = q_row[q - 1] <= q_row[q];
        std::vector<int64_t> order;
        if (!sorted) {                                          // counting sort of the queries by row
            std::vector<int64_t> start((size_t)nrow + 1, 0);
            for (int64_t q = 0; q < nq; q++) {
                ELAG_REQUIRE(q_row[q] >= 0 && q_row[q] < nrow, "lookup_find: row %lld outside the table", (long long)q_row[q]);
                start[(size_t)q_row[q] + 1]++;
            }
            for (int64_t r = 0; r < nrow; r++) start[(size_t)r + 1] += start[(size_t)r];
            order.resize((size_t)nq);
            for (int64_t q = 0; q < nq; q++) order[(size_t)start[(size_t)q_row[q]]++] = q;
        }
        std::vector<int64_t> mark((size_t)nid + 1, -1), stamp((size_t)nid + 1, -1);
        int64_t cur = -1;
        for (int64_t t = 0; t < nq; t++) {
            const int64_t q = sorted ? t : order[(size_t)t];
            const int64_t r = q_row[q];
            ELAG_REQUIRE(r >= 0 && r < nrow, "lookup_find: row %lld outside the table", (long long)r);
            if (r != cur) {
                cur = r;
                for (int64_t k = ptr[r]; k < ptr[r + 1]; k++) {
                    const int64_t i = idx[k];
                    ELAG_REQUIRE(i >= 0 && i <= nid, "lookup_find: id %lld outside 0..%lld", (long long)i, (long long)nid);
                    if (stamp[(size_t)i] != r) { stamp[(size_t)i] = r; mark[(size_t)i] = k - ptr[r]; }   // first occurrence wins
                }
            }
            const int64_t i = q_id[q];
            pos[q] = (i >= 0 && i <= nid && stamp[(size_t)i] == r) ? mark[(size_t)i] : -1;
        }
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
        std::vector<int64_t> mark((size_t)m_cols, -1), stamp((size_t)m_cols, -1);
        for (int64_t e = 0; e < n_ent; e++) {
            const int64_t nr = row_ptr[e + 1] - row_ptr[e];
            ELAG_REQUIRE(col_ptr[e + 1] - col_ptr[e] <= ncol[e] && out_ptr[e + 1] - out_ptr[e] == nr * ncol[e], "gather_dense: block %lld has the wrong size", (long long)e);
            for (int64_t k = col_ptr[e]; k < col_ptr[e + 1]; k++) {
                const int64_t c = col_idx[k];
                ELAG_REQUIRE(c >= 0 && c < m_cols, "gather_dense: column %lld outside the matrix", (long long)c);
                if (stamp[(size_t)c] != e) { stamp[(size_t)c] = e; mark[(size_t)c] = k - col_ptr[e]; }
            }
            double *blk = out + out_ptr[e];
            for (int64_t lr = 0; lr < nr; lr++) {
                const int64_t g = rows[row_ptr[e] + lr];
                ELAG_REQUIRE(g >= 0 && g < m_rows, "gather_dense: row %lld outside the matrix", (long long)g);
                for (int32_t k = m_indptr[g]; k < m_indptr[g + 1]; k++) {
                    const int32_t c = m_indices[k];
                    if (stamp[(size_t)c] == e) blk[mark[(size_t)c] * nr + lr] = m_data[k];
                }
            }
        }
    });
}

// idx = concatenation of [starts[i], starts[i] + counts[i]); owner[k] = i.  `total` must equal the sum of the counts.
ELAG_HIP_API int elag_hip_host_expand_ranges(int64_t n, const int64_t *starts, const int64_t *counts, int64_t total, int64_t *idx, int64_t *owner)
{
    return guarded([&] {
        ELAG_REQUIRE(n >= 0 && total >= 0 && (n == 0 || (starts && counts)) && (total == 0 || (idx && owner)), "expand_ranges: null or negative argument");
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++) {
            ELAG_REQUIRE(counts[i] >= 0 && k + counts[i] <= total, "expand_ranges: the counts do not add up to the given total");
            for (int64_t j = 0; j < counts[i]; j++, k++) { idx[k] = starts[i] + j; owner[k] = i; }
        }
        ELAG_REQUIRE(k == total, "expand_ranges: the counts do not add up to the given total");
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
        for (int64_t e = 0; e < n_ent; e++) {
            const int64_t n = ptr[e + 1] - ptr[e];
            double *blk = out + ptr[e] * nT;
            for (int64_t i = 0; i < n; i++) {
                const int64_t d = dofs[ptr[e] + i];
                ELAG_REQUIRE(d >= 0 && d < ndof, "pack_cols: dof %lld outside the vector", (long long)d);
                for (int64_t c = 0; c < nT; c++) blk[c * n + i] = T[d * nT + c];
            }
        }
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
        for (int64_t r = 0; r < nrow; r++) {
            for (int64_t k = ptr[r]; k < ptr[r + 1]; k++)
                ELAG_REQUIRE(idx[k] >= 0 && idx[k] < nid, "sort_rows_by_key: id %lld outside 0..%lld", (long long)idx[k], (long long)nid);
            std::sort(idx + ptr[r], idx + ptr[r + 1], [key](int64_t a, int64_t b) { return key[a] != key[b] ? key[a] < key[b] : a < b; });
        }
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
