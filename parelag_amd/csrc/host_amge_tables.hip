// Host-side tables of the two-form AMGe coarsening (parelag_amd/amge.py), one agglomerate at a time like the reference's own loops
// (DofAgglomeration, src/amge/DOFAgglomeration.cpp:86-131, 533-655; hFacetExtension's local index work,
// src/amge/DeRhamSequence.cpp:2214-2400), spread over host threads: an agglomerate only reads shared tables and writes its own
// slices, so the loops split by agglomerate range with nothing to lock.  They replace whole-array numpy compositions
// (unique / argsort / searchsorted joins on (agglomerate, dof) keys) and produce the same arrays bit for bit.
// No device work; callable before elag_hip_init.
#include "common.hpp"
#include "host_parallel.hpp"

#include <malloc.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

using namespace elag;


extern "C" {

// Number of host threads the table functions may use (n <= 0: back to the default / ELAG_HIP_HOST_THREADS); returns the count in force.
ELAG_HIP_API int elag_hip_host_set_threads(int n)
{
    host_threads_setting() = n > 0 ? std::min(n, 256) : 0;
    return host_threads();
}

// Work (about a nanosecond per unit) a thread must get before a table loop is split; min_work <= 0: the default, 131072.  Returns the value in force.
ELAG_HIP_API int64_t elag_hip_host_set_min_work(int64_t min_work)
{
    host_min_work() = min_work > 0 ? min_work : 131072;
    return host_min_work();
}

// Allocator policy for the table stage: its arrays are tens to hundreds of MB, allocated and freed once per stage; glibc hands such blocks
// back to the kernel on free, so every stage pays the page faults again (and faults taken by many threads at once serialise on the
// process's address-space lock: 0.76 s -> 0.46 s for the finest coarsening of the SPE10-size grid).  on != 0: large blocks come from the
// heap and freed pages stay with the process (M_MMAP_MAX = 0, M_TRIM_THRESHOLD = 2 GiB - 1); on == 0: glibc's defaults again and the
// kept pages returned (malloc_trim).  The Python callers switch it on for the duration of one coarsening only (_lib.host_pages_kept).
ELAG_HIP_API int elag_hip_host_keep_pages(int on)
{
    return guarded([&] {
        const int ok = on ? (mallopt(M_MMAP_MAX, 0) && mallopt(M_TRIM_THRESHOLD, 0x7fffffff))
                          : (mallopt(M_MMAP_MAX, 65536) && mallopt(M_TRIM_THRESHOLD, 128 * 1024));
        ELAG_REQUIRE(ok, "elag_hip_host_keep_pages: mallopt refused the setting");
        if (!on) {
            // the pages kept meanwhile go back to the system -- when there are enough of them to matter: malloc_trim walks every heap and
            // costs ~50 ms after a cube456 x3 coarsening whose whole table stage takes 0.2 s; below 512 MiB at the top of the heap
            // (ELAG_HIP_HOST_TRIM_FROM_MB) glibc's own trimming, back in force with the thresholds above, takes care of it at the next large free
            static long long from_mb = -1;
            if (from_mb < 0) { const char *e = getenv("ELAG_HIP_HOST_TRIM_FROM_MB"); from_mb = e ? atoll(e) : 512; }
            const struct mallinfo2 mi = mallinfo2();
            if ((long long)(mi.keepcost >> 20) >= from_mb) (void)malloc_trim(0);
        }
    });
}

// DofAgglomeration of one form on the agglomerated elements.  Agglomerate a holds the elements els[ae_ptr[a] .. ae_ptr[a+1]); an
// element e carries the dofs e_idx[e_ptr[e] .. e_ptr[e+1]) (its "repeated" dofs, in this order).  sep[d] != 0 marks the dofs that lie on
// agglomerated facets.  Output: the dofs of every agglomerate without repetition, interior ones (sep == 0) first, each group
// ascending (reorderRow) -- ae_dof_ptr (nAE+1), ae_dofs (room for the number of repeated dofs), n_int (nAE) -- and for every repeated dof,
// in (agglomerate, element, local) order, its position in its agglomerate's list (loc, same length as the repeated dofs).
ELAG_HIP_API int elag_hip_host_amge_dof_agglomeration(int64_t nAE, const int64_t *ae_ptr, const int64_t *els, const int64_t *e_ptr, const int64_t *e_idx,
                                                      int64_t ndof, const int64_t *sep, int64_t *ae_dof_ptr, int64_t *n_int, int64_t *ae_dofs, int64_t *loc)
{
    return guarded([&] {
        ELAG_REQUIRE(nAE >= 0 && ae_ptr && els && e_ptr && e_idx && sep && ae_dof_ptr && n_int && ae_dofs && loc, "dof_agglomeration: null argument");
        // repeated-dof offset of every agglomerate
        std::vector<int64_t> rd_ptr((size_t)nAE + 1, 0);
        for (int64_t a = 0; a < nAE; a++) {
            int64_t s = 0;
            for (int64_t q = ae_ptr[a]; q < ae_ptr[a + 1]; q++) s += e_ptr[els[q] + 1] - e_ptr[els[q]];
            rd_ptr[(size_t)a + 1] = rd_ptr[(size_t)a] + s;
        }
        std::vector<int64_t> tmp((size_t)rd_ptr[(size_t)nAE]);          // sorted distinct keys of agglomerate a at tmp[rd_ptr[a] ..)
        std::vector<int64_t> nu((size_t)nAE, 0);
        const int64_t SEP = int64_t(1) << 62;
        parallel_ranges(nAE, [&](int64_t b, int64_t e) {
            std::vector<int64_t> keys;
            for (int64_t a = b; a < e; a++) {
                keys.clear();
                for (int64_t q = ae_ptr[a]; q < ae_ptr[a + 1]; q++) {
                    const int64_t el = els[q];
                    for (int64_t t = e_ptr[el]; t < e_ptr[el + 1]; t++) {
                        const int64_t d = e_idx[t];
                        if (d < 0 || d >= ndof) throw Oops("dof_agglomeration: an element dof is outside the dof range");
                        keys.push_back(sep[d] ? (d | SEP) : d);
                    }
                }
                int64_t *u = tmp.data() + rd_ptr[(size_t)a];
                std::copy(keys.begin(), keys.end(), u);
                std::sort(u, u + keys.size());
                const int64_t cnt = std::unique(u, u + keys.size()) - u;
                nu[(size_t)a] = cnt;
                int64_t ni = 0;
                while (ni < cnt && u[ni] < SEP) ni++;
                n_int[a] = ni;
                int64_t *l = loc + rd_ptr[(size_t)a];
                for (size_t t = 0; t < keys.size(); t++) l[t] = std::lower_bound(u, u + cnt, keys[t]) - u;
            }
        });
        ae_dof_ptr[0] = 0;
        for (int64_t a = 0; a < nAE; a++) ae_dof_ptr[a + 1] = ae_dof_ptr[a] + nu[(size_t)a];
        parallel_ranges(nAE, [&](int64_t b, int64_t e) {
            for (int64_t a = b; a < e; a++) {
                const int64_t *u = tmp.data() + rd_ptr[(size_t)a];
                int64_t *o = ae_dofs + ae_dof_ptr[a];
                for (int64_t t = 0; t < nu[(size_t)a]; t++) o[t] = u[t] & ~SEP;
            }
        });
    });
}

// Coarse trace columns on the boundary of every agglomerate and the traces restricted to its boundary dofs.  The coarse
// element-facet table (B_ptr, B_idx) lists the agglomerated facets F of agglomerate a in column order; F carries the coarse dofs
// cf_ptr[F] .. cf_ptr[F+1].  cb_ptr (nAE+1) / cb (their concatenation) are written; R_b gets, for every agglomerate, the column-major
// (nb x next) block of trace values on its boundary dofs (positions n_int .. of its dof list): the value of coarse column c of
// facet F at fine dof d sits in the facet kernel's packed output p_loc at  af_d_ptr[F] * k + c * (dofs of F) + pos_in_af[d].
// Rb_ptr (nAE+1) is written too; cb must have room for cb_ptr[nAE] entries and R_b for Rb_ptr[nAE] (call with cb == NULL to get the sizes).
ELAG_HIP_API int elag_hip_host_amge_boundary_traces(int64_t nAE, const int64_t *B_ptr, const int64_t *B_idx, const int64_t *cf_ptr, const int64_t *af_of_dof,
                                                    const int64_t *pos_in_af, const int64_t *af_d_ptr, int64_t k, const double *p_loc,
                                                    const int64_t *ae_dof_ptr, const int64_t *ae_dofs, const int64_t *n_int, int64_t *cb_ptr, int64_t *Rb_ptr,
                                                    int64_t *cb, double *R_b)
{
    return guarded([&] {
        ELAG_REQUIRE(nAE >= 0 && B_ptr && B_idx && cf_ptr && ae_dof_ptr && n_int && cb_ptr && Rb_ptr, "boundary_traces: null argument");
        cb_ptr[0] = 0; Rb_ptr[0] = 0;
        for (int64_t a = 0; a < nAE; a++) {
            int64_t nx = 0;
            for (int64_t q = B_ptr[a]; q < B_ptr[a + 1]; q++) nx += cf_ptr[B_idx[q] + 1] - cf_ptr[B_idx[q]];
            cb_ptr[a + 1] = cb_ptr[a] + nx;
            const int64_t nb = (ae_dof_ptr[a + 1] - ae_dof_ptr[a]) - n_int[a];
            Rb_ptr[a + 1] = Rb_ptr[a] + nb * nx;
        }
        if (!cb) return;
        ELAG_REQUIRE(R_b && af_of_dof && pos_in_af && af_d_ptr && p_loc && ae_dofs, "boundary_traces: null argument");
        parallel_ranges(nAE, [&](int64_t b, int64_t e) {
            for (int64_t a = b; a < e; a++) {
                const int64_t nb = (ae_dof_ptr[a + 1] - ae_dof_ptr[a]) - n_int[a];
                const int64_t *bd = ae_dofs + ae_dof_ptr[a] + n_int[a];
                double *R = R_b + Rb_ptr[a];
                std::fill(R, R + (Rb_ptr[a + 1] - Rb_ptr[a]), 0.0);
                int64_t j = 0;
                for (int64_t q = B_ptr[a]; q < B_ptr[a + 1]; q++) {
                    const int64_t F = B_idx[q], nF = af_d_ptr[F + 1] - af_d_ptr[F];
                    for (int64_t c = 0; c < cf_ptr[F + 1] - cf_ptr[F]; c++, j++) {
                        cb[cb_ptr[a] + j] = cf_ptr[F] + c;
                        for (int64_t t = 0; t < nb; t++) {
                            const int64_t d = bd[t];
                            if (af_of_dof[d] == F) R[j * nb + t] = p_loc[af_d_ptr[F] * k + c * nF + pos_in_af[d]];
                        }
                    }
                }
            }
        });
    });
}

// Local triplets of a derivative on the agglomerates (DistributeAgglomerateMatrix): rows = the dofs of the range form of agglomerate a
// (row_ptr / row_dofs, local row = position in that list), columns = positions in the agglomerate's dof list of the domain form
// (ae_dof_ptr / ae_dofs / n_int as produced above: two ascending runs).  D is CSR with 32-bit arrays (scipy).  d_ptr (nAE+1) is always
// written; with d_row != NULL also the triplets, rows in order, entries in D's storage order.  A column outside the agglomerate is an error.
ELAG_HIP_API int elag_hip_host_amge_local_derivative(int64_t nAE, const int64_t *row_ptr, const int64_t *row_dofs, const int32_t *D_indptr,
                                                     const int32_t *D_indices, const double *D_data, const int64_t *ae_dof_ptr, const int64_t *ae_dofs,
                                                     const int64_t *n_int, int64_t *d_ptr, int32_t *d_row, int32_t *d_col, double *d_val)
{
    return guarded([&] {
        ELAG_REQUIRE(nAE >= 0 && row_ptr && row_dofs && D_indptr && D_indices && ae_dof_ptr && ae_dofs && n_int && d_ptr, "local_derivative: null argument");
        d_ptr[0] = 0;
        for (int64_t a = 0; a < nAE; a++) {
            int64_t s = 0;
            for (int64_t r = row_ptr[a]; r < row_ptr[a + 1]; r++) s += D_indptr[row_dofs[r] + 1] - D_indptr[row_dofs[r]];
            d_ptr[a + 1] = d_ptr[a] + s;
        }
        if (!d_row) return;
        ELAG_REQUIRE(d_col && d_val && D_data, "local_derivative: null argument");
        parallel_ranges(nAE, [&](int64_t b, int64_t e) {
            for (int64_t a = b; a < e; a++) {
                const int64_t *u = ae_dofs + ae_dof_ptr[a];
                const int64_t nu = ae_dof_ptr[a + 1] - ae_dof_ptr[a], ni = n_int[a];
                int64_t w = d_ptr[a];
                for (int64_t r = row_ptr[a]; r < row_ptr[a + 1]; r++) {
                    const int64_t rd = row_dofs[r];
                    for (int32_t q = D_indptr[rd]; q < D_indptr[rd + 1]; q++, w++) {
                        const int64_t c = D_indices[q];
                        const int64_t *p = std::lower_bound(u, u + ni, c);
                        if (p == u + ni || *p != c) {
                            p = std::lower_bound(u + ni, u + nu, c);
                            if (p == u + nu || *p != c) throw Oops("DistributeAgglomerateMatrix: D_2 couples an agglomerate with a dof outside it");
                        }
                        d_row[w] = (int32_t)(r - row_ptr[a]);
                        d_col[w] = (int32_t)(p - u);
                        d_val[w] = D_data[q];
                    }
                }
            }
        });
    });
}

}  // extern "C"

extern "C" {

// Local triplets of a sparse matrix D (32-bit CSR) on entities (DistributeAgglomerateMatrix for any pair of forms): entity e has the rows
// rows[rows_ptr[e]..) (local row = position) and the columns col_idx[col_ptr[e]..) in any order (local column = first position).  d_ptr
// (n_ent+1) is always written; with d_row != NULL the triplets too, rows in order, entries in D's storage order.  An entry whose column
// is not among the entity's columns is an error.
ELAG_HIP_API int elag_hip_host_local_triplets(int64_t n_ent, const int64_t *rows_ptr, const int64_t *rows, const int32_t *D_indptr, const int32_t *D_indices,
                                              const double *D_data, const int64_t *col_ptr, const int64_t *col_idx, int64_t *d_ptr, int64_t *d_row,
                                              int64_t *d_col, double *d_val)
{
    return guarded([&] {
        ELAG_REQUIRE(n_ent >= 0 && rows_ptr && D_indptr && col_ptr && d_ptr, "local_triplets: null argument");
        d_ptr[0] = 0;
        for (int64_t e = 0; e < n_ent; e++) {
            int64_t s = 0;
            for (int64_t r = rows_ptr[e]; r < rows_ptr[e + 1]; r++) s += D_indptr[rows[r] + 1] - D_indptr[rows[r]];
            d_ptr[e + 1] = d_ptr[e] + s;
        }
        if (!d_row) return;
        ELAG_REQUIRE(d_col && d_val && (d_ptr[n_ent] == 0 || (D_indices && D_data && col_idx)), "local_triplets: null argument");
        parallel_ranges(n_ent, [&](int64_t b, int64_t e_end) {
            RowFinder f;
            for (int64_t e = b; e < e_end; e++) {
                f.set(col_idx + col_ptr[e], col_ptr[e + 1] - col_ptr[e]);
                int64_t w = d_ptr[e];
                for (int64_t r = rows_ptr[e]; r < rows_ptr[e + 1]; r++)
                    for (int32_t q = D_indptr[rows[r]]; q < D_indptr[rows[r] + 1]; q++, w++) {
                        const int64_t at = f.find(D_indices[q]);
                        if (at < 0) throw Oops("DistributeAgglomerateMatrix: a derivative couples an agglomerate with a dof outside it");
                        d_row[w] = r - rows_ptr[e]; d_col[w] = at; d_val[w] = D_data[q];
                    }
            }
        }, grain_for(n_ent, 8 * d_ptr[n_ent]));
    });
}

// Element-to-dof table of form 2 on a level: element e carries the dofs of its facets (ef_ptr / ef_idx rows, facet f = dofs
// f_dof_ptr[f] .. f_dof_ptr[f+1]), facet after facet, then its own interior dofs nfd + e_idof_ptr[e] .. (DofHandler::GetrDof order of
// src/amge/DofHandler.cpp:548-640 for codimension 0).  ptr (nE+1) is always written; idx when it is not NULL (room for ptr[nE]).
ELAG_HIP_API int elag_hip_host_amge_elem_dof_table(int64_t nE, const int64_t *ef_ptr, const int64_t *ef_idx, const int64_t *f_dof_ptr,
                                                   const int64_t *e_idof_ptr, int64_t nfd, int64_t *ptr, int64_t *idx)
{
    return guarded([&] {
        ELAG_REQUIRE(nE >= 0 && ef_ptr && f_dof_ptr && e_idof_ptr && ptr && (ef_idx || ef_ptr[nE] == 0), "elem_dof_table: null argument");
        if (!idx) {
            ptr[0] = 0;
            for (int64_t e = 0; e < nE; e++) {
                int64_t s = e_idof_ptr[e + 1] - e_idof_ptr[e];
                for (int64_t q = ef_ptr[e]; q < ef_ptr[e + 1]; q++) s += f_dof_ptr[ef_idx[q] + 1] - f_dof_ptr[ef_idx[q]];
                ptr[e + 1] = ptr[e] + s;
            }
            return;
        }
        parallel_ranges(nE, [&](int64_t b, int64_t e_end) {
            for (int64_t e = b; e < e_end; e++) {
                int64_t w = ptr[e];
                for (int64_t q = ef_ptr[e]; q < ef_ptr[e + 1]; q++)
                    for (int64_t d = f_dof_ptr[ef_idx[q]]; d < f_dof_ptr[ef_idx[q] + 1]; d++) idx[w++] = d;
                for (int64_t d = e_idof_ptr[e]; d < e_idof_ptr[e + 1]; d++) idx[w++] = nfd + d;
            }
        });
    });
}

// Sparse matrix whose entity blocks are dense (mfem's SetSubMatrix over all entities, exact zeros dropped), straight to CSR with sorted
// 32-bit columns.  The blocks come in families (elag_hip_block_family): block e of a family couples rows rows_idx[rows_ptr[e]..) with
// columns cols_idx[cols_ptr[e]..); its values are the next (slot_rows ? slot_rows[e] : nrows_e) * ld[e] entries of vals (col_major: entry
// (r,c) at c*nrows_e + r; else at r*ld[e] + c).  Entries that meet in one (row, column) are added family by family, block by block.
// indptr (nrow+1) is always written; indices / data when not NULL (room for indptr[nrow] before duplicates merge; *out_nnz = entries
// kept, and indptr is rewritten if duplicates merged).
ELAG_HIP_API int elag_hip_host_block_families_to_csr(int n_fam, const elag_hip_block_family *fam, int64_t nrow, int64_t ncol, int64_t *indptr,
                                                     int32_t *indices, double *data, int64_t *out_nnz)
{
    return guarded([&] {
        ELAG_REQUIRE(n_fam >= 0 && (n_fam == 0 || fam) && indptr && nrow >= 0 && ncol >= 0 && ncol < (int64_t(1) << 31), "blocks_to_csr: bad argument");
        struct Ent { const int64_t *R, *C; const double *v; int64_t nr, nc, ld; bool cm; };
        int64_t n_ent = 0;
        for (int f = 0; f < n_fam; f++) {
            ELAG_REQUIRE(fam[f].n_ent >= 0 && fam[f].rows_ptr && fam[f].cols_ptr && fam[f].ld, "blocks_to_csr: null or negative argument in family %d", f);
            n_ent += fam[f].n_ent;
        }
        std::vector<Ent> ent((size_t)n_ent);
        {
            int64_t w = 0;
            for (int f = 0; f < n_fam; f++) {
                const elag_hip_block_family &F = fam[f];
                int64_t slot = 0;
                for (int64_t e = 0; e < F.n_ent; e++, w++) {
                    const int64_t nr = F.rows_ptr[e + 1] - F.rows_ptr[e], nc = F.cols_ptr[e + 1] - F.cols_ptr[e];
                    ELAG_REQUIRE(nr >= 0 && nc >= 0 && F.ld[e] >= nc, "blocks_to_csr: block %lld of family %d does not fit its slot", (long long)e, f);
                    ent[(size_t)w] = Ent{F.rows_idx + F.rows_ptr[e], F.cols_idx + F.cols_ptr[e], F.vals + slot, nr, nc, F.ld[e], F.col_major != 0};
                    slot += (F.slot_rows ? F.slot_rows[e] : nr) * F.ld[e];
                }
                ELAG_REQUIRE(slot <= F.n_vals, "blocks_to_csr: the value array of family %d is shorter than its slots", f);
            }
        }
        auto at = [](const Ent &E, int64_t r, int64_t c) { return E.v[E.cm ? c * E.nr + r : r * E.ld + c]; };
        if (!indices) {
            std::fill(indptr, indptr + nrow + 1, (int64_t)0);
            parallel_ranges(n_ent, [&](int64_t b, int64_t e_end) {
                for (int64_t e = b; e < e_end; e++) {
                    const Ent &E = ent[(size_t)e];
                    const int64_t nr = E.nr, nc = E.nc;
                    const int64_t *R = E.R;
                    for (int64_t r = 0; r < nr; r++) {
                        if (R[r] < 0 || R[r] >= nrow) throw Oops("blocks_to_csr: a row index is outside the matrix");
                        int64_t n = 0;
                        for (int64_t c = 0; c < nc; c++) n += at(E, r, c) != 0.0;
                        if (n) __atomic_fetch_add(&indptr[R[r] + 1], n, __ATOMIC_RELAXED);
                    }
                }
            });
            for (int64_t i = 0; i < nrow; i++) indptr[i + 1] += indptr[i];
            return;
        }
        ELAG_REQUIRE(data && out_nnz, "blocks_to_csr: null argument");
        std::vector<int64_t> cur;
        // a row's entries land in whatever order the threads reach it and are sorted by column afterwards; only when two blocks set the
        // same (row, column) does the order matter (the sum), and then the fill is repeated on one thread, in block order
        auto fill = [&](int64_t b, int64_t e_end, bool shared) {
            for (int64_t e = b; e < e_end; e++) {
                const Ent &E = ent[(size_t)e];
                const int64_t nr = E.nr, nc = E.nc;
                const int64_t *R = E.R, *Cc = E.C;
                for (int64_t r = 0; r < nr; r++) {
                    int64_t n = 0;
                    for (int64_t c = 0; c < nc; c++) n += at(E, r, c) != 0.0;
                    if (!n) continue;
                    int64_t w = shared ? __atomic_fetch_add(&cur[(size_t)R[r]], n, __ATOMIC_RELAXED) : cur[(size_t)R[r]];
                    if (!shared) cur[(size_t)R[r]] += n;
                    for (int64_t c = 0; c < nc; c++) {
                        const double x = at(E, r, c);
                        if (x != 0.0) {
                            if (Cc[c] < 0 || Cc[c] >= ncol) throw Oops("blocks_to_csr: a column index is outside the matrix");
                            indices[w] = (int32_t)Cc[c]; data[w] = x; w++;
                        }
                    }
                }
            }
        };
        std::atomic<int> dup{0};
        auto sort_rows = [&] {
            parallel_ranges(nrow, [&](int64_t b, int64_t e_end) {
                std::vector<std::pair<int32_t, double>> tmp;
                for (int64_t i = b; i < e_end; i++) {
                    const int64_t lo = indptr[i], n = indptr[i + 1] - lo;
                    bool sorted = true;
                    for (int64_t t = 1; t < n; t++)
                        if (indices[lo + t] <= indices[lo + t - 1]) { sorted = false; break; }
                    if (sorted) continue;
                    tmp.resize((size_t)n);
                    for (int64_t t = 0; t < n; t++) tmp[(size_t)t] = {indices[lo + t], data[lo + t]};
                    std::stable_sort(tmp.begin(), tmp.end(),
                                     [](const std::pair<int32_t, double> &x, const std::pair<int32_t, double> &y) { return x.first < y.first; });
                    for (int64_t t = 0; t < n; t++) {
                        indices[lo + t] = tmp[(size_t)t].first; data[lo + t] = tmp[(size_t)t].second;
                        if (t && tmp[(size_t)t].first == tmp[(size_t)t - 1].first) dup.store(1, std::memory_order_relaxed);
                    }
                }
            });
        };
        cur.assign(indptr, indptr + nrow);
        parallel_ranges(n_ent, [&](int64_t b, int64_t e_end) { fill(b, e_end, true); });
        sort_rows();
        if (dup.load()) {
            cur.assign(indptr, indptr + nrow);
            fill(0, n_ent, false);
            sort_rows();
        }
        int64_t nnz = indptr[nrow];
        if (dup.load()) {
            int64_t w = 0, lo = 0;
            for (int64_t i = 0; i < nrow; i++) {
                const int64_t hi = indptr[i + 1];
                for (int64_t t = lo; t < hi; t++) {
                    if (t > lo && indices[t] == indices[w - 1]) data[w - 1] += data[t];
                    else { indices[w] = indices[t]; data[w] = data[t]; w++; }
                }
                lo = hi;
                indptr[i + 1] = w;
            }
            nnz = w;
        }
        *out_nnz = nnz;
    });
}

// One family (the layout of elag_hip_host_blocks_to_coo).
ELAG_HIP_API int elag_hip_host_blocks_to_csr(int64_t n_ent, const int64_t *rows_ptr, const int64_t *rows_idx, const int64_t *cols_ptr,
                                             const int64_t *cols_idx, const double *vals, const int64_t *ld, const int64_t *slot_rows, int col_major,
                                             int64_t nrow, int64_t ncol, int64_t *indptr, int32_t *indices, double *data, int64_t *out_nnz)
{
    elag_hip_block_family F{n_ent, rows_ptr, rows_idx, cols_ptr, cols_idx, vals, INT64_MAX, ld, slot_rows, col_major};
    return elag_hip_host_block_families_to_csr(1, &F, nrow, ncol, indptr, indices, data, out_nnz);
}

}  // extern "C"
