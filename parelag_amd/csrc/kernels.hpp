// Host-callable launchers of the hand-written gfx950 kernels (defined in kernels_*.hip).
// Everything is enqueued on ctx().stream unless a stream is passed explicitly.
#pragma once
#include <functional>
#include <cstdint>
#include <vector>
#include "common.hpp"

namespace elag {
namespace k {

// -------------------------------------------------------------------------------------------------
// Fused CSR SpMV family.  One kernel computes, per local row i,
//     sum  = sum_j A_ij * x_j * (cs ? cs_j : 1)  [+ sum_j Offd_ij * xg_j * (csg ? csg_j : 1)]
//     out_i = beta * yin_i + (s ? s_i : 1) * (gamma * f_i + alpha * (s2 ? s2_i : 1) * sum)
//     out2_i = x_i * (cs ? cs_i : 1)                      (optional; square matrices only)
// which covers y = alpha A x + beta y, r = b - A x, the l1-Jacobi sweep, the zero-guess pre-smooth
// fused with the residual, and every Chebyshev step (see smoother.cpp).  Null pointers mean 0 / 1.
// -------------------------------------------------------------------------------------------------
struct SpmvArgs {
    const double *x = nullptr, *cs = nullptr;
    const double *xg = nullptr, *csg = nullptr;
    const double *yin = nullptr;
    double beta = 0.0;
    const double *f = nullptr;
    double gamma = 0.0;
    const double *s = nullptr, *s2 = nullptr;
    double alpha = 1.0;
    double *out = nullptr, *out2 = nullptr;
    const double *o2s = nullptr;  // out2[i] = x[i] * o2s[i] (without cs; with cs out2 is the scaled operand x[i] * cs[i]): the operand of a product whose matrix carries the column scaling
    // row subset (multi-GPU overlap): rows == nullptr -> all rows [0,nrows)
    const int *rows = nullptr;
    int nrows_sel = 0;
    int skip_rows_with_offd = 0;  // interior pass: rows whose offd row is non-empty are left untouched
    int xcd_chunks = 0;           // x-cache kernel: window = f(workgroup id) that keeps an XCD on one contiguous run of windows (set by the launcher)
    // stream / x-cache kernels only: the result row i goes to out[out_perm[i]] (the hierarchy's last kernel writes the caller's
    // numbering directly instead of a scatter pass after it); with out_add the value is added to out_add[out_perm[i]] on the way
    const int *out_perm = nullptr;
    const double *out_add = nullptr;
};
void spmv_fused(const DevCSR &A, const DevCSR *offd, const SpmvArgs &args, hipStream_t st = nullptr);
// A dependent chain of launch-bound products as ONE kernel with grid-wide barriers in between (kernels_spmv.hip, "Resident sequence").
// Pointers equal to resident_rhs() / resident_sol() in an op's arguments stand for the vectors given at launch time.
struct ResidentOp {
    int kind = 0;                    // 0: CSR row-group product (spmv_fused on the row-group kernel), 1 / 2: dense product (workgroup / wavefront per row)
    int lpr = 1, has_offd = 0, has_cs = 0;
    int nrows = 0, ncols = 0;
    const int *I = nullptr, *J = nullptr;
    const double *a = nullptr;
    const int *oI = nullptr, *oJ = nullptr;
    const double *oa = nullptr;
    const double *M = nullptr;       // dense, row-major nrows x ncols
    SpmvArgs p;
};
struct ResidentSeq {
    ResidentOp *d_ops = nullptr;
    int nops = 0;
    unsigned *d_bar = nullptr;       // arrivals, generation
    int *d_err = nullptr;            // set when a barrier wait ran out
    unsigned grid = 0;
};
constexpr uintptr_t kResidentRhs = 8, kResidentSol = 16;
inline const double *resident_rhs() { return reinterpret_cast<const double *>(kResidentRhs); }
inline double *resident_sol() { return reinterpret_cast<double *>(kResidentSol); }
bool resident_op_spmv(const DevCSR &A, const DevCSR *offd, const SpmvArgs &args, ResidentOp &op);   // false: spmv_fused would not take the row-group kernel
void resident_op_dense(int nrows, int ncols, const double *M, const double *x, double *y, ResidentOp &op);
void resident_build(const std::vector<ResidentOp> &ops, ResidentSeq &seq);
void resident_free(ResidentSeq &seq);
void resident_launch(const ResidentSeq &seq, const double *rhs, double *sol, hipStream_t st = nullptr);
int resident_error(const ResidentSeq &seq);     // synchronous read of the error flag
// ---- "flat stage": several INDEPENDENT products in ONE launch (kernels_spmv.hip) --------------------------------------------
// A launch-bound cycle costs what its chain of dependent launches costs (~4 us each on MI355X, whatever their size), so the flattened
// form of a cycle (hierarchy.hip, plan_flat) puts everything that depends on the same inputs into one launch.  Two shapes:
//   FlatSegs: up to five segments, each  out = A x  (CSR with its own lanes per row, or a dense row-major matrix), the workgroups of
//             the launch dealt to the segments by ranges of blockIdx;
//   FlatSum:  out = yin + sum_t A_t x_t  over up to four CSR terms with common rows.
// Both are structs of arrays passed by value: a workgroup finds everything it needs in the kernel-argument preload (a descriptor
// fetched through a pointer, or a dynamically indexed array of structs, costs every wave one more dependent round trip -- measured:
// 7 us per stage instead of 4).
constexpr int kFlatSegs = 5, kFlatTerms = 4;
struct FlatSegs {
    int nsegs = 0;
    unsigned grid = 0;
    unsigned block_begin[kFlatSegs] = {0, 0, 0, 0, 0};
    int nrows[kFlatSegs] = {0, 0, 0, 0, 0};
    int lpr[kFlatSegs] = {1, 1, 1, 1, 1};          // lanes per row: 1..64, 256 = a workgroup per row, 0 = dense (a = M row-major, ncols wide)
    int ncols[kFlatSegs] = {0, 0, 0, 0, 0};
    const int *I[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const int *J[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const double *a[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const double *x[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double *out[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const int *pJ[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};        // the segment's flat pack (DevCSR::fpJ / fpa) or null
    const double *pa[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const unsigned short *j16p[kFlatSegs] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // DevCSR::J16 of the segment's operator (null: none)
};
struct FlatSum {
    int nrows = 0, nterms = 0, lpr = 1;
    unsigned grid = 0;
    const int *I[kFlatTerms] = {nullptr, nullptr, nullptr, nullptr};
    const int *J[kFlatTerms] = {nullptr, nullptr, nullptr, nullptr};
    const double *a[kFlatTerms] = {nullptr, nullptr, nullptr, nullptr};
    const double *x[kFlatTerms] = {nullptr, nullptr, nullptr, nullptr};
    const double *yin = nullptr;
    double *out = nullptr;
    const unsigned short *j16p[kFlatTerms] = {nullptr, nullptr, nullptr, nullptr};            // DevCSR::J16 of the term's operator (null: none)
};
void flat_segs_add_csr(FlatSegs &S, const DevCSR &A, const double *x, double *out);       // lanes from the rows' average length
void flat_segs_add_dense(FlatSegs &S, int nrows, int ncols, const double *M, const double *x, double *out);
void flat_sum_add(FlatSum &S, const DevCSR &A, const double *x);                          // call flat_sum_finish after the last term
void flat_sum_finish(FlatSum &S, const double *yin, double *out);
// pointers equal to resident_rhs() / resident_sol() stand for the vectors given here
void flat_segs_launch(const FlatSegs &S, const double *rhs, double *sol, hipStream_t st = nullptr);
void flat_sum_launch(const FlatSum &S, const double *rhs, double *sol, hipStream_t st = nullptr);
// row-block plan of the LDS-staged stream kernel (block_row, nblocks, max_row_nnz); synchronous, setup time
void build_stream_plan(DevCSR &A);
// builds the x-cache form of a stream-kernel matrix if it pays (synchronous, allocates: never call while capturing);
// spmv_fused does this lazily on the first uncaptured launch, hierarchies do it for their operators at Finalize
// Kernels live in per-source code objects that the HIP runtime loads when the first kernel of a source is used: the first sparse product
// of a process paid ~10 ms for it inside what a caller times as hierarchy setup (cube456 x3: 15.7 ms against 5.0 ms for every later
// build).  elag_hip_init touches one kernel of each large source instead (once per process).
void preload_setup_kernels();   // kernels_setup.hip
void preload_spmv_kernels();    // kernels_spmv.hip
void preload_dense_kernels();   // dense.hip
void prepare_spmv(const DevCSR &A, bool force = false);
// the stream / row-group kernels as they are: no window plan is sampled or built at the first launch (state -2: a later prepare_spmv(), i.e. a
// Finalize that does invest, examines the operator as if nobody had looked at it)
inline void decline_spmv_forms(const DevCSR &A) { if (A.xc_state == 0) A.xc_state = -2; }
void longest_row(int nrows, const int *I, int *d_out);      // *d_out = max_r I[r + 1] - I[r] (device scalar, compute stream)

// builds the 4-byte-per-entry sign form of a matrix whose stored values are all -1, 0 or +1 (returns false and leaves the
// matrix alone otherwise); synchronous, setup time.  The caller vouches that the values stay as they are.
bool prepare_sign_form(const DevCSR &A);
// x_out = x_in + D (wa .* (D^T pv)) in one launch for a sign-form D and its sign-form transpose DT (false: a form is missing, nothing launched)
bool sign_twohop(const DevCSR &D, const DevCSR &DT, const double *wa, const double *pv, const double *xin, double *xout);

// ---- vector kernels ----------------------------------------------------------------------------
void vec_axpby(int n, double a, const double *x, double b, const double *y, double *out);  // out = a x + b y
void vec_scale_mul(int n, double a, const double *x, const double *d, double *out);        // out = a x .* d
void vec_add_scaled_mul(int n, const double *y, const double *d, const double *z, double c, const double *r,
                        double *out);  // out = y + d .* (c r + z)   (z or r may be null)
void vec_fill(int n, double v, double *x);
void vec_dot(int n, const double *x, const double *y, double *d_result);  // deterministic 2-stage
void vec_gather(int n, const int *idx, const double *x, double *buf, hipStream_t st = nullptr);
void vec_pack_copy(int n, int nsend, const int *idx, const double *x, double *send_buf, double *copy);   // send_buf = x[idx] (nsend entries) and copy = x (n entries) in one launch
void vec_gather_scale(int n, const int *idx, const double *x, const double *d, double *buf, double *scaled);  // buf = x[idx], scaled = d .* buf
void vec_scatter(int n, const int *idx, const double *src, double *dst, int add);  // dst[idx[i]] (+)= src[i], idx a permutation
void vec_hypre_random(int n, int seed, double *out);  // Lehmer 16807 stream, 2 r - 1 (host generated)

// ---- row statistics ----------------------------------------------------------------------------
// option as hypre_ParCSRComputeL1Norms (1,2,3,4) incl. the negative-diagonal sign flip.  d_bad counts zero norms.
void l1_norms(const DevCSR &diag, const DevCSR *offd, int option, double *l1, int *d_bad);
void extract_diagonal(const DevCSR &diag, double *d);  // 0 where the row stores no diagonal entry
void vec_rsqrt(int n, const double *d, double *out, int *d_bad);            // 1/sqrt(d), counts d <= 0
void vec_reciprocal_scale(int n, double w, const double *d, double *out);   // w / d
int fix_zero_rows(DevCSR &diag, DevCSR *offd, bool count = true);   // the number of rows fixed (0 when count is false: no read-back)
void values_abs(int64_t n, const double *a, double *out);
void values_sign(int64_t n, double *a);
double reduce_max_abs(int64_t n, const double *a);
double reduce_sum_sq(int64_t n, const double *a);
double reduce_max(int64_t n, const double *a);

// ---- sparse setup (two-pass, wave-level) -------------------------------------------------------
void exclusive_scan(const int *in, int *out, int n, int *d_max = nullptr, int *d_total = nullptr);  // out has n+1 entries, out[n] = total; d_max (zeroed by the caller): the largest element, d_total: a copy of the total
void csr_transpose(const DevCSR &A, DevCSR &AT);
void csr_sort_rows(DevCSR &A);
// C = A*B, columns ascending.  spmv_plan = false for an intermediate that is only ever an operand of another product (no SpMV
// dispatch plan is built for it)
void csr_spgemm(const DevCSR &A, const DevCSR &B, DevCSR &C, bool spmv_plan = true);
// C = R (A P) with A P never leaving the layout its kernels wrote it in (row r: its entries, ascending columns, from the scanned upper
// bound of its products on): as the intermediate of a triple product it needs neither the scan of the rows' entry counts, nor the host
// round trip for their total, nor the compaction into CSR -- a quarter of A P's time at cube456 x5.  Falls back to two plain products
// where the first one's rows do not all fit that layout.
void csr_rap(const DevCSR &R, const DevCSR &A, const DevCSR &P, DevCSR &C, bool spmv_plan = true);
void csr_add(double a, const DevCSR &A, double b, const DevCSR &B, DevCSR &C);
void csr_delete_zeros(const DevCSR &A, double tol, DevCSR &out);
void csr_scale_rows(DevCSR &A, const double *d);
void csr_scale_cols(DevCSR &A, const double *d);  // a_ij *= d_j
void csr_copy(const DevCSR &A, DevCSR &out);
void csr_diagonal(int n, const double *d, double scale, DevCSR &out);
// In place, for operators whose pattern contains the other operand's (both with ascending columns):
//     C_ij <- c_scale * (c_row ? c_row[i] : 1) * C_ij * (c_col ? c_col[j] : 1)
//             + (B ? b_scale * B_ij * (b_row ? (b_row_divide ? 1 / b_row[i] : b_row[i]) : 1) : 0)  +  (diag && i == j ? d_scale * diag[i] : 0)
// One launch instead of copy / scale / scale / sparse add through the ESC engine (~0.1 against ~0.4 ms on the level operators of a small
// hierarchy).  Returns false -- C is then in an unspecified state -- when an entry of B (or a requested diagonal entry) has no place in C.
struct CsrCombine {
    double c_scale = 1.0; const double *c_row = nullptr, *c_col = nullptr;
    const DevCSR *B = nullptr; double b_scale = 1.0; const double *b_row = nullptr; int b_row_divide = 0;
    const double *diag = nullptr; double d_scale = 0.0;
};
bool csr_combine_inplace(DevCSR &C, const CsrCombine &op);   // out = scale * diag(d)  (d == nullptr: scale * I)
// Renumbering of a level under its coarse level's (hierarchy.hip, `reorder`): children contiguous under their strongest coarse parent.
// d_order_out[new] = old for the nf rows of P: a stable counting sort of the rows by rank_c[argmax_k |P_ik|] (first maximum in
// storage order; rows without entries last) -- the sort is a transpose of the (row -> key) incidence, whose rows come back ascending.
void child_order(const DevCSR &P, const int *d_rank_c, int *d_order_out);
void inverse_permutation(int n, const int *d_order, int *d_rank);     // rank[order[r]] = r
// out = A(order_rows, :) with columns renumbered by rank_cols (old -> new); rows come back column-sorted
void csr_permute(const DevCSR &A, const int *d_order_rows, const int *d_rank_cols, int ncols_out, DevCSR &out, bool sort_rows = true);

// ---- dense coarse solve ------------------------------------------------------------------------
void dense_gemv(int nrows, int ncols, const double *M, const double *x, double *y);  // y = M x, M row-major
void dense_transpose(int n, const double *in, double *out);                           // out = in^T (n x n)
// One level of the dense coarse-cycle operator (hierarchy.hip, plan_dense_tail): with the level maps f_c = G f, x = U f + V x_c and the
// dense operator Bc (nc x nc) of the levels below, the level's own operator is  B = U + V (Bc G)  (n x n, row-major):
//   dense_times_csr_t : X (nc x n) = Bc GT^T, GT = G^T in CSR (n rows), one thread per entry of X;
//   dense_level_op    : B = rs . (V X) + U-part, rs_i = vw ? -vw_i : 1;  U-part: U given -> B += U;  else (A, w) -> B += 2 diag(w) - diag(w) A diag(w)
void dense_times_csr_t(int nc, int n, const double *Bc, const DevCSR &GT, double *X);
// dense blocks of m columns, row-major: out = (add_identity ? I : 0) + beta out + alpha S X;  out = M X
void csr_times_dense(const DevCSR &S, int m, double alpha, const double *X, double beta, bool add_identity, double *out);
void dense_mm(int nr, int nk, int m, const double *M, const double *X, double *out);
void dense_level_op(int n, int nc, const DevCSR &V, const double *vw, const double *X, const DevCSR *U, const DevCSR *A, const double *w, double *B);
// Minv (n x n row-major, device) = A^-1 for a general square CSR (Gauss-Jordan, partial pivoting); throws when singular
// An inverse that runs beside the rest of a build: with a job whose `stream` is set, the launches go to that stream (ordered behind what the
// compute stream has been given so far), nothing is read back, and `finish` -- to be called before anybody reads Minv -- waits for the
// stream, looks at the flag and gives the scratch back.  Forms that do not offer it run as without a job and leave `finish` empty.
struct DenseJob {
    hipStream_t stream = nullptr;
    std::function<void()> finish;
};
void dense_inverse_general(const DevCSR &A, double *Minv, DenseJob *job = nullptr);
// Minv = A^-1 for a symmetric positive definite CSR (Gauss-Jordan without pivoting on the device, result symmetrised); throws when a pivot is not positive
void dense_inverse_spd(const DevCSR &A, double *Minv, DenseJob *job = nullptr);

}  // namespace k
}  // namespace elag
