// Blocked saddle-point layer on the device.  Reference:
//   MfemBlockOperator            src/linalg/utilities/ParELAG_MfemBlockOperator.cpp:93-130
//   BlockDiagonalSolver          src/linalg/solver_ops/ParELAG_BlockDiagonalSolver.cpp:67-112
//   BlockTriangularSolver        src/linalg/solver_ops/ParELAG_BlockTriangularSolver.cpp:73-300
//   Block2x2LDUInverseOperator   src/linalg/solver_ops/ParELAG_Block2x2LDUInverseOperator.cpp:73-137
//   SchurComplementFactory       src/linalg/factories/ParELAG_SchurComplementFactory.cpp:36-181
//   buildBlockedHierarchyFromDeRhamSequence  src/linalg/solver_ops/ParELAG_Hierarchy.cpp:400-541
#include "blocked.hpp"

#include <algorithm>
#include <numeric>

#include "comm.hpp"
#include "staging.hpp"

using namespace elag;

namespace {

constexpr int kB = 256;
inline unsigned g1(long long n) { return (unsigned)std::max<long long>(1, (n + kB - 1) / kB); }

// global column id of one space -> global column id of the merged operator: the shift is constant per owner rank
__global__ void remap_owner_shift_kernel(int64_t nnz, int *__restrict__ J, int nr, const int *__restrict__ starts, const int *__restrict__ shift)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int j = J[k];
    int lo = 0, hi = nr;   // last q with starts[q] <= j
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= j) lo = mid; else hi = mid; }
    J[k] = j + shift[lo];
}

void check_rc(int rc)
{
    if (rc) fail(rc, "%s", elag_hip_last_error());
}

}  // namespace

// ---- block operator -------------------------------------------------------------------------------------------------
elag_hip_blockop_s::~elag_hip_blockop_s()
{
    for (size_t k = 0; k < blk.size(); k++)
        if (owned[k]) delete blk[k];
}

void elag_hip_blockop_s::set_block(int i, int j, ParCSR *M, bool own)
{
    ELAG_REQUIRE(i >= 0 && i < nbr && j >= 0 && j < nbc, "MfemBlockOperator::SetBlock(%d,%d): block index out of range (%d x %d blocks)", i, j, nbr, nbc);
    ELAG_REQUIRE(M, "null block");
    // the size check of MfemBlockOperator::SetBlock (ParELAG_MfemBlockOperator.cpp:107-118)
    ELAG_REQUIRE(M->nrows() == row_off[(size_t)i + 1] - row_off[(size_t)i] && M->ncols() == col_off[(size_t)j + 1] - col_off[(size_t)j],
                 "MfemBlockOperator::SetBlock(...): Input operator has wrong size (size = %dx%d; should be %dx%d)", M->nrows(), M->ncols(),
                 row_off[(size_t)i + 1] - row_off[(size_t)i], col_off[(size_t)j + 1] - col_off[(size_t)j]);
    const size_t k = (size_t)i * nbc + j;
    if (owned[k]) delete blk[k];
    blk[k] = M; owned[k] = own ? 1 : 0;
    mono.reset();
}

ParCSR *elag_hip_blockop_s::monolithic()
{
    if (mono) return mono.get();
    Context &c = ctx();
    hipStream_t st = c.stream;
    const int nr = c.nranks;
    // every block row / column needs one stored block to define its partition
    std::vector<ParCSR *> rrep((size_t)nbr, nullptr), crep((size_t)nbc, nullptr);
    bool dist = false;
    for (int i = 0; i < nbr; i++)
        for (int j = 0; j < nbc; j++)
            if (ParCSR *M = block(i, j)) {
                if (!rrep[(size_t)i]) rrep[(size_t)i] = M;
                if (!crep[(size_t)j]) crep[(size_t)j] = M;
                ELAG_REQUIRE(M->first_row == rrep[(size_t)i]->first_row && M->global_rows == rrep[(size_t)i]->global_rows,
                             "blocks of block row %d do not share a row partition", i);
                ELAG_REQUIRE(M->first_col == crep[(size_t)j]->first_col && M->global_cols == crep[(size_t)j]->global_cols,
                             "blocks of block column %d do not share a column partition", j);
                if (is_distributed(*M)) dist = true;
            }
    for (int i = 0; i < nbr; i++) ELAG_REQUIRE(rrep[(size_t)i] || row_off[(size_t)i + 1] == row_off[(size_t)i], "block row %d has no stored block", i);
    for (int j = 0; j < nbc; j++) ELAG_REQUIRE(crep[(size_t)j] || col_off[(size_t)j + 1] == col_off[(size_t)j], "block column %d has no stored block", j);
    const int nloc = row_off.back(), mloc = col_off.back();

    int64_t first_row = 0, grows = 0, first_col = 0, gcols = 0;
    std::vector<int *> dtab;   // device tables of the distributed path
    std::vector<const int *> d_starts((size_t)nbc, nullptr), d_shift((size_t)nbc, nullptr);
    if (dist) {
        ELAG_REQUIRE(nr > 1, "distributed blocks need an initialised communicator");
        for (int i = 0; i < nbr; i++) if (rrep[(size_t)i]) { first_row += rrep[(size_t)i]->first_row; grows += rrep[(size_t)i]->global_rows; }
        std::vector<std::vector<int>> starts((size_t)nbc);
        std::vector<int> mono_first((size_t)nr, 0);
        for (int j = 0; j < nbc; j++) {
            const double fc = crep[(size_t)j] ? (double)crep[(size_t)j]->first_col : 0.0;
            std::vector<double> s = comm_allgather_scalar(fc);
            starts[(size_t)j].assign(s.begin(), s.end());
            starts[(size_t)j].push_back(crep[(size_t)j] ? (int)crep[(size_t)j]->global_cols : 0);
            for (int q = 0; q < nr; q++) mono_first[(size_t)q] += starts[(size_t)j][(size_t)q];
            if (crep[(size_t)j]) { first_col += crep[(size_t)j]->first_col; gcols += crep[(size_t)j]->global_cols; }
        }
        ELAG_REQUIRE(gcols < (int64_t(1) << 31) && grows < (int64_t(1) << 31), "global sizes beyond 2^31 are not supported");
        std::vector<int> before((size_t)nr, 0);   // columns of the earlier spaces on rank q
        for (int j = 0; j < nbc; j++) {
            std::vector<int> shift((size_t)nr);
            for (int q = 0; q < nr; q++) {
                shift[(size_t)q] = mono_first[(size_t)q] + before[(size_t)q] - starts[(size_t)j][(size_t)q];
                before[(size_t)q] += starts[(size_t)j][(size_t)q + 1] - starts[(size_t)j][(size_t)q];
            }
            dtab.push_back(dev_from_host(starts[(size_t)j].data(), (size_t)nr + 1));
            d_starts[(size_t)j] = dtab.back();
            dtab.push_back(dev_from_host(shift.data(), (size_t)nr));
            d_shift[(size_t)j] = dtab.back();
            ELAG_HIP_CHECK(hipStreamSynchronize(st));   // host tables go out of scope
        }
    } else {
        grows = nloc; gcols = mloc;
    }

    DevCSR all;
    bool have_all = false;
    for (int i = 0; i < nbr; i++) {
        const int ni = row_off[(size_t)i + 1] - row_off[(size_t)i];
        DevCSR row;
        bool have_row = false;
        for (int j = 0; j < nbc; j++) {
            ParCSR *M = block(i, j);
            if (!M) continue;
            DevCSR G, next;
            const DevCSR *piece = &M->diag;
            int off = col_off[(size_t)j];
            if (dist) {
                parcsr_to_global(*M, G);
                if (G.nnz)
                    hipLaunchKernelGGL(remap_owner_shift_kernel, dim3(g1(G.nnz)), dim3(kB), 0, st, G.nnz, G.J, nr, d_starts[(size_t)j], d_shift[(size_t)j]);
                ELAG_LAUNCH_CHECK();
                piece = &G; off = 0;
            }
            const int width = dist ? (int)gcols : mloc;
            if (!have_row) csr_hstack(*piece, nullptr, off, nullptr, nullptr, 0, width, dist, next);
            else csr_hstack(row, nullptr, 0, piece, nullptr, off, width, dist, next);
            ELAG_HIP_CHECK(hipStreamSynchronize(st));
            row = std::move(next);
            have_row = true;
        }
        if (!have_row) row.set_empty(ni, dist ? (int)gcols : mloc);
        if (!have_all) { all = std::move(row); have_all = true; }
        else { DevCSR st2; csr_vstack(all, row, st2); ELAG_HIP_CHECK(hipStreamSynchronize(st)); all = std::move(st2); }
    }
    if (!have_all) all.set_empty(0, dist ? (int)gcols : mloc);
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    for (int *t : dtab) dev_free(t);
    if (dist) mono.reset(parcsr_from_global_columns(all, first_row, grows, first_col, mloc, gcols));
    else mono.reset(parcsr_from_diag(std::move(all), 0, 0, grows, gcols));
    return mono.get();
}

// ---- Schur complement -----------------------------------------------------------------------------------------------
static ParCSR *schur_complement(elag_hip_blockop_s &B, int type, double alpha, bool negate)
{
    ELAG_REQUIRE(B.nbr == 2 && B.nbc == 2, "SchurComplementFactory needs a 2x2 block operator");
    ELAG_REQUIRE(type == ELAG_HIP_SCHUR_DIAGONAL || type == ELAG_HIP_SCHUR_ABSROWSUM,
                 "Schur complement type = %d is invalid. Valid device types are DIAGONAL (1) and ABSROWSUM (2)", type);
    ParCSR *A00 = B.block(0, 0), *A01 = B.block(0, 1), *A10 = B.block(1, 0), *A11 = B.block(1, 1);
    ELAG_REQUIRE(A00 && A01 && A10, "Schur complement needs the blocks A00, A01 and A10");   // :74 "I need three of the blocks"
    const int n0 = A00->nrows();
    Temp<double> d((size_t)n0), dinv((size_t)n0);
    if (type == ELAG_HIP_SCHUR_DIAGONAL) {
        k::extract_diagonal(A00->diag, d);
    } else {
        Temp<int> bad(1);
        zero_async(bad.p, sizeof(int));
        k::l1_norms(A00->diag, &A00->offd, 1, d, bad);   // +-(sum |diag row| + sum |offd row|)
        k::values_abs(n0, d, d);
    }
    k::vec_reciprocal_scale(n0, 1.0, d, dinv);
    // tmp = diag^-1 A01 (InvScaleRows), product = A10 * tmp
    std::unique_ptr<ParCSR> T(parcsr_clone(*A01));
    k::csr_scale_rows(T->diag, dinv);
    if (T->offd.nnz) k::csr_scale_rows(T->offd, dinv);
    elag_hip_parcsr_t prod = nullptr;
    check_rc(elag_hip_parcsr_matmul(A10, T.get(), &prod));
    std::unique_ptr<ParCSR> P(prod);
    const double sgn = negate ? -1.0 : 1.0;
    if (A11) {
        elag_hip_parcsr_t S = nullptr;
        check_rc(elag_hip_parcsr_add(sgn, A11, -sgn * alpha, P.get(), &S));
        return S;
    }
    const double f = -sgn * alpha;
    if (P->diag.nnz) k::vec_axpby((int)P->diag.nnz, f, P->diag.a, 0.0, P->diag.a, P->diag.a);
    if (P->offd.nnz) k::vec_axpby((int)P->offd.nnz, f, P->offd.a, 0.0, P->offd.a, P->offd.a);
    return P.release();
}

// ---- block smoothers --------------------------------------------------------------------------------------------------
namespace elag {

BlockSmoother::~BlockSmoother() { dev_free(resid); dev_free(corr); dev_free(tmp); dev_free(spare); dev_free(dscale); }

bool BlockSmoother::same_scaling_as(const elag_hip_smoother_s *o) const { return o == this; }

// a block solver replays inside a graph when every product with its blocks is rank-local and every one of its inverses says it may
bool BlockSmoother::graph_capturable() const
{
    if (!A || is_distributed(*A)) return false;
    for (ParCSR *blk : B->blk)
        if (blk && is_distributed(*blk)) return false;
    for (auto *s : inv)
        if (!s || !s->graph_capturable()) return false;
    return true;
}

void BlockSmoother::setup()
{
    ELAG_REQUIRE(B->nbr == B->nbc, "block smoothers need a block-square operator");
    A = B->monolithic();
    const int nb = B->nbr;
    if (kind == ELAG_HIP_BLOCK_LDU) {
        ELAG_REQUIRE(nb == 2 && inv.size() == 4, "Block LDU needs a 2x2 operator and the four inverses {A00_1, A00_2, A00_3, S}");
        ELAG_REQUIRE(B->block(0, 1) && B->block(1, 0), "Block LDU needs the off-diagonal blocks");
        for (int q = 0; q < 3; q++) ELAG_REQUIRE(inv[(size_t)q]->n() == B->row_off[1], "A00 inverse %d has the wrong size", q + 1);
        ELAG_REQUIRE(inv[3]->n() == B->row_off[2] - B->row_off[1], "S inverse has the wrong size");
        fold = !(getenv("ELAG_HIP_LDU_FOLD") && !atoi(getenv("ELAG_HIP_LDU_FOLD")));
    } else {
        ELAG_REQUIRE(kind == ELAG_HIP_BLOCK_JACOBI || kind == ELAG_HIP_BLOCK_GS_LOWER || kind == ELAG_HIP_BLOCK_GS_UPPER, "unknown block smoother kind %d", kind);
        ELAG_REQUIRE((int)inv.size() == nb, "block smoother needs one inverse per block (%d given, %d blocks)", (int)inv.size(), nb);
        for (int i = 0; i < nb; i++)
            ELAG_REQUIRE(inv[(size_t)i] && inv[(size_t)i]->n() == B->row_off[(size_t)i + 1] - B->row_off[(size_t)i], "inverse of block %d has the wrong size", i);
    }
    for (ParCSR *blk : B->blk)
        if (blk) k::prepare_spmv(blk->diag);
    for (auto *s : inv) if (s->A) k::prepare_spmv(s->A->diag);
    const size_t n = (size_t)A->nrows();
    resid = dev_alloc<double>(n); corr = dev_alloc<double>(n); tmp = dev_alloc<double>(n); spare = dev_alloc<double>(n);
    if (kind == ELAG_HIP_BLOCK_JACOBI) {
        bool all_diag = true;
        for (auto *s : inv) if (!s->diag_scaling()) all_diag = false;
        if (all_diag) {
            // one sweep of diagonal relaxations in every block: the block solver is itself a diagonal scaling
            dscale = dev_alloc<double>(n);
            for (int i = 0; i < nb; i++) {
                const size_t ni = (size_t)(B->row_off[(size_t)i + 1] - B->row_off[(size_t)i]);
                if (ni)
                    dev_copy(dscale + B->row_off[(size_t)i], inv[(size_t)i]->diag_scaling(), sizeof(double) * ni, ctx().stream);
            }
        }
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
}

// out <- S(rhs) from the zero guess (the inner solvers run with iterative_mode == false: BlockDiagonalSolver.cpp:101)
void BlockSmoother::inner(elag_hip_smoother_s *S, const double *rhs, double *out, double *scratch, bool transpose)
{
    const int nf = S->flips(true);
    double *c = (nf & 1) ? scratch : out, *a = (nf & 1) ? out : scratch;
    S->apply(rhs, c, a, true, transpose);
    if (c != out) fail(ELAG_HIP_ERR_STATE, "internal: inner smoother left its result in the scratch vector");
}

void BlockSmoother::block_solve(const double *r, double *t, bool transpose)
{
    const int nb = B->nbr;
    const std::vector<int> &off = B->row_off;
    if (kind == ELAG_HIP_BLOCK_JACOBI) {
        for (int i = 0; i < nb; i++) inner(inv[(size_t)i], r + off[(size_t)i], t + off[(size_t)i], spare + off[(size_t)i], transpose);
        return;
    }
    if (kind == ELAG_HIP_BLOCK_LDU) {
        ELAG_REQUIRE(!transpose, "Block2x2LDUInverseOperator::MultTranspose is not implemented");   // as the reference (:140-146)
        const int n0 = off[1], n1 = off[2] - off[1];
        // r is the caller's right-hand side (zero guess) or the residual in resid; r_g' = r_g - A10 du~ goes to resid either way, and
        // resid's first block is free for the third application's result (no copy of the right-hand side first)
        const double *rf = r, *rg = r + n0;
        double *r0 = resid, *r1 = resid + n0;
        double *c0 = t, *c1 = t + n0;
        double *t0 = tmp, *s0 = spare, *s1 = spare + n0;
        static const bool reuse = !(getenv("ELAG_HIP_LDU_REUSE") && !atoi(getenv("ELAG_HIP_LDU_REUSE")));
        // An inverse that multiplies its result with -1 (the AMGe cycle on the positive B D^-1 B^T standing for the negative Schur complement)
        // is given the negated right-hand side instead -- the sign of the product that forms it -- and applied without its factor: the
        // same bits (smoother.hpp: result_scale()), one pass over the pressure vector less per application.
        const bool neg_s = fold && inv[3]->result_scale() == -1.0 && (inv[3]->plain_forms() & 1);
        if (fold && reuse && inv[0] == inv[1] && inv[2]->result_scale() == 1.0 && (inv[2]->plain_forms() & 2)) {
            // ... and where the third inverse can add its result in place, du~ is computed where du belongs, the product A01 dp is formed
            // with the other sign and  du = du~ + A_3^-1 (-A01 dp)  needs no subtraction of its own: the same bits again.
            inner(inv[1], rf, c0, s0, false);                          // du~ = A_2^-1 r_f, in du's place
            { k::SpmvArgs a; a.x = c0; a.alpha = neg_s ? 1.0 : -1.0; a.f = rg; a.gamma = neg_s ? -1.0 : 1.0; a.out = r1; par_spmv(*B->block(1, 0), a); }
            if (neg_s) inv[3]->apply_plain(r1, c1, false); else inner(inv[3], r1, c1, s1, false);
            { k::SpmvArgs a; a.x = c1; a.alpha = -1.0; a.out = t0; par_spmv(*B->block(0, 1), a); }                         // -A01 dp
            inv[2]->apply_plain(t0, c0, true);                          // du = du~ + A_3^-1 (-A01 dp)
            if (damping != 1.0) k::vec_axpby(n0 + n1, damping, t, 0.0, t, t);
            return;
        }
        inner(inv[1], rf, t0, s0, false);                              // du~ = A_2^-1 r_f
        { k::SpmvArgs a; a.x = t0; a.alpha = neg_s ? 1.0 : -1.0; a.f = rg; a.gamma = neg_s ? -1.0 : 1.0; a.out = r1; par_spmv(*B->block(1, 0), a); }   // r_g' = r_g - A10 du~ (or its negative)
        if (neg_s) inv[3]->apply_plain(r1, c1, false); else inner(inv[3], r1, c1, s1, false);                              // dp = S^-1 (...)
        { k::SpmvArgs a; a.x = c1; a.alpha = 1.0; a.out = c0; par_spmv(*B->block(0, 1), a); }                              // A01 dp
        if (reuse && inv[0] == inv[1]) {
            // A_1^-1 and A_2^-1 are ONE object (the reference's own lists give one A00 inverse to all three places) applied to the same r_f from
            // the zero guess: du^ is du~, bit for bit -- kept in t0, and the third application writes into r_f's place, which nobody reads again.
            // One of the four inner cycles of every application less.
            inner(inv[2], c0, r0, s0, false);                          // A_3^-1 A01 dp
            k::vec_axpby(n0, 1.0, t0, -1.0, r0, c0);                   // du = du^ - A_3^-1 A01 dp
        } else {
            inner(inv[2], c0, t0, s0, false);                          // A_3^-1 A01 dp
            inner(inv[0], rf, c0, s0, false);                          // du^ = A_1^-1 r_f
            k::vec_axpby(n0, 1.0, c0, -1.0, t0, c0);                   // du = du^ - A_3^-1 A01 dp
        }
        if (damping != 1.0) k::vec_axpby(n0 + n1, damping, t, 0.0, t, t);
        return;
    }
    // block triangular: lower counts up with A_ij (j < i); its transpose counts down with A_ji^T (j > i); upper mirrored
    const bool lower = (kind == ELAG_HIP_BLOCK_GS_LOWER);
    const bool ascending = (lower != transpose);
    for (int step = 0; step < nb; step++) {
        const int i = ascending ? step : nb - 1 - step;
        const int ni = off[(size_t)i + 1] - off[(size_t)i];
        const double *ri = r + off[(size_t)i];
        double *ti = tmp + off[(size_t)i];
        bool first = true;
        for (int j = ascending ? 0 : i + 1; j < (ascending ? i : nb); j++) {
            if (!transpose) {
                ParCSR *Aij = B->block(i, j);
                if (!Aij) continue;
                k::SpmvArgs a;
                a.x = t + off[(size_t)j]; a.alpha = -1.0; a.out = ti;
                if (first) { a.f = ri; a.gamma = 1.0; } else { a.yin = ti; a.beta = 1.0; }
                par_spmv(*Aij, a);
            } else {
                ParCSR *Aji = B->block(j, i);
                if (!Aji) continue;
                if (first && ni) dev_copy(ti, ri, sizeof(double) * (size_t)ni, ctx().stream);
                par_spmvT(*Aji, -1.0, t + off[(size_t)j], 1.0, ti);
            }
            first = false;
        }
        inner(inv[(size_t)i], first ? ri : ti, t + off[(size_t)i], spare + off[(size_t)i], transpose);
    }
}

void BlockSmoother::apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose)
{
    const int nn = n();
    if (dscale) {
        if (zero_guess) { k::vec_scale_mul(nn, 1.0, f, dscale, cur); return; }
        k::SpmvArgs a;   // sol + W (rhs - A sol) in one pass, out of place
        a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.s = dscale; a.yin = cur; a.beta = 1.0; a.out = alt;
        par_spmv(*A, a);
        std::swap(cur, alt);
        return;
    }
    const double *r = f;
    if (!zero_guess) {
        k::SpmvArgs a;   // Resid = rhs - A sol (BlockDiagonalSolver.cpp:84-91)
        a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.out = resid;
        par_spmv(*A, a);
        r = resid;
    }
    double *t = zero_guess ? cur : corr;
    block_solve(r, t, transpose);
    if (!zero_guess) k::vec_axpby(nn, 1.0, cur, 1.0, corr, cur);   // sol += correction
}

}  // namespace elag

// ---- C ABI --------------------------------------------------------------------------------------------------------------
extern "C" {

int elag_hip_blockop_create(elag_hip_blockop_t *out, int num_block_rows, int num_block_cols, const int *row_offsets, const int *col_offsets)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && num_block_rows > 0 && num_block_cols > 0 && row_offsets && col_offsets, "bad argument");
        std::unique_ptr<elag_hip_blockop_s> B(new elag_hip_blockop_s());
        B->nbr = num_block_rows; B->nbc = num_block_cols;
        B->row_off.assign(row_offsets, row_offsets + num_block_rows + 1);
        B->col_off.assign(col_offsets, col_offsets + num_block_cols + 1);
        ELAG_REQUIRE(B->row_off[0] == 0 && B->col_off[0] == 0, "block offsets start at 0");
        for (int i = 0; i < num_block_rows; i++) ELAG_REQUIRE(B->row_off[(size_t)i + 1] >= B->row_off[(size_t)i], "row offsets must not decrease");
        for (int j = 0; j < num_block_cols; j++) ELAG_REQUIRE(B->col_off[(size_t)j + 1] >= B->col_off[(size_t)j], "column offsets must not decrease");
        B->blk.assign((size_t)num_block_rows * num_block_cols, nullptr);
        B->owned.assign(B->blk.size(), 0);
        *out = B.release();
    });
}

int elag_hip_blockop_set_block(elag_hip_blockop_t B, int block_row, int block_col, elag_hip_parcsr_t A)
{
    return guarded([&] { ELAG_REQUIRE(B, "null block operator"); B->set_block(block_row, block_col, A, false); });
}

int elag_hip_blockop_get_block(elag_hip_blockop_t B, int block_row, int block_col, elag_hip_parcsr_t *A)
{
    return guarded([&] {
        ELAG_REQUIRE(B && A, "null argument");
        ELAG_REQUIRE(block_row >= 0 && block_row < B->nbr && block_col >= 0 && block_col < B->nbc, "block index out of range");
        *A = B->block(block_row, block_col);   // NULL <=> IsZeroBlock
    });
}

int elag_hip_blockop_monolithic(elag_hip_blockop_t B, elag_hip_parcsr_t *A)
{
    return guarded([&] { require_init(); ELAG_REQUIRE(B && A, "null argument"); *A = B->monolithic(); });
}

int elag_hip_blockop_mult(elag_hip_blockop_t B, int trans, double alpha, const double *x, double beta, double *y, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(B, "null block operator");
        check_rc(elag_hip_parcsr_spmv(B->monolithic(), trans, alpha, x, beta, y, loc));
    });
}

int elag_hip_blockop_destroy(elag_hip_blockop_t B)
{
    return guarded([&] {
        if (B) { (void)hipStreamSynchronize(ctx().stream); delete B; }
    });
}

int elag_hip_schur_complement(elag_hip_blockop_t B, int s_type, double alpha, int negate, elag_hip_parcsr_t *S)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(B && S, "null argument");
        *S = schur_complement(*B, s_type, alpha, negate != 0);
    });
}

int elag_hip_block_smoother_create(elag_hip_smoother_t *out, elag_hip_blockop_t B, int kind, int num_inverses, const elag_hip_smoother_t *inverses,
                                   double damping)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && B && inverses && num_inverses > 0, "bad argument");
        std::unique_ptr<BlockSmoother> S(new BlockSmoother());
        S->B = B; S->kind = kind; S->damping = damping;
        S->inv.assign(inverses, inverses + num_inverses);
        for (auto *s : S->inv) ELAG_REQUIRE(s, "null inverse");
        S->setup();
        *out = S.release();
    });
}

int elag_hip_blocked_default_options(elag_hip_blocked_options *opt)
{
    return guarded([&] {
        ELAG_REQUIRE(opt, "null options");
        opt->block_smoother = ELAG_HIP_BLOCK_JACOBI;
        check_rc(elag_hip_amge_default_options(&opt->inv_a00));
        check_rc(elag_hip_amge_default_options(&opt->inv_a11));
        opt->s_type = ELAG_HIP_SCHUR_DIAGONAL; opt->alpha = 1.0; opt->use_negative_s = 1; opt->damping = 1.0;
        opt->coarse_kind = 1; opt->use_graph = 1;
    });
}

int elag_hip_hierarchy_build_blocked(elag_hip_hierarchy_t *out, elag_hip_blockop_t A, int num_P, const elag_hip_parcsr_t *P,
                                     const int *fix_zero_rows, const elag_hip_blocked_options *opt_in)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && A && num_P >= 0 && (num_P == 0 || P), "bad argument");
        // the checks of buildBlockedHierarchyFromDeRhamSequence (Hierarchy.cpp:411-436)
        ELAG_REQUIRE(A->nbr == A->nbc, "buildBlockedHierarchyFromDeRhamSequence(...): A is not block-square!");
        const int nb = A->nbr;
        elag_hip_blocked_options opt;
        if (opt_in) opt = *opt_in; else check_rc(elag_hip_blocked_default_options(&opt));
        const bool needs_s = opt.s_type != ELAG_HIP_SCHUR_NONE;
        if (needs_s || opt.block_smoother == ELAG_HIP_BLOCK_LDU) ELAG_REQUIRE(nb == 2, "Schur-complement block smoothers need a 2x2 operator");

        elag_hip_hierarchy_t H = nullptr;
        check_rc(elag_hip_hierarchy_create(&H, num_P + 1));
        struct Guard { elag_hip_hierarchy_t h; ~Guard() { if (h) (void)elag_hip_hierarchy_destroy(h); } } guard{H};
        auto adopt_op = [&](elag_hip_blockop_s *b) { hierarchy_adopt(H, [b] { delete b; }); };
        auto adopt_mat = [&](ParCSR *m) { hierarchy_adopt(H, [m] { delete m; }); };
        auto adopt_sm = [&](elag_hip_smoother_s *s) { hierarchy_adopt(H, [s] { delete s; }); };

        std::vector<elag_hip_blockop_s *> ops((size_t)num_P + 1, nullptr);
        ops[0] = A;
        for (int l = 0; l < num_P; l++) {
            elag_hip_blockop_s *Af = ops[(size_t)l];
            const elag_hip_parcsr_t *Pl = P + (size_t)l * nb;
            std::vector<int> coff((size_t)nb + 1, 0);
            for (int i = 0; i < nb; i++) {
                ELAG_REQUIRE(Pl[i], "null P block %d on level %d", i, l);
                ELAG_REQUIRE(Pl[i]->nrows() == Af->row_off[(size_t)i + 1] - Af->row_off[(size_t)i],
                             "buildBlockedHierarchy: P[%d] block %d has %d rows but the operator block has %d", l, i, Pl[i]->nrows(),
                             Af->row_off[(size_t)i + 1] - Af->row_off[(size_t)i]);
                coff[(size_t)i + 1] = coff[(size_t)i] + Pl[i]->ncols();
            }
            elag_hip_blockop_t Ac = nullptr, Pb = nullptr;
            check_rc(elag_hip_blockop_create(&Ac, nb, nb, coff.data(), coff.data()));
            adopt_op(Ac);
            check_rc(elag_hip_blockop_create(&Pb, nb, nb, Af->row_off.data(), coff.data()));
            adopt_op(Pb);
            hierarchy_phase(H, "galerkin_products", [&] {
                for (int i = 0; i < nb; i++) {
                    for (int j = 0; j < nb; j++) {
                        ParCSR *Aij = Af->block(i, j);
                        if (!Aij) continue;                          // zero blocks stay zero (Hierarchy.cpp:503)
                        elag_hip_parcsr_t C = nullptr;
                        check_rc(elag_hip_parcsr_rap(Pl[i], Aij, Pl[j], &C));
                        Ac->set_block(i, j, C, true);
                        if (i == j && fix_zero_rows && fix_zero_rows[i]) k::fix_zero_rows(C->diag, &C->offd, false);   // :515-519
                    }
                    Pb->set_block(i, i, Pl[i], false);
                }
            });
            ops[(size_t)l + 1] = Ac;
            hierarchy_phase(H, "merged_operators", [&] { check_rc(elag_hip_hierarchy_set_prolongator(H, l + 1, Pb->monolithic())); });
        }
        // the coarsest level's direct solve of the (indefinite) merged operator: its dense LU inverse -- 5.9 ms of launches for the 1 471 dofs of
        // cube456 x3 -- starts as soon as that operator is merged and runs on the second stream beside the other levels' merges, the Schur
        // complements, the smoothers and the fused level operators of Finalize, which waits for it where the dense tail is planned
        elag_hip_coarse_t C = nullptr;
        ParCSR *Ac_merged = nullptr;
        hierarchy_phase(H, "merged_operators", [&] { Ac_merged = ops[(size_t)num_P]->monolithic(); });
        if (!is_distributed(*Ac_merged)) hierarchy_phase(H, "coarse_solver", [&] { C = coarse_create_beside(Ac_merged, opt.coarse_kind); });
        hierarchy_phase(H, "merged_operators", [&] {
            for (int l = 0; l <= num_P; l++) check_rc(elag_hip_hierarchy_set_operator(H, l, ops[(size_t)l]->monolithic()));
        });

        auto relax = [&](ParCSR *M, const elag_hip_amge_options &o) {
            elag_hip_smoother_t S = nullptr;
            check_rc(elag_hip_smoother_create(&S, M, o.relax_type, o.sweeps, o.damping, 1.0, o.cheby_order, o.cheby_fraction));
            adopt_sm(S);
            return S;
        };
        for (int l = 0; l < num_P; l++) {
            elag_hip_blockop_s *Bl = ops[(size_t)l];
            std::vector<elag_hip_smoother_t> inv;
            if (nb == 2 && (needs_s || opt.block_smoother == ELAG_HIP_BLOCK_LDU)) {
                ELAG_REQUIRE(Bl->block(0, 0), "block smoother: A00 is a zero block on level %d", l);
                ParCSR *A11 = Bl->block(1, 1);
                if (needs_s) {
                    hierarchy_phase(H, "schur_complements", [&] { A11 = schur_complement(*Bl, opt.s_type, opt.alpha, opt.use_negative_s != 0); });
                    adopt_mat(A11);
                }
                ELAG_REQUIRE(A11, "block smoother: A11 is a zero block and no Schur complement was requested (level %d)", l);
                elag_hip_smoother_t s00 = nullptr, s11 = nullptr;
                hierarchy_phase(H, "smoothers", [&] { s00 = relax(Bl->block(0, 0), opt.inv_a00); s11 = relax(A11, opt.inv_a11); });
                if (opt.block_smoother == ELAG_HIP_BLOCK_LDU) inv = {s00, s00, s00, s11};
                else inv = {s00, s11};
            } else {
                for (int i = 0; i < nb; i++) {
                    ELAG_REQUIRE(Bl->block(i, i), "block smoother: diagonal block %d is zero on level %d", i, l);
                    inv.push_back(relax(Bl->block(i, i), i == 0 ? opt.inv_a00 : opt.inv_a11));
                }
            }
            elag_hip_smoother_t S = nullptr;
            hierarchy_phase(H, "smoothers", [&] { check_rc(elag_hip_block_smoother_create(&S, Bl, opt.block_smoother, (int)inv.size(), inv.data(), opt.damping)); });
            adopt_sm(S);
            check_rc(elag_hip_hierarchy_set_smoothers(H, l, S, S));
        }
        // coarsest level: direct solve of the (indefinite) merged operator
        ParCSR *Ac = ops[(size_t)num_P]->monolithic();
        if (is_distributed(*Ac)) {
            std::vector<int64_t> starts;
            ParCSR *Arep = replicate_matrix(*Ac, &starts);
            adopt_mat(Arep);
            check_rc(elag_hip_coarse_create_replicated(&C, Arep, starts.data(), opt.coarse_kind));
        }
        hierarchy_adopt(H, [C] { delete C; });
        check_rc(elag_hip_hierarchy_set_coarse_solver(H, C));
        check_rc(elag_hip_hierarchy_finalize(H, opt.use_graph));
        guard.h = nullptr;
        *out = H;
    });
}

}  // extern "C"
