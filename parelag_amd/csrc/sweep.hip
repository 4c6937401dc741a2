// Sequential-per-rank relaxations of hypre_ParCSRRelax on the device: relax types 2 / 4 ("L1 Gauss-Seidel" -- the type every
// shipped parameter list of the reference selects, examples/testing_helpers/Create0FormParameterList.hpp:103-116 -- and its
// truncated variant) and 6 ("Gauss-Seidel", hypre_BoomerAMGRelax's hybrid symmetric sweep); type table
// src/linalg/factories/ParELAG_HypreSmootherFactory.cpp:92-109.
//
// A Gauss-Seidel sweep over the local rows 0..n-1 is a chain only along the edges of the matrix graph.  With
//     level(i) = 1 + max { level(j) : j < i, a_ij != 0 or a_ji != 0 }          (0 without such a j)
// the rows of one level are mutually independent, row i reads every neighbour j < i after its update (lower level) and every
// neighbour j > i before it (higher level) -- exactly what the sequential loop does -- so running the levels in ascending
// order IS the forward sweep and in descending order IS the backward sweep, value for value: each row is summed by one lane in
// storage order, so the results carry the CPU loop's bits.  Ghost values are exchanged once per symmetric sweep and stay frozen,
// as in hypre.
//
// Launch plan ("segments"): a level with more than kChainRows rows is one multi-workgroup launch; a run of consecutive smaller
// levels is ONE launch of a single workgroup that walks the levels with a workgroup barrier in between (the waves of a
// workgroup share their CU's L1, so values written before the barrier are read after it without leaving the CU).  All
// launches sit in the V-cycle's hipGraph like any other kernel of the cycle.  A sweep on a small operator is a latency chain
// (one level = one dependent round trip through memory), so the kernel is built to make that trip short: a 16-byte schedule
// slot per row (row, first entry, length), kLPR lanes fetching a row's entries and their u values together, kUnroll rows in
// flight per lane group; only the subtraction itself is sequential, done from shuffled products in storage order.
#include <algorithm>
#include <numeric>

#include "comm.hpp"
#include "smoother.hpp"

namespace elag {

namespace {

constexpr int kLPR = 16;           // lanes that fetch the entries of one row together (the row sum itself stays sequential)
constexpr int kUnroll = 4;         // rows a lane group has in flight at once
constexpr int kChainBlock = 1024;
constexpr int kChainRows = (kChainBlock / kLPR) * kUnroll;   // levels up to this many rows are walked by one workgroup: 256
constexpr int kWideBlock = 256;
constexpr int kMaxChain = 2048;    // levels per chain launch (their offsets sit in LDS)

typedef int int4s __attribute__((ext_vector_type(4)));

// One row, kLPR lanes: the lanes fetch a_ik, u_k of up to kLPR entries at once; the subtraction is then done in storage order
// (every lane does it, from shuffled products), so the value is the sequential loop's.
// MODE 0 (types 2 / 4):  u_i += (f_i - sum_k a_ik u_k - sum_g o_ig ughost_g) / l1_i
// MODE 1 (type 6):       u_i  = (f_i - sum_{k != i} a_ik u_k - sum_g o_ig ughost_g) / a_ii      rows with a_ii == 0 (or none) skipped
template <int MODE>
__device__ __forceinline__ void ordered_chunk(int lane, int cnt, int row, int j, double v, double x, double &res, double &aii, bool &have)
{
    const double p = v * x;
    for (int q = 0; q < cnt; q++) {
        const double pq = __shfl(p, q, kLPR);
        if (MODE == 1) {
            const int jq = __shfl(j, q, kLPR);
            const double vq = __shfl(v, q, kLPR);
            if (jq == row && !have) { aii = vq; have = true; continue; }
        }
        res -= pq;
    }
    (void)lane;
}

template <int MODE>
__global__ void sweep_levels_kernel(const int *__restrict__ lev_ptr, int l_first, int l_count, int dir, const int4s *__restrict__ sched,
                                    const int *__restrict__ J, const double *__restrict__ a,
                                    const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa,
                                    const double *__restrict__ f, const double *__restrict__ d, const double *__restrict__ ug, double *u)
{
    __shared__ int lp[kMaxChain + 1];
    // offsets of this launch's levels, lowest level first
    const int l_lo = dir > 0 ? l_first : l_first - (l_count - 1);
    for (int c = threadIdx.x; c <= l_count; c += blockDim.x) lp[c] = lev_ptr[l_lo + c];
    __syncthreads();
    const int lane = threadIdx.x & (kLPR - 1);
    const int grp = (int)(blockIdx.x * blockDim.x + threadIdx.x) / kLPR;
    const int ngrp = (int)(gridDim.x * blockDim.x) / kLPR;
    // l_count > 1 only with a grid of ONE workgroup (launch_segment): the barrier below orders the levels
    for (int c = 0; c < l_count; c++) {
        const int li = dir > 0 ? c : l_count - 1 - c;
        const int b = lp[li], e = lp[li + 1];
        for (int t0 = b + grp; t0 < e; t0 += kUnroll * ngrp) {
            int4s sl[kUnroll];
            int j0[kUnroll];
            double v0[kUnroll], x0[kUnroll], fr[kUnroll], dr[kUnroll], uo[kUnroll];
#pragma unroll
            for (int q = 0; q < kUnroll; q++) {
                const int t = t0 + q * ngrp;
                sl[q] = t < e ? sched[t] : int4s{0, 0, 0, 0};
            }
#pragma unroll
            for (int q = 0; q < kUnroll; q++) {
                const bool has = lane < sl[q].z;
                j0[q] = has ? J[sl[q].y + lane] : 0;
                v0[q] = has ? a[sl[q].y + lane] : 0.0;
                fr[q] = f[sl[q].x];
                dr[q] = (MODE == 0) ? d[sl[q].x] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kUnroll; q++) {
                x0[q] = (lane < sl[q].z) ? u[j0[q]] : 0.0;
                uo[q] = (MODE == 0) ? u[sl[q].x] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kUnroll; q++) {
                const int t = t0 + q * ngrp;
                if (t >= e) continue;
                const int row = sl[q].x, s = sl[q].y, len = sl[q].z;
                double res = fr[q], aii = 0.0;
                bool have = false;
                ordered_chunk<MODE>(lane, min(len, kLPR), row, j0[q], v0[q], x0[q], res, aii, have);
                for (int k0 = kLPR; k0 < len; k0 += kLPR) {          // long rows: further chunks, one after the other
                    const bool has = k0 + lane < len;
                    const int j = has ? J[s + k0 + lane] : 0;
                    const double v = has ? a[s + k0 + lane] : 0.0;
                    const double x = has ? u[j] : 0.0;
                    ordered_chunk<MODE>(lane, min(len - k0, kLPR), row, j, v, x, res, aii, have);
                }
                if (oI) {
                    const int os = oI[row], ol = oI[row + 1] - os;
                    for (int k0 = 0; k0 < ol; k0 += kLPR) {
                        const bool has = k0 + lane < ol;
                        const double v = has ? oa[os + k0 + lane] : 0.0;
                        const double x = has ? ug[oJ[os + k0 + lane]] : 0.0;
                        const double p = v * x;
                        const int cnt = min(ol - k0, kLPR);
                        for (int r = 0; r < cnt; r++) res -= __shfl(p, r, kLPR);
                    }
                }
                if (lane == 0) {
                    if (MODE == 0) { if (dr[q] != 0.0) u[row] = uo[q] + res / dr[q]; }
                    else if (have && aii != 0.0) u[row] = res / aii;
                }
            }
        }
        if (l_count > 1) __syncthreads();
    }
}

// hypre_ParCSRComputeL1Norms options 2 / 4 with the CPU loop's order of additions (one lane per row, storage order), so that
// a sweep divides by the very same number: |a_ii| (+ |offd| terms; option 4: halved and dropped when small), negative where
// the first diagonal entry is negative.
__global__ void l1_norms_ordered_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                        const int *__restrict__ oI, const double *__restrict__ oa, int option, double *__restrict__ l1,
                                        int *__restrict__ bad)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double d = 0.0, o = 0.0, r;
        bool neg = false, first = true;
        for (int k = I[i]; k < I[i + 1]; k++)
            if (J[k] == i) { d += fabs(a[k]); if (first) { neg = a[k] < 0.0; first = false; } }
        if (option == 2) {
            r = d;
            if (oI) for (int k = oI[i]; k < oI[i + 1]; k++) r += fabs(oa[k]);
        } else {
            if (oI) for (int k = oI[i]; k < oI[i + 1]; k++) o += 0.5 * fabs(oa[k]);
            r = d + o;
            if (r <= 4.0 / 3.0 * d) r = d;
        }
        if (neg) r = -r;
        if (fabs(r) < 2.220446049250313e-16) atomicAdd(bad, 1);
        l1[i] = r;
    }
}

__global__ void diag_scale_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                  const double *__restrict__ y, double *__restrict__ x)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double aii = 0.0;
        for (int k = I[i]; k < I[i + 1]; k++)
            if (J[k] == i) { aii = a[k]; break; }
        x[i] = y[i] / aii;
    }
}

}  // namespace

SweepSmoother::~SweepSmoother() { dev_free(d_rows); dev_free(d_lev_ptr); dev_free(d_l1); }

void SweepSmoother::setup()
{
    ELAG_REQUIRE(type == ELAG_HIP_SMOOTHER_L1_GS || type == ELAG_HIP_SMOOTHER_L1_GS_TRUNCATED || type == ELAG_HIP_SMOOTHER_GS,
                 "internal: relax type %d is not a sweep", type);
    // hypre_ParCSRRelax 2/4 and hypre_BoomerAMGRelax 6 branch on (relax_weight, omega): (1, 1) is the symmetric Gauss-Seidel of
    // every shipped parameter list ("Damping Factor" 1.0, "Omega" 1.0); the SSOR branches are not built
    if (weight != 1.0 || omega != 1.0)
        fail(ELAG_HIP_ERR_ARG, "relax type %d: only \"Damping Factor\" = \"Omega\" = 1.0 (symmetric Gauss-Seidel) is available, got %g / %g",
             type, weight, omega);
    const int n = A->nrows();
    hipStream_t st = ctx().stream;
    if (type != ELAG_HIP_SMOOTHER_GS) {
        // mfem::HypreSmoother::SetOperator: l1_norms = hypre_ParCSRComputeL1Norms(A, type) for types 1..4
        Temp<int> bad(1);
        ELAG_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), st));
        d_l1 = dev_alloc<double>((size_t)std::max(n, 1));
        if (n) {
            const bool ho = A->offd.nnz > 0;
            const unsigned g = (unsigned)std::min<long long>(((long long)n + 255) / 256, 2048);
            hipLaunchKernelGGL(l1_norms_ordered_kernel, dim3(g), dim3(256), 0, st, n, (const int *)A->diag.I, (const int *)A->diag.J,
                               (const double *)A->diag.a, ho ? (const int *)A->offd.I : nullptr, ho ? (const double *)A->offd.a : nullptr, type,
                               d_l1, bad.p);
            ELAG_LAUNCH_CHECK();
        }
        int nbad = 0;
        ELAG_HIP_CHECK(hipMemcpyAsync(&nbad, bad.p, sizeof(int), hipMemcpyDeviceToHost, st));
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        if (nbad) fail(ELAG_HIP_ERR_NUMERIC, "l1 Gauss-Seidel: matrix has %d zero l1 row norm(s)", nbad);
    }
    // ---- dependency levels of the symmetrised pattern, one pass over the rows ----
    std::vector<int> I = host_from_dev(A->diag.I, (size_t)n + 1);
    std::vector<int> J = host_from_dev(A->diag.J, (size_t)A->diag.nnz);
    std::vector<int> level((size_t)n, 0);     // doubles as "pending" for rows not reached yet
    int nlev = n ? 1 : 0;
    for (int i = 0; i < n; i++) {
        int lv = level[(size_t)i];
        for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) {
            const int j = J[(size_t)k];
            if (j < i) lv = std::max(lv, level[(size_t)j] + 1);
        }
        level[(size_t)i] = lv;
        nlev = std::max(nlev, lv + 1);
        for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) {
            const int j = J[(size_t)k];
            if (j > i && j < n) level[(size_t)j] = std::max(level[(size_t)j], lv + 1);
        }
    }
    std::vector<int> ptr((size_t)nlev + 1, 0), rows((size_t)n);
    for (int i = 0; i < n; i++) ptr[(size_t)level[(size_t)i] + 1]++;
    std::partial_sum(ptr.begin(), ptr.end(), ptr.begin());
    {
        std::vector<int> cur(ptr.begin(), ptr.end() - 1);
        for (int i = 0; i < n; i++) rows[(size_t)cur[(size_t)level[(size_t)i]]++] = i;   // ascending row inside a level
    }
    num_levels = nlev;
    segments.clear();
    for (int l = 0; l < nlev;) {
        const int sz = ptr[(size_t)l + 1] - ptr[(size_t)l];
        if (sz > kChainRows) { segments.push_back(Segment{l, 1, sz}); l++; continue; }
        int m = l, mx = 0;
        while (m < nlev && m - l < kMaxChain && ptr[(size_t)m + 1] - ptr[(size_t)m] <= kChainRows) {
            mx = std::max(mx, ptr[(size_t)m + 1] - ptr[(size_t)m]);
            m++;
        }
        segments.push_back(Segment{l, m - l, mx});
        l = m;
    }
    // schedule slots: (row, first entry, row length) in level order -- one 16-byte load per row instead of three dependent ones
    std::vector<int> slot((size_t)std::max(n, 1) * 4, 0);
    for (int t = 0; t < n; t++) {
        const int i = rows[(size_t)t];
        slot[(size_t)t * 4] = i; slot[(size_t)t * 4 + 1] = I[(size_t)i]; slot[(size_t)t * 4 + 2] = I[(size_t)i + 1] - I[(size_t)i];
    }
    d_rows = dev_from_host(slot.data(), slot.size());
    d_lev_ptr = dev_from_host(ptr.data(), ptr.size());
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    if (getenv("ELAG_HIP_SWEEP_VERBOSE"))
        fprintf(stderr, "[elag_hip] sweep smoother type %d: n=%d, %d dependency levels in %d launches per direction\n", type, n, nlev,
                (int)segments.size());
}

void SweepSmoother::launch_segment(const Segment &s, int dir, const double *f, double *u)
{
    hipStream_t st = ctx().stream;
    const bool ho = A->offd.nnz > 0;
    const int *oI = ho ? A->offd.I : nullptr, *oJ = ho ? A->offd.J : nullptr;
    const double *oa = ho ? A->offd.a : nullptr;
    const int l_first = dir > 0 ? s.first : s.first + s.count - 1;
    unsigned grid = 1, block = kChainBlock;
    if (s.count == 1 && s.max_rows > kChainRows) {
        // one big level: enough workgroups for one row per lane group
        block = kWideBlock;
        grid = (unsigned)(((long long)s.max_rows * kLPR + kWideBlock - 1) / kWideBlock);
    } else {
        // chain: as many lane groups as the largest level needs with kUnroll rows each, in whole waves
        const int groups = (s.max_rows + kUnroll - 1) / kUnroll;
        block = (unsigned)std::min(kChainBlock, std::max(64, ((groups * kLPR + 63) / 64) * 64));
    }
    const int4s *sched = reinterpret_cast<const int4s *>(d_rows);
    if (type == ELAG_HIP_SMOOTHER_GS)
        hipLaunchKernelGGL(sweep_levels_kernel<1>, dim3(grid), dim3(block), 0, st, (const int *)d_lev_ptr, l_first, s.count, dir, sched,
                           (const int *)A->diag.J, (const double *)A->diag.a, oI, oJ, oa, f, (const double *)nullptr,
                           (const double *)A->d_ghost, u);
    else
        hipLaunchKernelGGL(sweep_levels_kernel<0>, dim3(grid), dim3(block), 0, st, (const int *)d_lev_ptr, l_first, s.count, dir, sched,
                           (const int *)A->diag.J, (const double *)A->diag.a, oI, oJ, oa, f, (const double *)d_l1,
                           (const double *)A->d_ghost, u);
    ELAG_LAUNCH_CHECK();
}

void SweepSmoother::apply(const double *f, double *&cur, double *&, bool zero_guess, bool /*transpose: the symmetric sweep is self-adjoint*/)
{
    const int n = A->nrows();
    if (zero_guess) k::vec_fill(n, 0.0, cur);
    for (int sw = 0; sw < sweeps; sw++) {
        // "Copy off-diag values of u to the current processor": once per symmetric sweep
        if (A->needs_exchange()) par_halo_exchange(*A, cur);
        for (size_t s = 0; s < segments.size(); s++) launch_segment(segments[s], +1, f, cur);
        for (size_t s = segments.size(); s-- > 0;) launch_segment(segments[s], -1, f, cur);
    }
}

// mfem::HypreDiagScale ("Hypre Jacobi" = 413, ParELAG_HypreSmootherFactory.cpp:28-31): HYPRE_ParCSRDiagScale, x = b ./ diag(A)
// whatever x held.
void DiagScaleSmoother::apply(const double *f, double *&cur, double *&, bool, bool)
{
    const int n = A->nrows();
    if (n == 0) return;
    const unsigned g = (unsigned)std::min<long long>(((long long)n + 255) / 256, 2048);
    hipLaunchKernelGGL(diag_scale_kernel, dim3(g), dim3(256), 0, ctx().stream, n, (const int *)A->diag.I, (const int *)A->diag.J,
                       (const double *)A->diag.a, f, cur);
    ELAG_LAUNCH_CHECK();
}

}  // namespace elag
