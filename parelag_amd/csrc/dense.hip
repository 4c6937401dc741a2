// Dense inverse of a small general (indefinite) operator on the device: Gauss-Jordan with partial pivoting.  Used for the
// coarsest saddle-point operator of the blocked (Darcy) hierarchy, where the reference's XML names a "Direct" solver (SuperLU;
// src/linalg/factories/ParELAG_DirectSolverFactory.cpp) or a short GMRES; the SPD single-form hierarchies keep the Cholesky /
// unpivoted Gauss-Jordan paths further down.
//
// Default: the blocked in-place form (gj_block_*_kernel, 8 pivots per step, two launches per step; up to 3 072 dofs).  Fallback and
// A/B reference (ELAG_HIP_LU_BLOCKED=0): one column at a time on the augmented matrix [A | I], two launches per pivot
// (gj_column_kernel + gj_eliminate_kernel) or four (ELAG_HIP_LU_FUSED_COLUMN=0).
#include "kernels.hpp"

namespace elag {
namespace k {

namespace {

constexpr int kB = 256;

__global__ void dense_scatter_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                     double *__restrict__ M, int ld)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int k = I[r]; k < I[r + 1]; k++) M[(size_t)r * ld + J[k]] += a[k];   // duplicates (if any) add up, one thread per row
    M[(size_t)r * ld + n + r] = 1.0;
}

__global__ void dense_scatter_plain_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a, double *__restrict__ M)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int k = I[r]; k < I[r + 1]; k++) M[(size_t)r * n + J[k]] += a[k];
}

// argmax_i>=k |M[i][k]|, lowest index on ties (deterministic)
__global__ void __launch_bounds__(1024) gj_pivot_kernel(int n, int k, const double *__restrict__ M, int ld, int *__restrict__ piv,
                                                        double *__restrict__ pivval)
{
    __shared__ double sv[1024];
    __shared__ int si[1024];
    double best = -1.0;
    int bi = n;
    for (int i = k + (int)threadIdx.x; i < n; i += 1024) {
        const double v = fabs(M[(size_t)i * ld + k]);
        if (v > best) { best = v; bi = i; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int p = si[0] < n ? si[0] : k;
        piv[0] = p;
        pivval[0] = M[(size_t)p * ld + k];
    }
}

// pivot search, row swap + scaling and the saved column in ONE single-workgroup launch (three launches before: 1 471 columns of the Darcy
// hierarchy's coarsest saddle-point operator were 5 900 launches, 27 of the 34 ms of that hierarchy's setup); the elimination keeps the grid
__global__ void __launch_bounds__(1024) gj_column_kernel(int n, int k, double *__restrict__ M, int ld, double *__restrict__ colk, int *__restrict__ singular)
{
    __shared__ double sv[1024];
    __shared__ int si[1024];
    __shared__ int s_p;
    __shared__ double s_pv;
    double best = -1.0;
    int bi = n;
    for (int i = k + (int)threadIdx.x; i < n; i += 1024) {
        const double v = fabs(M[(size_t)i * ld + k]);
        if (v > best) { best = v; bi = i; }
    }
    sv[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const double ov = sv[threadIdx.x + s];
            const int oi = si[threadIdx.x + s];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int p = si[0] < n ? si[0] : k;
        s_p = p;
        s_pv = M[(size_t)p * ld + k];
    }
    __syncthreads();
    const int p = s_p;
    const double pv = s_pv;
    if (pv == 0.0 || pv != pv) {
        if (threadIdx.x == 0) atomicAdd(singular, 1);
        for (int i = threadIdx.x; i < n; i += 1024) colk[i] = 0.0;          // (the elimination that follows is a no-op)
        return;
    }
    for (int c = threadIdx.x; c < ld; c += 1024) {
        const double tk = M[(size_t)k * ld + c], tp = M[(size_t)p * ld + c];
        if (p != k) M[(size_t)p * ld + c] = tk;
        M[(size_t)k * ld + c] = tp / pv;
    }
    __syncthreads();                                                           // (the workgroup's own global writes are visible to it after the barrier)
    for (int i = threadIdx.x; i < n; i += 1024) colk[i] = (i == k) ? 0.0 : M[(size_t)i * ld + k];
}

__global__ void gj_swap_scale_kernel(int k, double *__restrict__ M, int ld, const int *__restrict__ piv, const double *__restrict__ pivval,
                                     int *__restrict__ singular)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ld) return;
    const int p = piv[0];
    const double pv = pivval[0];
    if (pv == 0.0 || pv != pv) { if (c == 0) atomicAdd(singular, 1); return; }
    const double tk = M[(size_t)k * ld + c], tp = M[(size_t)p * ld + c];
    if (p != k) M[(size_t)p * ld + c] = tk;
    M[(size_t)k * ld + c] = tp / pv;
}

__global__ void gj_savecol_kernel(int n, int k, const double *__restrict__ M, int ld, double *__restrict__ colk)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) colk[i] = (i == k) ? 0.0 : M[(size_t)i * ld + k];
}

// M[i][c] -= colk[i] * M[k][c] for every row i != k (colk[k] == 0 makes row k a no-op)
__global__ void gj_eliminate_kernel(int n, int k, double *__restrict__ M, int ld, const double *__restrict__ colk)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int i0 = blockIdx.y * 8;
    if (c >= ld) return;
    const double mk = M[(size_t)k * ld + c];
    if (mk == 0.0) return;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int i = i0 + q;
        if (i < n) {
            const double f = colk[i];
            if (f != 0.0) M[(size_t)i * ld + c] -= f * mk;
        }
    }
}

__global__ void gj_extract_kernel(int n, const double *__restrict__ M, int ld, double *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c < n) out[(size_t)r * n + c] = M[(size_t)r * ld + n + c];
}


// ---- blocked form: B = 8 pivots per step, n x n (no augmented half), two launches per step ---------------------------------------------------
// The panel kernel (ONE workgroup, the panel's rows from k0 on in LDS) finds the B pivots by an unblocked elimination of the panel's own columns
// and leaves a small record: the rows that move (at most 2 B of them) with the row each receives, and Dinv = (M_KK)^-1 of the swapped matrix.
// The update kernel is then one pass from the old matrix to the new one (two buffers, swapped every step: a row that moves is simply read from
// where it was -- no row is swapped in memory, nothing is read that the same launch writes):
//     rows of K:  Dinv (columns of K) | Dinv M_K:          other rows:  -M_:K Dinv (columns of K) | M - M_:K (Dinv M_K:)
// -- Gauss-Jordan with the identity's columns stored where the eliminated ones were.  With P the row swaps, the result is (P A)^-1 = A^-1 P^T: the
// columns are put back by one gather at the end (column j of the inverse = column colsrc[j], the swaps applied to 0..n-1 in reverse order).
// 1 471 dofs: 2 x 184 launches and ~6 GB of traffic where the column-by-column form had 2 942 launches and 150 GB.  (A first version did the row
// swaps and the products Dinv M_K:, M_:K in the panel kernel: ~300 KB through ONE compute unit in five dependent phases, 48 us per step.)
constexpr int kGB = 8;
constexpr int kGPanelRows = 3072;                  // largest n: a thread of the panel kernel keeps up to six rows of the panel (kGThreads x kGRowsPer)
struct GjStep {                                    // what the panel kernel leaves for the update
    int nr, pad[3];
    int row[2 * kGB], src[2 * kGB];                // row[z] receives the old row src[z]  (rows not listed stay)
    double Dinv[kGB * kGB];
};
// argmax (|value|, lowest index on ties) across lanes without LDS: DPP moves inside a row of 16 lanes, lane reads across the four rows
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void argmax_take(double &best, int &bi, double &bsv, double ov, int oi, double os)
{
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; bsv = os; }
}
template <int CTRL>
__device__ __forceinline__ void argmax_dpp(double &best, int &bi, double &bsv)
{
    const double ov = dpp_f64<CTRL>(best), os = dpp_f64<CTRL>(bsv);
    const int oi = __builtin_amdgcn_update_dpp(bi, bi, CTRL, 0xf, 0xf, false);
    argmax_take(best, bi, bsv, ov, oi, os);
}
__device__ __forceinline__ void argmax_row16(double &best, int &bi, double &bsv)      // afterwards every lane of a row of 16 holds the row's winner
{
    argmax_dpp<0xB1>(best, bi, bsv);             // quad_perm [1, 0, 3, 2]
    argmax_dpp<0x4E>(best, bi, bsv);             // quad_perm [2, 3, 0, 1]
    argmax_dpp<0x141>(best, bi, bsv);            // row_half_mirror
    argmax_dpp<0x140>(best, bi, bsv);            // row_mirror
}
__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

// (Measured on MI355X with the wall clock at the phase boundaries, ELAG_HIP_LU_TIMING=1, n = 1 471: with the panel in LDS the eight pivots were
//  22-26 us of the kernel -- 16 waves' shuffles, candidate reads and row updates all through the one LDS pipe of the CU.  Now a thread keeps
//  its rows of the panel in registers, the reductions are DPP moves and lane reads, and LDS carries two rows and 16 candidates per pivot.)
//  The kernel is bound by vector-instruction issue (a wave64 instruction holds its SIMD for 4+ cycles, and every wave pays the reductions): 512
//  threads -- two waves per SIMD -- with up to six rows each, empty row slots skipped, one reciprocal of the pivot instead of a division per row.
constexpr int kGThreads = 512;
constexpr int kGRowsPer = 6;                       // rows of the panel per thread: n <= 3 072
// LOAD::row(i, out): the kGB entries of row i in the panel's columns -- read from the matrix (GjPlainRows), or worked out from the matrix of
// the step before (GjLookAheadRows: gj_block_step_kernel)
struct GjPlainRows {
    const double *M;
    int n, k0;
    __device__ __forceinline__ void row(int i, double (&out)[kGB]) const
    {
#pragma unroll
        for (int c = 0; c < kGB; c++) out[c] = k0 + c < n ? M[(size_t)i * n + k0 + c] : 0.0;
    }
};
template <class LOAD>
__device__ __forceinline__ void gj_panel_body(int n, int k0, const LOAD &load, int *__restrict__ piv_all, GjStep *__restrict__ step,
                                              int *__restrict__ singular, long long *__restrict__ dbg)
{
    // dbg (ELAG_HIP_LU_TIMING=1): the 100 MHz wall clock at the kernel's phase boundaries, six per step
    auto stamp = [&](int w) { if (dbg && threadIdx.x == 0) dbg[(size_t)(k0 / kGB) * 6 + w] = (long long)wall_clock64(); };
    stamp(0);
    constexpr int kWaves = kGThreads / 64;
    __shared__ double sv[kWaves], ss[kWaves];        // a candidate per wave: |value|, value
    __shared__ int si[kWaves];
    __shared__ double prow[kGB], qrow[kGB];          // the pivot row and the row it changes places with
    __shared__ int s_piv[kGB], s_src[2 * kGB];
    __shared__ double D[kGB][kGB + 1], ck[kGB], rk[kGB];
    const int t = threadIdx.x, m = n - k0, nb = min(kGB, m);
    // this thread's rows of the panel (positions t, t + 512, ... from k0 on), 64 contiguous bytes each
    double rv[kGRowsPer][kGB];
#pragma unroll
    for (int u = 0; u < kGRowsPer; u++) {
        const int r = t + u * kGThreads;
        double rowv[kGB];
#pragma unroll
        for (int c = 0; c < kGB; c++) rowv[c] = 0.0;
        if (r < m) load.row(k0 + r, rowv);
#pragma unroll
        for (int c = 0; c < kGB; c++) rv[u][c] = (r < m && c < nb) ? rowv[c] : 0.0;
    }
    stamp(1);
#pragma unroll
    for (int q = 0; q < kGB; q++) {
        if (q >= nb) break;                                                    // (uniform)
        // argmax |column q| over the rows from q on, lowest index on ties
        double best = -1.0, bsv = 0.0;
        int bi = m;
#pragma unroll
        for (int u = 0; u < kGRowsPer; u++) {
            if (u * kGThreads >= m) break;                                     // (uniform: an empty slot)
            const int r = t + u * kGThreads;
            const double v = fabs(rv[u][q]);
            if (r >= q && r < m && v > best) { best = v; bi = r; bsv = rv[u][q]; }
        }
        argmax_row16(best, bi, bsv);
        {
            double b = readlane_f64(best, 0), sg = readlane_f64(bsv, 0);
            int ix = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
            for (int l = 16; l < 64; l += 16) argmax_take(b, ix, sg, readlane_f64(best, l), __builtin_amdgcn_readlane(bi, l), readlane_f64(bsv, l));
            if ((t & 63) == 0) { sv[t >> 6] = b; si[t >> 6] = ix; ss[t >> 6] = sg; }
        }
        __syncthreads();
        best = sv[0]; bi = si[0]; bsv = ss[0];                                 // (the waves' candidates, by every thread)
#pragma unroll
        for (int w = 1; w < kWaves; w++) argmax_take(best, bi, bsv, sv[w], si[w], ss[w]);
        const int p = __builtin_amdgcn_readfirstlane(bi < m ? bi : q);
        const double pv = readlane_f64(bi < m ? bsv : 0.0, 0);
        const bool bad = pv == 0.0 || pv != pv;
        if (bad && t == 0) atomicAdd(singular, 1);
        if (t == 0) s_piv[q] = k0 + p;
        // the two rows through LDS: position q takes the pivot row, position p the row that was at q
        const int pt = p % kGThreads, pu = p / kGThreads;
        if (t == pt) {
#pragma unroll
            for (int u = 0; u < kGRowsPer; u++)
                if (u == pu) {
#pragma unroll
                    for (int c = 0; c < kGB; c++) prow[c] = rv[u][c];
                }
        }
        if (t == q) {
#pragma unroll
            for (int c = 0; c < kGB; c++) qrow[c] = rv[0][c];
        }
        __syncthreads();
        double pr[kGB];
#pragma unroll
        for (int c = 0; c < kGB; c++) pr[c] = prow[c];
        if (p != q) {
            if (t == pt) {
#pragma unroll
                for (int u = 0; u < kGRowsPer; u++)
                    if (u == pu) {
#pragma unroll
                        for (int c = 0; c < kGB; c++) rv[u][c] = qrow[c];
                    }
            }
            if (t == q) {
#pragma unroll
                for (int c = 0; c < kGB; c++) rv[0][c] = pr[c];
            }
        }
        // eliminate the rows below
        const double ipv = bad ? 0.0 : 1.0 / pv;
#pragma unroll
        for (int u = 0; u < kGRowsPer; u++) {
            if (u * kGThreads >= m) break;                                     // (uniform)
            const int r = t + u * kGThreads;
            if (r > q && r < m) {
                const double f = rv[u][q] * ipv;
#pragma unroll
                for (int c = q + 1; c < kGB; c++) rv[u][c] -= f * pr[c];
            }
        }
        // (prow / qrow / sv are written again only behind the next pivot's first barrier, and read here in front of it)
    }
    stamp(2);
    if (t >= 64) return;                                                       // the rest is one wavefront's work (no workgroup barrier below)
    __builtin_amdgcn_wave_barrier();
    // The rows that move and, for each, the row it receives: position x ends up with the old row found by walking the swaps backwards from x
    // (8 compare-and-select steps), independently for each of the 2 kGB positions a swap touches -- a lane each, no serial walk over lists.
    if (t < 2 * kGB) {
        const int q0 = t >> 1;
        int y = (q0 < nb) ? ((t & 1) ? s_piv[q0] : k0 + q0) : -1;
        const int x = y;
        for (int q = nb - 1; q >= 0; q--) {
            const int k = k0 + q, p = s_piv[q];
            if (y == k) y = p;
            else if (y == p) y = k;
        }
        s_src[t] = y;
        step->row[t] = x; step->src[t] = y;
    }
    if (t == 0) step->nr = 2 * kGB;
    if (t < nb) piv_all[k0 + t] = s_piv[t];
    __builtin_amdgcn_wave_barrier();
    stamp(3);
    // D = the swapped M_KK (padded with the identity), inverted without further pivoting: the swaps made its leading pivots the ones found above
    const int r = t / kGB, c = t % kGB;
    {
        double rowv[kGB];
#pragma unroll
        for (int q = 0; q < kGB; q++) rowv[q] = 0.0;
        if (r < nb && c < nb) load.row(s_src[2 * r], rowv);                                                            // (position k0 + r is entry 2 r of the list)
        double mine = 0.0;
#pragma unroll
        for (int q = 0; q < kGB; q++) mine = q == c ? rowv[q] : mine;
        D[r][c] = (r < nb && c < nb) ? mine : (r == c ? 1.0 : 0.0);
    }
    __builtin_amdgcn_wave_barrier();
    stamp(4);
    for (int k = 0; k < nb; k++) {
        const double p = D[k][k];
        if (t < kGB) {
            ck[t] = (t == k) ? 0.0 : D[t][k];
            rk[t] = (t == k) ? 1.0 / p : D[k][t] / p;
        }
        __builtin_amdgcn_wave_barrier();
        double v = D[r][c];
        if (r == k) v = rk[c];
        else if (c == k) v = -ck[r] * rk[c];                                   // rk[k] = 1 / p
        else v -= ck[r] * rk[c];
        __builtin_amdgcn_wave_barrier();
        D[r][c] = v;
        __builtin_amdgcn_wave_barrier();
    }
    step->Dinv[t] = D[r][c];
    stamp(5);
}

__global__ void __launch_bounds__(kGThreads) gj_block_panel_kernel(int n, int k0, const double *__restrict__ M, int *__restrict__ piv_all, GjStep *__restrict__ step,
                                                              int *__restrict__ singular, long long *__restrict__ dbg)
{
    gj_panel_body(n, k0, GjPlainRows{M, n, k0}, piv_all, step, singular, dbg);
}

constexpr int kGRows = 16;                           // rows per workgroup of the update (a thread per column)
__global__ void __launch_bounds__(256) gj_block_update_kernel(int n, int k0, const double *__restrict__ Mo, double *__restrict__ Mn, const GjStep *__restrict__ step)
{
    __shared__ GjStep st;
    __shared__ int s_old[kGRows + kGB];              // the old row each of this workgroup's rows -- and each row of K -- receives
    __shared__ double cc[kGRows][kGB];               // ... and its entries in the panel's columns
    for (int e = threadIdx.x; e < (int)(sizeof(GjStep) / sizeof(int)); e += 256) reinterpret_cast<int *>(&st)[e] = reinterpret_cast<const int *>(step)[e];
    __syncthreads();
    const int i0 = blockIdx.y * kGRows;
    const int nb = min(kGB, n - k0);
    if (threadIdx.x < kGRows + kGB) {
        const int i = threadIdx.x < kGRows ? i0 + (int)threadIdx.x : k0 + (int)threadIdx.x - kGRows;
        int o = i;
        for (int z = 0; z < st.nr; z++)
            if (st.row[z] == i) o = st.src[z];
        s_old[threadIdx.x] = min(o, n - 1);
    }
    __syncthreads();
    if (threadIdx.x < kGRows * kGB) {
        const int sr = threadIdx.x / kGB, q = threadIdx.x % kGB;
        cc[sr][q] = (i0 + sr < n && q < nb) ? Mo[(size_t)s_old[sr] * n + k0 + q] : 0.0;
    }
    __syncthreads();
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const bool jcol = j >= k0 && j < k0 + nb;
    // rp = column j of the new pivot rows: Dinv itself on the columns of K, Dinv M_K: elsewhere
    double rp[kGB];
    if (jcol) {
#pragma unroll
        for (int q = 0; q < kGB; q++) rp[q] = q < nb ? st.Dinv[q * kGB + (j - k0)] : 0.0;
    } else {
        double col[kGB];
#pragma unroll
        for (int w = 0; w < kGB; w++) col[w] = w < nb ? Mo[(size_t)s_old[kGRows + w] * n + j] : 0.0;
#pragma unroll
        for (int q = 0; q < kGB; q++) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < kGB; w++) sum += st.Dinv[q * kGB + w] * col[w];
            rp[q] = q < nb ? sum : 0.0;
        }
    }
    // (this thread's column of the workgroup's rows, all requested before the first is used)
    double mo[kGRows];
#pragma unroll
    for (int s = 0; s < kGRows; s++) mo[s] = (i0 + s < n && !jcol) ? Mo[(size_t)s_old[s] * n + j] : 0.0;
#pragma unroll
    for (int s = 0; s < kGRows; s++) {
        const int i = i0 + s;
        if (i >= n) break;
        double *mp = Mn + (size_t)i * n + j;
        if (i >= k0 && i < k0 + nb) { *mp = rp[i - k0]; continue; }           // rows of the pivot block
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < kGB; q++) acc += cc[s][q] * rp[q];
        *mp = jcol ? -acc : mo[s] - acc;
    }
}

// Look-ahead: the next panel needs nothing of the updated matrix but its own kGB columns.  A step is two launches as before, cut differently:
//   gj_lookahead_cols_kernel   the update of columns k0 + kGB .. k0 + 2 kGB - 1 alone (n x kGB entries: a few microseconds), then
//   gj_block_step_kernel       workgroup 0 factorises the next panel out of those columns WHILE the other workgroups update the rest of the matrix
// -- a step costs a small launch plus the longer of panel (22 us) and update (13 us) instead of their sum.  Entry for entry the update's
// operations in the update's order: the inverse is the same to the last bit.
__global__ void __launch_bounds__(256) gj_lookahead_cols_kernel(int n, int k0, const double *__restrict__ Mo, double *__restrict__ Mn, const GjStep *__restrict__ step)
{
    constexpr int kRows = 256 / kGB;                 // 32 rows x kGB columns per workgroup
    __shared__ GjStep st;
    __shared__ int s_old[kRows + kGB];
    __shared__ double cc[kRows][kGB], rpn[kGB][kGB];
    for (int e = threadIdx.x; e < (int)(sizeof(GjStep) / sizeof(int)); e += 256) reinterpret_cast<int *>(&st)[e] = reinterpret_cast<const int *>(step)[e];
    __syncthreads();
    const int i0 = blockIdx.x * kRows, nb = min(kGB, n - k0);
    if (threadIdx.x < kRows + kGB) {
        const int i = threadIdx.x < kRows ? i0 + (int)threadIdx.x : k0 + (int)threadIdx.x - kRows;
        int o = i;
        for (int z = 0; z < st.nr; z++)
            if (st.row[z] == i) o = st.src[z];
        s_old[threadIdx.x] = min(o, n - 1);
    }
    __syncthreads();
    {
        const int sr = threadIdx.x / kGB, q = threadIdx.x % kGB;
        cc[sr][q] = (i0 + sr < n && q < nb) ? Mo[(size_t)s_old[sr] * n + k0 + q] : 0.0;
    }
    if (threadIdx.x < kGB * kGB) {
        // rpn[q][c]: row q of the new pivot rows in column k0 + kGB + c  (the update's rp[q] of that column)
        const int q = threadIdx.x / kGB, c = threadIdx.x % kGB, j = k0 + kGB + c;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kGB; w++) {
            const double col = (w < nb && j < n) ? Mo[(size_t)s_old[kRows + w] * n + j] : 0.0;
            sum += st.Dinv[q * kGB + w] * col;
        }
        rpn[q][c] = q < nb ? sum : 0.0;
    }
    __syncthreads();
    const int sr = threadIdx.x / kGB, c = threadIdx.x % kGB, i = i0 + sr, j = k0 + kGB + c;
    if (i >= n || j >= n) return;
    double *mp = Mn + (size_t)i * n + j;
    if (i >= k0 && i < k0 + nb) { *mp = rpn[i - k0][c]; return; }               // rows of the pivot block
    const double mo = Mo[(size_t)s_old[sr] * n + j];
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < kGB; q++) acc += cc[sr][q] * rpn[q][c];
    *mp = mo - acc;
}

__global__ void __launch_bounds__(kGThreads) gj_block_step_kernel(int n, int k0, const double *__restrict__ Mo, double *__restrict__ Mn, const GjStep *__restrict__ step,
                                                             GjStep *__restrict__ step_next, int *__restrict__ piv_all, int *__restrict__ singular)
{
    static_assert(kGThreads == 512, "two 256-thread halves");
    if (blockIdx.x == 0) {
        // ---- the next panel, out of the columns gj_lookahead_cols_kernel has just written
        if (k0 + kGB < n) gj_panel_body(n, k0 + kGB, GjPlainRows{Mn, n, k0 + kGB}, piv_all, step_next, singular, nullptr);
        return;
    }
    // ---- the update (gj_block_update_kernel's, two row blocks per workgroup) of every column but those
    __shared__ GjStep st;
    __shared__ int s_old[2][kGRows + kGB];
    __shared__ double cc[2][kGRows][kGB];
    for (int e = threadIdx.x; e < (int)(sizeof(GjStep) / sizeof(int)); e += kGThreads) reinterpret_cast<int *>(&st)[e] = reinterpret_cast<const int *>(step)[e];
    __syncthreads();
    const int nb = min(kGB, n - k0);
    const int ntile = (n + 255) / 256, nrow = (n + 2 * kGRows - 1) / (2 * kGRows);
    const int half = (int)(threadIdx.x >> 8), tx = (int)(threadIdx.x & 255);
    const int wid = (int)blockIdx.x - 1;
    const int bx = wid % ntile, by = wid / ntile;
    const int i0 = (by * 2 + half) * kGRows;
    if (tx < kGRows + kGB) {
        const int i = tx < kGRows ? i0 + tx : k0 + tx - kGRows;
        int o = i;
        for (int z = 0; z < st.nr; z++)
            if (st.row[z] == i) o = st.src[z];
        s_old[half][tx] = min(o, n - 1);
    }
    __syncthreads();
    if (tx < kGRows * kGB) {
        const int sr = tx / kGB, q = tx % kGB;
        cc[half][sr][q] = (i0 + sr < n && q < nb) ? Mo[(size_t)s_old[half][sr] * n + k0 + q] : 0.0;
    }
    __syncthreads();
    const int j = bx * 256 + tx;
    if (j >= n || by >= nrow) return;
    if (j >= k0 + kGB && j < k0 + 2 * kGB) return;                            // (the look-ahead columns are done)
    const bool jcol = j >= k0 && j < k0 + nb;
    double rp[kGB];
    if (jcol) {
#pragma unroll
        for (int q = 0; q < kGB; q++) rp[q] = q < nb ? st.Dinv[q * kGB + (j - k0)] : 0.0;
    } else {
        double col[kGB];
#pragma unroll
        for (int w = 0; w < kGB; w++) col[w] = w < nb ? Mo[(size_t)s_old[half][kGRows + w] * n + j] : 0.0;
#pragma unroll
        for (int q = 0; q < kGB; q++) {
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < kGB; w++) sum += st.Dinv[q * kGB + w] * col[w];
            rp[q] = q < nb ? sum : 0.0;
        }
    }
    double mo[kGRows];
#pragma unroll
    for (int sI = 0; sI < kGRows; sI++) mo[sI] = (i0 + sI < n && !jcol) ? Mo[(size_t)s_old[half][sI] * n + j] : 0.0;
#pragma unroll
    for (int sI = 0; sI < kGRows; sI++) {
        const int i = i0 + sI;
        if (i >= n) break;
        double *mp = Mn + (size_t)i * n + j;
        if (i >= k0 && i < k0 + nb) { *mp = rp[i - k0]; continue; }           // rows of the pivot block
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < kGB; q++) acc += cc[half][sI][q] * rp[q];
        *mp = jcol ? -acc : mo[sI] - acc;
    }
}

// colsrc = the row swaps applied to 0 .. n-1 in reverse order (one workgroup, the list in LDS, one thread walks the swaps)
__global__ void __launch_bounds__(1024) gj_colsrc_kernel(int n, const int *__restrict__ piv_all, int *__restrict__ colsrc)
{
    extern __shared__ int sh[];                      // idx[n] | piv[n]
    int *idx = sh, *pv = sh + n;
    for (int i = threadIdx.x; i < n; i += 1024) { idx[i] = i; pv[i] = piv_all[i]; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = n - 1; k >= 0; k--) {
            const int p = pv[k];
            if (p != k) { const int a = idx[k]; idx[k] = idx[p]; idx[p] = a; }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) colsrc[i] = idx[i];
}

__global__ void gj_gather_cols_kernel(int n, const double *__restrict__ M, const int *__restrict__ colsrc, double *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c < n) out[(size_t)r * n + c] = M[(size_t)r * n + colsrc[c]];
}
}  // namespace

namespace {
// the launches of a job go to its stream: the library's helpers launch on ctx().stream, which is swapped for as long as this lives; the job's
// stream first waits for what the compute stream has been given so far (the operator to invert)
struct JobStream {
    Context &c;
    hipStream_t saved;
    explicit JobStream(hipStream_t side) : c(ctx()), saved(c.stream)
    {
        if (!side || side == saved) return;
        hipEvent_t ev = nullptr;
        ELAG_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ELAG_HIP_CHECK(hipEventRecord(ev, saved));
        ELAG_HIP_CHECK(hipStreamWaitEvent(side, ev, 0));
        ELAG_HIP_CHECK(hipEventDestroy(ev));                              // (released by the runtime once the wait has gone through)
        c.stream = side;
    }
    ~JobStream() { c.stream = saved; }
};
// scratch of the blocked LU that outlives the call (a job gives it back in finish(); Temp's stack would have to be unwound in order)
struct GjScratch {
    double *Mb = nullptr;
    int *piv_all = nullptr, *colsrc = nullptr, *singular = nullptr;
    GjStep *step = nullptr, *step2 = nullptr;
    void release()
    {
        dev_free(Mb); dev_free(piv_all); dev_free(colsrc); dev_free(singular); dev_free(step); dev_free(step2);
    }
    ~GjScratch() { release(); }
};
}  // namespace

void dense_inverse_general(const DevCSR &A, double *Minv, DenseJob *job)
{
    const int n = A.nrows, ld = 2 * n;
    ELAG_REQUIRE(n == A.ncols, "dense inverse needs a square matrix");
    if (n == 0) return;
    const char *e_blk = getenv("ELAG_HIP_LU_BLOCKED");                    // (0 = the column-by-column form on [A | I]: A/B runs and a test)
    {
        // the default form -- blocked, with look-ahead -- on scratch of its own, beside the rest of a build when the caller brings a job
        const char *e_tim = getenv("ELAG_HIP_LU_TIMING"), *e_ovl = getenv("ELAG_HIP_LU_LOOKAHEAD");
        if (!(e_blk && !atoi(e_blk)) && n <= kGPanelRows && !(e_tim && atoi(e_tim)) && !(e_ovl && !atoi(e_ovl))) {
            auto B = std::make_shared<GjScratch>();
            const hipStream_t side = job ? job->stream : nullptr;
            {
                JobStream js(side);
                hipStream_t st = ctx().stream;
                B->Mb = dev_alloc<double>((size_t)n * n * 2);                // two buffers: every step reads one and writes the other
                B->piv_all = dev_alloc<int>((size_t)n); B->colsrc = dev_alloc<int>((size_t)n); B->singular = dev_alloc<int>(1);
                B->step = dev_alloc<GjStep>(1); B->step2 = dev_alloc<GjStep>(1);
                double *Mo = B->Mb, *Mn = B->Mb + (size_t)n * n;
                zero_async(Mo, sizeof(double) * (size_t)n * n);
                zero_async(B->singular, sizeof(int));
                hipLaunchKernelGGL(dense_scatter_plain_kernel, dim3((n + kB - 1) / kB), dim3(kB), 0, st, n, (const int *)A.I, (const int *)A.J, (const double *)A.a, Mo);
                // the update of a step and the panel of the next one in one launch (gj_block_step_kernel); the first panel by itself
                GjStep *sa = B->step, *sb = B->step2;
                hipLaunchKernelGGL(gj_block_panel_kernel, dim3(1), dim3(kGThreads), 0, st, n, 0, (const double *)Mo, B->piv_all, sa, B->singular, (long long *)nullptr);
                const dim3 gs(1 + ((n + 255) / 256) * ((n + 2 * kGRows - 1) / (2 * kGRows)));
                for (int k0 = 0; k0 < n; k0 += kGB) {
                    if (k0 + kGB < n)
                        hipLaunchKernelGGL(gj_lookahead_cols_kernel, dim3((n + 31) / 32), dim3(256), 0, st, n, k0, (const double *)Mo, Mn, (const GjStep *)sa);
                    hipLaunchKernelGGL(gj_block_step_kernel, gs, dim3(kGThreads), 0, st, n, k0, (const double *)Mo, Mn, (const GjStep *)sa, sb, B->piv_all, B->singular);
                    std::swap(Mo, Mn);
                    std::swap(sa, sb);
                }
                hipLaunchKernelGGL(gj_colsrc_kernel, dim3(1), dim3(1024), sizeof(int) * 2 * (size_t)n, st, n, (const int *)B->piv_all, B->colsrc);
                hipLaunchKernelGGL(gj_gather_cols_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)Mo, (const int *)B->colsrc, Minv);
                ELAG_LAUNCH_CHECK();
            }
            auto tail = [B]() {
                const int bad = fetch_int(B->singular);
                B->release();
                if (bad) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is singular (%d zero pivots in the dense LU)", bad);
            };
            if (side && side != ctx().stream) job->finish = [tail, side]() { ELAG_HIP_CHECK(hipStreamSynchronize(side)); tail(); };
            else tail();
            return;
        }
    }
    hipStream_t st = ctx().stream;
    if (!(e_blk && !atoi(e_blk)) && n <= kGPanelRows) {
        double *Mb = dev_alloc<double>((size_t)n * n * 2);                   // two buffers: every step reads one and writes the other
        try {
            Temp<int> piv_all((size_t)n), colsrc((size_t)n), singular(1);
            Temp<GjStep> step(1);
            const char *e_tim = getenv("ELAG_HIP_LU_TIMING");
            const size_t nsteps = (size_t)(n + kGB - 1) / kGB;
            Temp<long long> dbg(e_tim && atoi(e_tim) ? nsteps * 6 : 1);
            long long *dbgp = e_tim && atoi(e_tim) ? dbg.p : nullptr;
            double *Mo = Mb, *Mn = Mb + (size_t)n * n;
            zero_async(Mo, sizeof(double) * (size_t)n * n);
            zero_async(singular.p, sizeof(int));
            hipLaunchKernelGGL(dense_scatter_plain_kernel, dim3((n + kB - 1) / kB), dim3(kB), 0, st, n, (const int *)A.I, (const int *)A.J, (const double *)A.a, Mo);
            const dim3 gu((n + 255) / 256, (n + kGRows - 1) / kGRows);
            const char *e_ovl = getenv("ELAG_HIP_LU_LOOKAHEAD");            // (0 = a panel launch and an update launch per step: A/B runs and a test)
            if (!(e_ovl && !atoi(e_ovl)) && !dbgp) {
                // the update of a step and the panel of the next one in one launch (gj_block_step_kernel); the first panel by itself
                Temp<GjStep> step2(1);
                GjStep *sa = step.p, *sb = step2.p;
                hipLaunchKernelGGL(gj_block_panel_kernel, dim3(1), dim3(kGThreads), 0, st, n, 0, (const double *)Mo, piv_all.p, sa, singular.p, (long long *)nullptr);
                const dim3 gs(1 + ((n + 255) / 256) * ((n + 2 * kGRows - 1) / (2 * kGRows)));
                for (int k0 = 0; k0 < n; k0 += kGB) {
                    if (k0 + kGB < n)
                        hipLaunchKernelGGL(gj_lookahead_cols_kernel, dim3((n + 31) / 32), dim3(256), 0, st, n, k0, (const double *)Mo, Mn, (const GjStep *)sa);
                    hipLaunchKernelGGL(gj_block_step_kernel, gs, dim3(kGThreads), 0, st, n, k0, (const double *)Mo, Mn, (const GjStep *)sa, sb, piv_all.p, singular.p);
                    std::swap(Mo, Mn);
                    std::swap(sa, sb);
                }
            } else
            for (int k0 = 0; k0 < n; k0 += kGB) {
                hipLaunchKernelGGL(gj_block_panel_kernel, dim3(1), dim3(kGThreads), 0, st, n, k0, (const double *)Mo, piv_all.p, step.p,
                                   singular.p, dbgp);
                hipLaunchKernelGGL(gj_block_update_kernel, gu, dim3(256), 0, st, n, k0, (const double *)Mo, Mn, (const GjStep *)step.p);
                std::swap(Mo, Mn);
            }
            hipLaunchKernelGGL(gj_colsrc_kernel, dim3(1), dim3(1024), sizeof(int) * 2 * (size_t)n, st, n, (const int *)piv_all.p, colsrc.p);
            hipLaunchKernelGGL(gj_gather_cols_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)Mo, (const int *)colsrc.p, Minv);
            ELAG_LAUNCH_CHECK();
            const int bad = fetch_int(singular.p);
            if (dbgp) {
                std::vector<long long> h = host_from_dev(dbgp, nsteps * 6);
                double acc[5] = {0, 0, 0, 0, 0};
                for (size_t q = 0; q < nsteps; q++)
                    for (int w = 0; w < 5; w++) acc[w] += (double)(h[q * 6 + w + 1] - h[q * 6 + w]) * 0.01;     // 100 MHz ticks -> us
                fprintf(stderr, "[elag_hip] blocked LU panel kernel, n = %d, mean us per step: load %.2f, pivots %.2f, swap lists %.2f, D load %.2f, D inverse %.2f\n", n,
                        acc[0] / nsteps, acc[1] / nsteps, acc[2] / nsteps, acc[3] / nsteps, acc[4] / nsteps);
            }
            if (bad) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is singular (%d zero pivots in the dense LU)", bad);
        } catch (...) {
            dev_free(Mb);
            throw;
        }
        dev_free(Mb);
        return;
    }
    double *M = dev_alloc<double>((size_t)n * ld);
    try {
        Temp<int> piv(1), singular(1);
        Temp<double> pivval(1), colk((size_t)n);
        zero_async(M, sizeof(double) * (size_t)n * ld);
        zero_async(singular.p, sizeof(int));
        hipLaunchKernelGGL(dense_scatter_kernel, dim3((n + kB - 1) / kB), dim3(kB), 0, st, n, (const int *)A.I, (const int *)A.J,
                           (const double *)A.a, M, ld);
        const dim3 gcol((ld + kB - 1) / kB), grow((n + kB - 1) / kB), gel((ld + kB - 1) / kB, (n + 7) / 8);
        const char *e_fuse = getenv("ELAG_HIP_LU_FUSED_COLUMN");          // (0 = the four launches per column: A/B runs and a test)
        const bool fuse = !(e_fuse && !atoi(e_fuse));
        for (int k = 0; k < n; k++) {
            if (fuse) {
                hipLaunchKernelGGL(gj_column_kernel, dim3(1), dim3(1024), 0, st, n, k, M, ld, colk.p, singular.p);
            } else {
                hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(1024), 0, st, n, k, (const double *)M, ld, piv.p, pivval.p);
                hipLaunchKernelGGL(gj_swap_scale_kernel, gcol, dim3(kB), 0, st, k, M, ld, (const int *)piv.p, (const double *)pivval.p, singular.p);
                hipLaunchKernelGGL(gj_savecol_kernel, grow, dim3(kB), 0, st, n, k, (const double *)M, ld, colk.p);
            }
            hipLaunchKernelGGL(gj_eliminate_kernel, gel, dim3(kB), 0, st, n, k, M, ld, (const double *)colk.p);
        }
        hipLaunchKernelGGL(gj_extract_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)M, ld, Minv);
        ELAG_LAUNCH_CHECK();
        const int bad = fetch_int(singular.p);
        if (bad) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is singular (%d zero pivots in the dense LU)", bad);
    } catch (...) {
        dev_free(M);
        throw;
    }
    dev_free(M);
}

// ---- symmetric positive definite operators: in-place Gauss-Jordan without pivoting ---------------------------------------------
// (the coarsest operator of the SPD single-form hierarchies when it is too big for the host's scalar Cholesky: 1 015 dofs on the H(div)
// hierarchy of cube456 x3 cost 358 ms there, most of a 367 ms setup).  Two launches per pivot: save column k and the scaled row k, then
// one sweep over the matrix  a_ij -= a_ik a_kj / p,  a_ik = -a_ik / p,  a_kj = a_kj / p,  a_kk = 1 / p.
namespace {
__global__ void spd_prep_kernel(int n, int k, const double *__restrict__ M, double *__restrict__ colk, double *__restrict__ rowk, int *__restrict__ bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double p = M[(size_t)k * n + k];
    if (i == 0 && !(p > 0.0)) atomicAdd(bad, 1);
    colk[i] = (i == k) ? 0.0 : M[(size_t)i * n + k];
    rowk[i] = (i == k) ? 1.0 / p : M[(size_t)k * n + i] / p;
}
__global__ void spd_sweep_kernel(int n, int k, double *__restrict__ M, const double *__restrict__ colk, const double *__restrict__ rowk)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i0 = blockIdx.y * 8;
    if (j >= n) return;
    const double rk = rowk[j];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int i = i0 + q;
        if (i >= n) break;
        double *m = M + (size_t)i * n + j;
        if (i == k) *m = rk;
        else if (j == k) *m = -colk[i] * rk;         // rowk[k] = 1 / p
        else *m -= colk[i] * rk;
    }
}
__global__ void symmetrise_kernel(int n, const double *__restrict__ M, double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
    if (j < n) out[(size_t)i * n + j] = 0.5 * (M[(size_t)i * n + j] + M[(size_t)j * n + i]);
}
}  // namespace

namespace {
// The same elimination for operators that fit ONE workgroup's LDS (n <= 141: n^2 doubles in the CU's 160 KB): scatter, symmetrise, n pivots
// with two barriers each, symmetrise, store -- one launch of ~0.25 ms where the host's Cholesky + 2 n triangular solves (and the copies
// either way) took 1.0 ms of a 5.8 ms cube456 x3 setup and the multi-launch form above 2 n launches.
constexpr int kLdsInvMax = 141;            // (n^2 doubles + the two pivot vectors within the 160 KB of a CU)
constexpr int kLdsInvBlock = 1024;
__global__ __launch_bounds__(kLdsInvBlock) void spd_inverse_lds_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                                                   double *__restrict__ Minv, int *__restrict__ bad)
{
    extern __shared__ double Ml[];          // n x n, row-major
    __shared__ double colk[kLdsInvMax + 1], rowk[kLdsInvMax + 1];
    // a thread owns column tx of the rows ty, ty + kRowsPerPass, ...: no index arithmetic inside the pivot loop
    constexpr int kCols = 144, kRowsPerPass = kLdsInvBlock / kCols;          // 7 rows x 144 columns = 1008 of the 1024 threads
    const int t = threadIdx.x, nt = blockDim.x;
    const int tx = t % kCols, ty = t / kCols;
    const bool own = tx < n && ty < kRowsPerPass;
    for (int e = t; e < n * n; e += nt) Ml[e] = 0.0;
    __syncthreads();
    for (int i = t; i < n; i += nt)
        for (int q = I[i]; q < I[i + 1]; q++) Ml[i * n + J[q]] += a[q];       // (one thread per row: repeated columns add up in storage order)
    __syncthreads();
    // exactly symmetric first, like the host path: the upper triangle's owner writes both
    if (own)
        for (int i = ty; i < n; i += kRowsPerPass)
            if (tx > i) { const double v = 0.5 * (Ml[i * n + tx] + Ml[tx * n + i]); Ml[i * n + tx] = v; Ml[tx * n + i] = v; }
    __syncthreads();
    for (int k = 0; k < n; k++) {
        const double p = Ml[k * n + k];
        if (t == 0 && !(p > 0.0)) atomicAdd(bad, 1);
        if (t < n) {
            colk[t] = (t == k) ? 0.0 : Ml[t * n + k];
            rowk[t] = (t == k) ? 1.0 / p : Ml[k * n + t] / p;
        }
        __syncthreads();
        if (own) {
            const double rk = rowk[tx];
            for (int i = ty; i < n; i += kRowsPerPass) {
                double *m = Ml + i * n + tx;
                if (i == k) *m = rk;
                else if (tx == k) *m = -colk[i] * rk;          // rowk[k] = 1 / p
                else *m -= colk[i] * rk;
            }
        }
        __syncthreads();
    }
    if (own)
        for (int i = ty; i < n; i += kRowsPerPass) Minv[(size_t)i * n + tx] = 0.5 * (Ml[i * n + tx] + Ml[tx * n + i]);
}

// The same launch with the matrix in REGISTERS during the elimination: the sweep above reads and writes every entry in LDS at every pivot (three LDS
// accesses per entry and pivot: 2.6 us per pivot at n = 141, LDS-throughput bound), here a thread keeps its column's entries of the rows ty, ty + 7,
// ... (21 registers) and a pivot costs it the broadcast reads of column k (two entries per read), one read of row k, one barrier (the two pivot
// vectors are double-buffered).  The loop over the pivots is unrolled over the register that holds row k, so no register is indexed dynamically.
// Same arithmetic, entry by entry, as the LDS sweep (ELAG_HIP_SPD_REG=0 selects that one).
constexpr int kRegCols = 144, kRegGroups = kLdsInvBlock / kRegCols, kRegRows = (kLdsInvMax + kRegGroups - 1) / kRegGroups;   // 7 groups x 21 rows
__global__ __launch_bounds__(kLdsInvBlock) void spd_inverse_reg_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                                                   double *__restrict__ Minv, int *__restrict__ bad)
{
    extern __shared__ double Ml[];          // n x n, row-major: the scatter, the first symmetrisation and the last one
    __shared__ double colk[2][kRegGroups][kRegRows + 1];   // colk[.][ty][s] = M[ty + 7 s][k]
    __shared__ double rowk[2][kRegCols];                    // M[k][.]
    const int t = threadIdx.x, nt = blockDim.x;
    const int tx = t % kRegCols, ty = t / kRegCols;
    const bool own = tx < n && ty < kRegGroups;
    for (int e = t; e < n * n; e += nt) Ml[e] = 0.0;
    __syncthreads();
    for (int i = t; i < n; i += nt)
        for (int q = I[i]; q < I[i + 1]; q++) Ml[i * n + J[q]] += a[q];       // (one thread per row: repeated columns add up in storage order)
    __syncthreads();
    if (own)
        for (int i = ty; i < n; i += kRegGroups)
            if (tx > i) { const double v = 0.5 * (Ml[i * n + tx] + Ml[tx * n + i]); Ml[i * n + tx] = v; Ml[tx * n + i] = v; }
    __syncthreads();
    double m[kRegRows];
#pragma unroll
    for (int s = 0; s < kRegRows; s++) {
        const int i = ty + kRegGroups * s;
        m[s] = (own && i < n) ? Ml[i * n + tx] : 0.0;
    }
    int nbad = 0;
#pragma unroll
    for (int s0 = 0; s0 < kRegRows; s0++) {
        if (kRegGroups * s0 >= n) break;
        for (int ty0 = 0; ty0 < kRegGroups; ty0++) {
            const int k = kRegGroups * s0 + ty0;
            if (k >= n) break;
            const int b = k & 1;
            if (ty == ty0 && ty < kRegGroups) rowk[b][tx] = m[s0];
            if (tx == k && ty < kRegGroups) {
#pragma unroll
                for (int s = 0; s < kRegRows; s++) colk[b][ty][s] = m[s];
            }
            __syncthreads();
            const double p = rowk[b][k];
            if (!(p > 0.0)) nbad++;
            const double rk = (tx == k) ? 1.0 / p : rowk[b][tx] / p;
            if (ty < kRegGroups) {
#pragma unroll
                for (int s = 0; s < kRegRows; s++) {
                    const double c = colk[b][ty][s];
                    const double upd = (tx == k) ? -c * rk : m[s] - c * rk;          // rk = 1 / p on column k
                    m[s] = (s == s0 && ty == ty0) ? rk : upd;
                }
            }
        }
    }
    if (t == 0 && nbad) atomicAdd(bad, nbad);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < kRegRows; s++) {
        const int i = ty + kRegGroups * s;
        if (own && i < n) Ml[i * n + tx] = m[s];
    }
    __syncthreads();
    if (own)
        for (int i = ty; i < n; i += kRegGroups) Minv[(size_t)i * n + tx] = 0.5 * (Ml[i * n + tx] + Ml[tx * n + i]);
}
}  // namespace

namespace {
// ---- the same elimination in blocks of kNB pivots (operators beyond one workgroup's LDS) -------------------------------------------------
// 2 n launches -- 2 030 for the 1 015-dof coarsest operator of the cube456 x3 H(div) hierarchy, 7.8 of its 13 ms of setup -- become 2 n / kNB:
// with K the block of pivots, D = A_KK,
//     A_KK <- D^-1,   A_Kj <- D^-1 A_Kj,   A_iK <- -A_iK D^-1,   A_ij <- A_ij - A_iK (D^-1 A_Kj)        (i, j outside K)
// which is kNB steps of the scalar sweep at once.  Launch 1 (panel): every workgroup inverts D in LDS by the scalar sweep (32 x 32: cheaper than a
// launch of its own), writes its columns of the row panel R = D^-1 A_K: and its rows of a copy of the column panel A_:K.  Launch 2 (update): 32 x 32
// tiles, the two panels' tiles staged in LDS.
constexpr int kNB = 32;
constexpr int kPanelCols = 32;
__global__ __launch_bounds__(1024) void spd_block_panel_kernel(int n, int K, const double *__restrict__ M, double *__restrict__ Dinv, double *__restrict__ R,
                                                               double *__restrict__ Cc, int *__restrict__ bad)
{
    __shared__ double D[kNB][kNB + 1];
    __shared__ double ck[2][kNB], rk[2][kNB];
    const int t = threadIdx.x, r = t >> 5, c = t & 31;
    const int nb = min(kNB, n - K);
    // D = A_KK, padded with the identity: a thread keeps its entry in a register during the elimination, the pivot's row and column go through
    // LDS (double-buffered: one barrier per pivot); the same operations, entry by entry, as the sweep over a matrix in LDS
    double m = (r < nb && c < nb) ? M[(size_t)(K + r) * n + K + c] : (r == c ? 1.0 : 0.0);
    int nbad = 0;
    for (int k = 0; k < nb; k++) {
        const int b = k & 1;
        if (c == k) ck[b][r] = m;
        if (r == k) rk[b][c] = m;
        __syncthreads();
        const double p = rk[b][k];
        if (!(p > 0.0)) nbad++;
        const double rkc = (c == k) ? 1.0 / p : rk[b][c] / p;
        if (r == k) m = rkc;
        else if (c == k) m = -ck[b][r] * rkc;               // rkc = 1 / p
        else m -= ck[b][r] * rkc;
    }
    if (t == 0 && blockIdx.x == 0 && nbad) atomicAdd(bad, nbad);
    D[r][c] = m;
    __syncthreads();
    if (blockIdx.x == 0) Dinv[r * kNB + c] = D[r][c];
    // this workgroup's kPanelCols columns of R = D^-1 A_K:  (row r of the panel; 32 columns per workgroup: at 256 the 1 015-dof
    // coarsest H(div) operator kept four workgroups busy for 45 us per panel, 2.0 ms of coarse-solver setup against 1.1)
    const int j0 = blockIdx.x * kPanelCols;
    for (int jj = 0; jj < kPanelCols; jj += 32) {
        const int j = j0 + jj + c;
        if (j < n && r < nb) {
            double sum = 0.0;
            for (int q = 0; q < nb; q++) sum += D[r][q] * M[(size_t)(K + q) * n + j];
            R[(size_t)r * n + j] = sum;
        }
    }
    // ... and as many rows of the column panel (the update overwrites it)
    for (int ii = 0; ii < kPanelCols; ii += 32) {
        const int i = j0 + ii + r;
        if (i < n) Cc[(size_t)i * kNB + c] = c < nb ? M[(size_t)i * n + K + c] : 0.0;
    }
}
__global__ __launch_bounds__(256) void spd_block_update_kernel(int n, int K, double *__restrict__ M, const double *__restrict__ Dinv, const double *__restrict__ R,
                                                               const double *__restrict__ Cc)
{
    __shared__ double Ct[kNB][kNB + 1], Rt[kNB][kNB + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads, four rows each
    const int i0 = blockIdx.y * kNB, j0 = blockIdx.x * kNB;
    const int nb = min(kNB, n - K);
    const bool irow = i0 == K, jcol = j0 == K;                        // (K is a multiple of the tile size)
    for (int q = ty; q < kNB; q += 8) {
        const int i = i0 + q;
        Ct[q][tx] = (i < n && tx < nb) ? Cc[(size_t)i * kNB + tx] : 0.0;                        // Ct[row][pivot]
        const int j = j0 + tx;
        Rt[q][tx] = (q < nb && j < n) ? (jcol ? (tx < nb ? Dinv[q * kNB + tx] : 0.0) : R[(size_t)q * n + j]) : 0.0;   // Rt[pivot][column]
    }
    __syncthreads();
    const int j = j0 + tx;
    if (j >= n) return;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int il = ty + 8 * s, i = i0 + il;
        if (i >= n) break;
        double *m = M + (size_t)i * n + j;
        if (irow) { *m = Rt[il][tx]; continue; }                      // rows of the pivot block: D^-1 (on the block) or D^-1 A_Kj
        double acc = 0.0;
        for (int q = 0; q < nb; q++) acc += Ct[il][q] * Rt[q][tx];
        *m = jcol ? -acc : *m - acc;
    }
}
}  // namespace

void dense_inverse_spd(const DevCSR &A, double *Minv, DenseJob *job)
{
    const int n = A.nrows;
    ELAG_REQUIRE(n == A.ncols, "dense inverse needs a square matrix");
    if (n == 0) return;
    if (n <= kLdsInvMax) {
        // one launch of one workgroup (0.22 ms at 141 dofs: a latency chain) -- beside the rest of a build when the caller brings a job
        const hipStream_t side = job ? job->stream : nullptr;
        int *bad_flag = dev_alloc<int>(1);
        std::shared_ptr<int> bad(bad_flag, [](int *q) { dev_free(q); });
        static thread_local bool raised = false;
        const size_t lds = sizeof(double) * (size_t)n * n;
        if (!raised && lds > (size_t)(48 << 10)) {
            ELAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(spd_inverse_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(sizeof(double) * kLdsInvMax * kLdsInvMax)));
            ELAG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(spd_inverse_reg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)(sizeof(double) * kLdsInvMax * kLdsInvMax)));
            raised = true;
        }
        {
            JobStream js(side);
            hipStream_t st = ctx().stream;
            zero_async(bad.get(), sizeof(int));
            const char *e_reg = getenv("ELAG_HIP_SPD_REG");                 // (read at every call: a test switches it)
            const bool in_regs = !(e_reg && !atoi(e_reg));
            if (in_regs)
                hipLaunchKernelGGL(spd_inverse_reg_kernel, dim3(1), dim3(kLdsInvBlock), lds, st, n, (const int *)A.I, (const int *)A.J, (const double *)A.a, Minv, bad.get());
            else
                hipLaunchKernelGGL(spd_inverse_lds_kernel, dim3(1), dim3(kLdsInvBlock), lds, st, n, (const int *)A.I, (const int *)A.J, (const double *)A.a, Minv, bad.get());
            ELAG_LAUNCH_CHECK();
        }
        auto tail = [bad]() {
            const int nbad = fetch_int(bad.get());
            if (nbad) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is not positive definite (%d non-positive pivots)", nbad);
        };
        if (side && side != ctx().stream) job->finish = [tail, side]() { ELAG_HIP_CHECK(hipStreamSynchronize(side)); tail(); };
        else tail();
        return;
    }
    {
        // the blocked form (the default) on scratch of its own, beside the rest of a build when the caller brings a job
        const char *e_blk0 = getenv("ELAG_HIP_SPD_BLOCKED");
        if (!(e_blk0 && !atoi(e_blk0))) {
            struct Scratch {
                double *M = nullptr, *Dinv = nullptr, *Rp = nullptr, *Cc = nullptr;
                int *bad = nullptr;
                void release() { dev_free(M); dev_free(Dinv); dev_free(Rp); dev_free(Cc); dev_free(bad); }
                ~Scratch() { release(); }
            };
            auto B = std::make_shared<Scratch>();
            const hipStream_t side = job ? job->stream : nullptr;
            {
                JobStream js(side);
                hipStream_t st = ctx().stream;
                B->M = dev_alloc<double>((size_t)n * n); B->bad = dev_alloc<int>(1);
                B->Dinv = dev_alloc<double>((size_t)kNB * kNB); B->Rp = dev_alloc<double>((size_t)kNB * n); B->Cc = dev_alloc<double>((size_t)n * kNB);
                double *M = B->M;
                zero_async(M, sizeof(double) * (size_t)n * n);
                zero_async(B->bad, sizeof(int));
                hipLaunchKernelGGL(dense_scatter_plain_kernel, dim3((n + kB - 1) / kB), dim3(kB), 0, st, n, (const int *)A.I, (const int *)A.J, (const double *)A.a, M);
                // the (numerically symmetric) input made exactly symmetric first, like the host path
                hipLaunchKernelGGL(symmetrise_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)M, Minv);
                dev_copy(M, Minv, sizeof(double) * (size_t)n * n, st);
                const dim3 gp((n + kPanelCols - 1) / kPanelCols), gu((n + kNB - 1) / kNB, (n + kNB - 1) / kNB);
                for (int K = 0; K < n; K += kNB) {
                    hipLaunchKernelGGL(spd_block_panel_kernel, gp, dim3(1024), 0, st, n, K, (const double *)M, B->Dinv, B->Rp, B->Cc, B->bad);
                    hipLaunchKernelGGL(spd_block_update_kernel, gu, dim3(256), 0, st, n, K, M, (const double *)B->Dinv, (const double *)B->Rp, (const double *)B->Cc);
                }
                // exact symmetry of the stored inverse keeps the cycle symmetric to rounding
                hipLaunchKernelGGL(symmetrise_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)M, Minv);
                ELAG_LAUNCH_CHECK();
            }
            auto tail = [B]() {
                const int nbad = fetch_int(B->bad);
                B->release();
                if (nbad) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is not positive definite (%d non-positive pivots)", nbad);
            };
            if (side && side != ctx().stream) job->finish = [tail, side]() { ELAG_HIP_CHECK(hipStreamSynchronize(side)); tail(); };
            else tail();
            return;
        }
    }
    hipStream_t st = ctx().stream;
    double *M = dev_alloc<double>((size_t)n * n);
    try {
        Temp<int> bad(1);
        Temp<double> colk((size_t)n), rowk((size_t)n);
        zero_async(M, sizeof(double) * (size_t)n * n);
        zero_async(bad.p, sizeof(int));
        hipLaunchKernelGGL(dense_scatter_plain_kernel, dim3((n + kB - 1) / kB), dim3(kB), 0, st, n, (const int *)A.I, (const int *)A.J, (const double *)A.a, M);
        // the (numerically symmetric) input made exactly symmetric first, like the host path
        hipLaunchKernelGGL(symmetrise_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)M, Minv);
        dev_copy(M, Minv, sizeof(double) * (size_t)n * n, st);
        const char *e_blk = getenv("ELAG_HIP_SPD_BLOCKED");                 // (0 = the pivot-by-pivot sweeps: A/B runs and a test)
        if (!(e_blk && !atoi(e_blk))) {
            Temp<double> Dinv((size_t)kNB * kNB), Rp((size_t)kNB * n), Cc((size_t)n * kNB);
            const dim3 gp((n + kPanelCols - 1) / kPanelCols), gu((n + kNB - 1) / kNB, (n + kNB - 1) / kNB);
            for (int K = 0; K < n; K += kNB) {
                hipLaunchKernelGGL(spd_block_panel_kernel, gp, dim3(1024), 0, st, n, K, (const double *)M, Dinv.p, Rp.p, Cc.p, bad.p);
                hipLaunchKernelGGL(spd_block_update_kernel, gu, dim3(256), 0, st, n, K, M, (const double *)Dinv.p, (const double *)Rp.p, (const double *)Cc.p);
            }
            ELAG_LAUNCH_CHECK();
            ELAG_HIP_CHECK(hipStreamSynchronize(st));                     // (the panels go out of scope)
        } else {
        const dim3 g1((n + kB - 1) / kB), g2((n + kB - 1) / kB, (n + 7) / 8);
        for (int k = 0; k < n; k++) {
            hipLaunchKernelGGL(spd_prep_kernel, g1, dim3(kB), 0, st, n, k, (const double *)M, colk.p, rowk.p, bad.p);
            hipLaunchKernelGGL(spd_sweep_kernel, g2, dim3(kB), 0, st, n, k, M, (const double *)colk.p, (const double *)rowk.p);
        }
        }
        // exact symmetry of the stored inverse keeps the cycle symmetric to rounding
        hipLaunchKernelGGL(symmetrise_kernel, dim3((n + kB - 1) / kB, n), dim3(kB), 0, st, n, (const double *)M, Minv);
        ELAG_LAUNCH_CHECK();
        const int nbad = fetch_int(bad.p);
        if (nbad) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is not positive definite (%d non-positive pivots)", nbad);
    } catch (...) {
        dev_free(M);
        throw;
    }
    dev_free(M);
}

namespace {
__global__ void dense_times_csr_t_kernel(int nc, int n, const double *__restrict__ Bc, const int *__restrict__ TI, const int *__restrict__ TJ,
                                         const double *__restrict__ Ta, double *__restrict__ X)
{
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)nc * n) return;
    const int r = (int)(id / n), j = (int)(id % n);
    double s = 0.0;
    for (int q = TI[j]; q < TI[j + 1]; q++) s += Bc[(size_t)r * nc + TJ[q]] * Ta[q];
    X[(size_t)r * n + j] = s;
}
__global__ void dense_level_vx_kernel(int n, const int *__restrict__ VI, const int *__restrict__ VJ, const double *__restrict__ Va,
                                      const double *__restrict__ vw, const double *__restrict__ X, double *__restrict__ B)
{
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)n * n) return;
    const int i = (int)(id / n), j = (int)(id % n);
    double s = 0.0;
    for (int q = VI[i]; q < VI[i + 1]; q++) s += Va[q] * X[(size_t)VJ[q] * n + j];
    B[(size_t)i * n + j] = vw ? -vw[i] * s : s;
}
// one thread per row: the row's entries are added one after the other (a row may store its diagonal or not)
__global__ void dense_level_u_kernel(int n, const int *__restrict__ UI, const int *__restrict__ UJ, const double *__restrict__ Ua,
                                     const double *__restrict__ w, double *__restrict__ B)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double *__restrict__ row = B + (size_t)i * n;
    if (w) {
        row[i] += 2.0 * w[i];
        for (int q = UI[i]; q < UI[i + 1]; q++) row[UJ[q]] -= w[i] * Ua[q] * w[UJ[q]];
    } else {
        for (int q = UI[i]; q < UI[i + 1]; q++) row[UJ[q]] += Ua[q];
    }
}
}  // namespace

namespace {
// out (nr x m, row-major) = (ident ? I : 0) + beta * out + alpha * (S X): one thread per entry, the row's terms in storage order
__global__ void csr_times_dense_kernel(int nr, int m, const int *__restrict__ SI, const int *__restrict__ SJ, const double *__restrict__ Sa, double alpha,
                                       const double *__restrict__ X, double beta, int ident, double *__restrict__ out)
{
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)nr * m) return;
    const int i = (int)(id / m), c = (int)(id % m);
    double s = 0.0;
    for (int q = SI[i]; q < SI[i + 1]; q++) s += Sa[q] * X[(size_t)SJ[q] * m + c];
    double v = alpha * s + (ident && i == c ? 1.0 : 0.0);
    if (beta != 0.0) v += beta * out[id];
    out[id] = v;
}
// out (nr x m) = M (nr x nk, row-major) X (nk x m)
__global__ void dense_mm_kernel(int nr, int nk, int m, const double *__restrict__ M, const double *__restrict__ X, double *__restrict__ out)
{
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)nr * m) return;
    const int i = (int)(id / m), c = (int)(id % m);
    double s = 0.0;
    for (int q = 0; q < nk; q++) s += M[(size_t)i * nk + q] * X[(size_t)q * m + c];
    out[id] = s;
}
}  // namespace

namespace {
// The same product on the matrix cores: the one dense GEMM of the path (the dense operator of a Gauss-Seidel hierarchy's coarse tail: 840 x 840 x 5 739
// at cube456 x3, 8.1 GFLOP -- 1.9 ms at one thread per entry of the result).  v_mfma_f64_16x16x4_f64: lane l holds A[l & 15][k = l >> 4] and
// B[k = l >> 4][l & 15], result register r of lane l is C[(l >> 4) + 4 r][l & 15].  A workgroup computes a 64 x 64 tile, its four wavefronts
// 32 x 32 corners (2 x 2 instruction tiles), the operands' 64 x 16 and 16 x 64 panels staged in LDS.  fp64 FMAs in ascending k per entry (fused,
// where the scalar kernel rounds the product and the sum separately).
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void dense_mm_mfma_kernel(int nr, int nk, int m, const double *__restrict__ M, const double *__restrict__ X, double *__restrict__ out)
{
    __shared__ double As[64][17];      // [row][k]
    __shared__ double Bs[16][66];      // [k][column]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int wr = (wv >> 1) * 32, wc = (wv & 1) * 32;
    mfma_d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
    const int ar = tid >> 2, ak = (tid & 3) * 4;       // this thread's four consecutive k of one row of the A panel
    const int bk = tid >> 4, bc = (tid & 15) * 4;      // ... and four consecutive columns of one k of the B panel
    for (int k0 = 0; k0 < nk; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = i0 + ar, k = k0 + ak + q;
            As[ar][ak + q] = (i < nr && k < nk) ? M[(size_t)i * nk + k] : 0.0;
            const int kk = k0 + bk, j = j0 + bc + q;
            Bs[bk][bc + q] = (kk < nk && j < m) ? X[(size_t)kk * m + j] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ks += 4) {
            const int kl = ks + (lane >> 4), l15 = lane & 15;
            const double a0 = As[wr + l15][kl], a1 = As[wr + 16 + l15][kl];
            const double b0 = Bs[kl][wc + l15], b1 = Bs[kl][wc + 16 + l15];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = i0 + wr + a * 16 + (lane >> 4) + 4 * r, j = j0 + wc + b * 16 + (lane & 15);
                if (i < nr && j < m) out[(size_t)i * m + j] = acc[a][b][r];
            }
}
}  // namespace

void csr_times_dense(const DevCSR &S, int m, double alpha, const double *X, double beta, bool add_identity, double *out)
{
    const long long tot = (long long)S.nrows * m;
    if (tot == 0) return;
    hipLaunchKernelGGL(csr_times_dense_kernel, dim3((unsigned)((tot + kB - 1) / kB)), dim3(kB), 0, ctx().stream, S.nrows, m, (const int *)S.I, (const int *)S.J,
                       (const double *)S.a, alpha, X, beta, add_identity ? 1 : 0, out);
    ELAG_LAUNCH_CHECK();
}

void dense_mm(int nr, int nk, int m, const double *M, const double *X, double *out)
{
    const long long tot = (long long)nr * m;
    if (tot == 0) return;
    const char *e = getenv("ELAG_HIP_DENSE_MFMA");                     // (read at every call: a test compares the two kernels)
    if (!(e && !atoi(e)) && nk > 0 && (long long)nr * nk * m >= (1ll << 18)) {
        hipLaunchKernelGGL(dense_mm_mfma_kernel, dim3((unsigned)((m + 63) / 64), (unsigned)((nr + 63) / 64)), dim3(256), 0, ctx().stream, nr, nk, m, M, X, out);
        ELAG_LAUNCH_CHECK();
        return;
    }
    hipLaunchKernelGGL(dense_mm_kernel, dim3((unsigned)((tot + kB - 1) / kB)), dim3(kB), 0, ctx().stream, nr, nk, m, M, X, out);
    ELAG_LAUNCH_CHECK();
}

void preload_dense_kernels()
{
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(dense_mm_kernel));
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(dense_mm_mfma_kernel));
}

void dense_times_csr_t(int nc, int n, const double *Bc, const DevCSR &GT, double *X)
{
    ELAG_REQUIRE(GT.nrows == n && GT.ncols == nc, "internal: dense_times_csr_t shape");
    const long long tot = (long long)nc * n;
    if (tot == 0) return;
    hipLaunchKernelGGL(dense_times_csr_t_kernel, dim3((unsigned)((tot + kB - 1) / kB)), dim3(kB), 0, ctx().stream, nc, n, Bc, (const int *)GT.I,
                       (const int *)GT.J, (const double *)GT.a, X);
    ELAG_LAUNCH_CHECK();
}

void dense_level_op(int n, int nc, const DevCSR &V, const double *vw, const double *X, const DevCSR *U, const DevCSR *A, const double *w, double *B)
{
    ELAG_REQUIRE(V.nrows == n && V.ncols == nc && ((U && U->nrows == n) || (A && w && A->nrows == n)), "internal: dense_level_op shape");
    if (n == 0) return;
    const long long tot = (long long)n * n;
    hipLaunchKernelGGL(dense_level_vx_kernel, dim3((unsigned)((tot + kB - 1) / kB)), dim3(kB), 0, ctx().stream, n, (const int *)V.I, (const int *)V.J,
                       (const double *)V.a, vw, X, B);
    ELAG_LAUNCH_CHECK();
    const DevCSR &M = U ? *U : *A;
    hipLaunchKernelGGL(dense_level_u_kernel, dim3((unsigned)((n + kB - 1) / kB)), dim3(kB), 0, ctx().stream, n, (const int *)M.I, (const int *)M.J,
                       (const double *)M.a, U ? nullptr : w, B);
    ELAG_LAUNCH_CHECK();
}

}  // namespace k
}  // namespace elag
