// Setup-side sparse kernels for gfx950: scan, row sort, transpose, two-pass SpGEMM, sparse add.
//
// SpGEMM / add / row-sort share one "expand - sort - compress" (ESC) engine that is deterministic
// (results do not depend on scheduling): a 256-thread workgroup owns a run of consecutive rows whose
// expanded products fit in LDS, wavefronts expand rows (a wave-level exclusive scan over the B-row
// lengths places every product), the workgroup sorts (row,col) keys with an in-LDS bitonic network,
// and run heads sum their equal-key segment.  Pass 1 (symbolic) counts the distinct columns per row;
// a device scan turns counts into row offsets; pass 2 (numeric) repeats the expansion with values and
// writes C.  This is the two-pass structure of hypre's RAP / mfem::Mult mapped onto
// wave64 + LDS instead of a per-thread marker array.
#include "kernels.hpp"

#include <cmath>
#include <mutex>
#include <type_traits>

namespace elag {
namespace k {

namespace {

constexpr int kBlock = 256;
typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------------
// exclusive scan of ints (n up to 2^31), 1024 items per block, recursive on the block sums
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

template <int G>
__device__ __forceinline__ int group_incl_scan(int v)   // inclusive scan inside aligned groups of G lanes
{
    const int lane = threadIdx.x & (G - 1);
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        const int t = __shfl_up(v, o, G);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across the block; returns exclusive prefix, total in *total
__device__ __forceinline__ int block_excl_scan(int v, int *sh /*>= 5 ints*/, int *total)
{
    const int incl = wave_incl_scan(v);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    int off = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) {
        const int s = sh[i];
        if (i < w) off += s;
        tot += s;
    }
    *total = tot;
    return off + incl - v;
}

constexpr int kScanDirectBlocks = 8192;            // up to this many blocks (8 M elements) the final pass adds up the block sums itself

// bmax (optional): the block's largest element, for callers that scan row lengths and want the longest row with it (the final pass's last
// block folds the blocks' maxima: one atomic per wavefront on one word was 1 272 serialised atomics, 12 us, at 324 k rows)
__global__ __launch_bounds__(kBlock) void scan_block_sums(const int *__restrict__ in, int n, int *__restrict__ bsum, int *__restrict__ bmax)
{
    __shared__ int shm[kBlock / 64];
    __shared__ int sh[8];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int s = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (base + q < n) { const int v = in[base + q]; s += v; mx = max(mx, v); }
    int tot;
    block_excl_scan(s, sh, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
    if (bmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / 64; w++) mx = max(mx, shm[w]);
            bmax[blockIdx.x] = mx;
        }
    }
}

// boff: the blocks' offsets (scanned block sums); or bsum: the block sums themselves -- a block then adds up the sums in front of it (a few
// thousand words out of L2: one launch and one scan level less; every setup phase is made of these scans)
__global__ __launch_bounds__(kBlock) void scan_block_final(const int *__restrict__ in, int n, const int *__restrict__ boff, const int *__restrict__ bsum,
                                                           int *__restrict__ out, int write_total, int *__restrict__ maxout = nullptr,
                                                           int *__restrict__ total_copy = nullptr, const int *__restrict__ bmax = nullptr)
{
    __shared__ int sh[8];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { v[q] = (base + q < n) ? in[base + q] : 0; s += v[q]; }
    if (maxout && (!bmax || blockIdx.x == gridDim.x - 1)) {
        // the one-block scan (no sums kernel in front): its own elements; otherwise the last block folds the blocks' maxima
        int mx = 0;
        if (!bmax) mx = max(max(v[0], v[1]), max(v[2], v[3]));
        else for (int i = threadIdx.x; i < (int)gridDim.x; i += kBlock) mx = max(mx, bmax[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
        if ((threadIdx.x & 63) == 0 && mx > 0) atomicMax(maxout, mx);
    }
    int front = boff ? boff[blockIdx.x] : 0;
    if (bsum) {
        int part = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) part += bsum[i];
        block_excl_scan(part, sh, &front);
    }
    int tot;
    int off = block_excl_scan(s, sh, &tot) + front;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (base + q < n) out[base + q] = off;
        off += v[q];
    }
    if (write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) { out[n] = off; if (total_copy) *total_copy = off; }
}

}  // namespace

// d_max (optional, a device word the caller has zeroed): receives the largest element; d_total (optional): a copy of out[n]
void exclusive_scan(const int *in, int *out, int n, int *d_max, int *d_total)
{
    hipStream_t st = ctx().stream;
    if (n == 0) {
        zero_async(out, sizeof(int));
        if (d_total) zero_async(d_total, sizeof(int));
        return;
    }
    const int nb = (n + 1023) / 1024;
    if (nb == 1) {
        hipLaunchKernelGGL(scan_block_final, dim3(1), dim3(kBlock), 0, st, in, n, (const int *)nullptr, (const int *)nullptr, out, 1, d_max, d_total);
    } else if (nb <= kScanDirectBlocks) {
        Temp<int> bsum((size_t)nb), bmax(d_max ? (size_t)nb : 1);
        hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(kBlock), 0, st, in, n, bsum.p, d_max ? bmax.p : (int *)nullptr);
        hipLaunchKernelGGL(scan_block_final, dim3(nb), dim3(kBlock), 0, st, in, n, (const int *)nullptr, (const int *)bsum.p, out, 1, d_max, d_total,
                           d_max ? (const int *)bmax.p : (const int *)nullptr);
    } else {
        Temp<int> bsum((size_t)nb), boff((size_t)nb + 1), bmax(d_max ? (size_t)nb : 1);
        hipLaunchKernelGGL(scan_block_sums, dim3(nb), dim3(kBlock), 0, st, in, n, bsum.p, d_max ? bmax.p : (int *)nullptr);
        exclusive_scan(bsum, boff, nb);
        hipLaunchKernelGGL(scan_block_final, dim3(nb), dim3(kBlock), 0, st, in, n, (const int *)boff.p, (const int *)nullptr, out, 1, d_max, d_total,
                           d_max ? (const int *)bmax.p : (const int *)nullptr);
    }
    ELAG_LAUNCH_CHECK();
}

// 64-bit variant (two quantities scanned at once: a count in the high word, a sum below 2^31 in the low one)
namespace {
typedef long long i64;
__device__ __forceinline__ i64 wave_incl_scan64(i64 v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const i64 t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ i64 block_excl_scan64(i64 v, i64 *sh /*>= 4*/, i64 *total)
{
    const i64 incl = wave_incl_scan64(v);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 63) sh[w] = incl;
    __syncthreads();
    i64 off = 0, tot = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) {
        const i64 s = sh[i];
        if (i < w) off += s;
        tot += s;
    }
    *total = tot;
    return off + incl - v;
}
__global__ __launch_bounds__(kBlock) void scan64_block_sums(const i64 *__restrict__ in, int n, i64 *__restrict__ bsum)
{
    __shared__ i64 sh[4];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    i64 s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (base + q < n) s += in[base + q];
    i64 tot;
    block_excl_scan64(s, sh, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
__global__ __launch_bounds__(kBlock) void scan64_block_final(const i64 *__restrict__ in, int n, const i64 *__restrict__ boff, const i64 *__restrict__ bsum,
                                                             i64 *__restrict__ out)
{
    __shared__ i64 sh[4];
    const int base = blockIdx.x * 1024 + threadIdx.x * 4;
    i64 v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) { v[q] = (base + q < n) ? in[base + q] : 0; s += v[q]; }
    i64 front = boff ? boff[blockIdx.x] : 0;
    if (bsum) {
        i64 part = 0;
        for (int i = threadIdx.x; i < (int)blockIdx.x; i += kBlock) part += bsum[i];
        block_excl_scan64(part, sh, &front);
    }
    i64 tot;
    i64 off = block_excl_scan64(s, sh, &tot) + front;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (base + q < n) out[base + q] = off;
        off += v[q];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == kBlock - 1) out[n] = off;
}
void exclusive_scan64(const i64 *in, i64 *out, int n)
{
    hipStream_t st = ctx().stream;
    if (n == 0) { zero_async(out, sizeof(i64)); return; }
    const int nb = (n + 1023) / 1024;
    if (nb == 1) {
        hipLaunchKernelGGL(scan64_block_final, dim3(1), dim3(kBlock), 0, st, in, n, (const i64 *)nullptr, (const i64 *)nullptr, out);
    } else if (nb <= kScanDirectBlocks) {
        Temp<i64> bsum((size_t)nb);
        hipLaunchKernelGGL(scan64_block_sums, dim3(nb), dim3(kBlock), 0, st, in, n, bsum.p);
        hipLaunchKernelGGL(scan64_block_final, dim3(nb), dim3(kBlock), 0, st, in, n, (const i64 *)nullptr, (const i64 *)bsum.p, out);
    } else {
        Temp<i64> bsum((size_t)nb), boff((size_t)nb + 1);
        hipLaunchKernelGGL(scan64_block_sums, dim3(nb), dim3(kBlock), 0, st, in, n, bsum.p);
        exclusive_scan64(bsum, boff, nb);
        hipLaunchKernelGGL(scan64_block_final, dim3(nb), dim3(kBlock), 0, st, in, n, (const i64 *)boff.p, (const i64 *)nullptr, out);
    }
    ELAG_LAUNCH_CHECK();
}
}  // namespace

// ================================================================================================
// ESC engine
// ================================================================================================
namespace {

constexpr int kCap = 2048;    // products per chunk (by start offset); LDS holds 2*kCap
constexpr int kBigCap = 8192; // single-row workgroups for rows with more products than kCap

struct CsrView {
    const int *I, *J;
    const double *a;
    const int *len = nullptr;      // rows shorter than I[r + 1] - I[r] (a LooseCSR operand); null: plain CSR
    int iscale = 1;                // a loose operand whose rows start at iscale * I[r] (A * ELL(B) left in its layout: W times A's row pointers)
    __device__ int rowlen(int r) const { return len ? len[r] : I[r + 1] - I[r]; }
    __device__ int rowbeg(int r) const { return I[r] * iscale; }
};

// Expanders: ub(row) = number of expanded entries, expand(row) writes them at lds offset `off`.
struct MatmulExpander {
    CsrView A, B;
    static constexpr bool kUnits = true;
    static constexpr bool kFlat = true;
    __device__ int ub_entry(int k) const { return B.rowlen(A.J[k]); }
    // hash path: a row is a list of units (the entries of A's row), each a run of products scale * a[pbeg + q] in column J[pbeg + q]
    __device__ void units(int row, int &ubeg, int &nu) const { ubeg = A.I[row]; nu = A.I[row + 1] - ubeg; }
    __device__ void unit(int row, int ubeg, int u, int &pbeg, int &plen, double &scale, const int *&J, const double *&a) const
    {
        const int j = A.J[ubeg + u];
        scale = A.a[ubeg + u];
        pbeg = B.rowbeg(j); plen = B.rowlen(j);
        J = B.J; a = B.a;
    }
    // cooperative expansion of one row by a group of G lanes (G = 64: a wavefront)
    template <bool NUMERIC, int G = 64>
    __device__ void expand(int row, unsigned rl, int off, u64 *key, double *val) const
    {
        const int lane = threadIdx.x & (G - 1);
        for (int k0 = A.I[row]; k0 < A.I[row + 1]; k0 += G) {
            const int k = k0 + lane;
            int len = 0, j = 0;
            if (k < A.I[row + 1]) { j = A.J[k]; len = B.rowlen(j); }
            const int incl = group_incl_scan<G>(len);
            int pos = off + incl - len;
            if (len) {
                const double av = NUMERIC ? A.a[k] : 0.0;
                const int bs = B.rowbeg(j);
                for (int q = 0; q < len; q++) {
                    key[pos + q] = ((u64)rl << 32) | (unsigned)B.J[bs + q];
                    if (NUMERIC) val[pos + q] = av * B.a[bs + q];
                }
            }
            off += __shfl(incl, G - 1, G);
        }
    }
};

// Expansion of ONE long row by a whole workgroup (the one-row-per-workgroup launches of esc_kernel): the products are dealt to the threads
// by their position in the row's product list -- a scan of the right factor's row lengths over the entries of the left factor's row, then a
// binary search per product -- so every thread fetches products all the time.  (A wavefront expanding the row entry by entry has 29 lanes
// writing ~44 products each one after the other for a row of G1 G0: most of the 700 us of such a launch.)  Same product order as expand().
constexpr int kFlatUnits = 1024;      // entries of the left factor's row the scan holds; longer rows take expand()
template <bool NUMERIC>
__device__ bool matmul_expand_flat(const CsrView &A, const CsrView &B, int row, u64 *key, double *val, int *pre /*kFlatUnits + 1*/)
{
    const int as = A.I[row], na = A.I[row + 1] - as;
    if (na > kFlatUnits) return false;                                 // (uniform over the workgroup)
    for (int t = threadIdx.x; t < na; t += blockDim.x) { const int j = A.J[as + t]; pre[t + 1] = B.rowlen(j); }
    if (threadIdx.x == 0) pre[0] = 0;
    __syncthreads();
    if (threadIdx.x < 64) {                                            // inclusive scan by the first wavefront, 64 entries at a time
        int carry = 0;
        for (int b = 0; b < na; b += 64) {
            const int t = b + (int)threadIdx.x;
            int v = t < na ? pre[t + 1] : 0;
            v = wave_incl_scan(v);
            if (t < na) pre[t + 1] = v + carry;
            carry += __shfl(v, 63, 64);
        }
    }
    __syncthreads();
    const int total = pre[na];
    for (int p = threadIdx.x; p < total; p += blockDim.x) {
        int lo = 0, hi = na;                                           // the entry whose run holds product p: pre[lo] <= p < pre[lo + 1]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= p) lo = mid; else hi = mid; }
        const int j = A.J[as + lo], q = B.rowbeg(j) + (p - pre[lo]);
        key[p] = (u64)(unsigned)B.J[q];
        if (NUMERIC) val[p] = A.a[as + lo] * B.a[q];
    }
    return true;
}

struct AddExpander {
    CsrView A, B;
    double ca, cb;
    static constexpr bool kUnits = true;
    static constexpr bool kFlat = false;
    // two units: A's row scaled by ca, then B's row scaled by cb
    __device__ void units(int row, int &ubeg, int &nu) const { ubeg = 0; nu = 2; }
    __device__ void unit(int row, int, int u, int &pbeg, int &plen, double &scale, const int *&J, const double *&a) const
    {
        const CsrView &M = u == 0 ? A : B;
        scale = u == 0 ? ca : cb;
        pbeg = M.I[row]; plen = M.I[row + 1] - pbeg;
        J = M.J; a = M.a;
    }
    template <bool NUMERIC, int G = 64>
    __device__ void expand(int row, unsigned rl, int off, u64 *key, double *val) const
    {
        const int lane = threadIdx.x & (G - 1);
        const int as = A.I[row], na = A.I[row + 1] - as;
        const int bs = B.I[row], nb = B.I[row + 1] - bs;
        for (int q = lane; q < na; q += G) {
            key[off + q] = ((u64)rl << 32) | (unsigned)A.J[as + q];
            if (NUMERIC) val[off + q] = ca * A.a[as + q];
        }
        for (int q = lane; q < nb; q += G) {
            key[off + na + q] = ((u64)rl << 32) | (unsigned)B.J[bs + q];
            if (NUMERIC) val[off + na + q] = cb * B.a[bs + q];
        }
    }
};

// A * B for a right factor whose rows hold at most W entries (W <= 4: nested low-order interpolators), stored a second time with W
// slots per row (column -1 past the row's end): a unit's product run starts at j * W -- one dependent memory round trip less per row
// than through B's row pointers (row bounds, A's entries, B's row bounds, B's entries: the hash kernel is bound by exactly these trips), and
// the upper bound of a row's products is W times its length, no gather at all.  Hash path only (every class): the padded slots are
// skipped there; expand() exists for the templates that are never launched with this expander.
struct MatmulEllExpander {
    CsrView A;
    const int *EJ;
    const double *Ea;
    int W;
    static constexpr bool kUnits = true;
    static constexpr bool kFlat = false;
    __device__ int ub_entry(int) const { return W; }
    __device__ void units(int row, int &ubeg, int &nu) const { ubeg = A.I[row]; nu = A.I[row + 1] - ubeg; }
    __device__ void unit(int, int ubeg, int u, int &pbeg, int &plen, double &scale, const int *&J, const double *&a) const
    {
        const int j = A.J[ubeg + u];
        scale = A.a[ubeg + u];
        pbeg = j * W; plen = W;
        J = EJ; a = Ea;
    }
    template <bool NUMERIC, int G = 64>
    __device__ void expand(int, unsigned, int, u64 *, double *) const {}
};

struct SortExpander {  // identity: re-emit the row (used to sort columns inside rows)
    CsrView A;
    static constexpr bool kUnits = false;
    static constexpr bool kFlat = false;
    template <bool NUMERIC, int G = 64>
    __device__ void expand(int row, unsigned rl, int off, u64 *key, double *val) const
    {
        const int lane = threadIdx.x & (G - 1);
        const int as = A.I[row], na = A.I[row + 1] - as;
        for (int q = lane; q < na; q += G) {
            key[off + q] = ((u64)rl << 32) | (unsigned)A.J[as + q];
            if (NUMERIC) val[off + q] = A.a[as + q];
        }
    }
};

// The barrier after the compare-exchange stage (size k, stride j) of a bitonic network run by kBlock threads, thread t on the pairs t, t + kBlock,
// ...: a stage of stride <= 64 stays inside the 128 keys that a wavefront's 64 pairs span, so between two such stages a wavefront only waits
// for itself (49 of the 55 stages of a 1 024-key network; 18 us of every 50-100 us chunk of a sparse product went to workgroup barriers here).
__device__ __forceinline__ void bitonic_stage_sync(int k, int j, int n2)
{
    const int next = j > 1 ? (j >> 1) : (k < n2 ? k : 0);        // the next stage's stride (0: the network ends)
    if (j > 64 || next > 64 || next == 0) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <bool HAS_VAL>
__device__ void bitonic_sort_lds(u64 *key, double *val, int n2)
{
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (n2 >> 1); t += kBlock) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = ((i & k) == 0);
                const u64 ki = key[i], kp = key[p];
                if ((ki > kp) == up) {
                    key[i] = kp; key[p] = ki;
                    if (HAS_VAL) { const double t2 = val[i]; val[i] = val[p]; val[p] = t2; }
                }
            }
            bitonic_stage_sync(k, j, n2);
        }
    }
}

__device__ __forceinline__ int lower_bound_dev(const int *a, int n, int v)
{
    int lo = 0, hi = n;  // first index in [0,n] with a[idx] >= v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

// NMAX: LDS capacity in entries.  BIG: one listed row per workgroup; otherwise chunks by start offset.
template <int NMAX, bool BIG, bool NUMERIC, class Expander>
__global__ __launch_bounds__(kBlock) void esc_kernel(int nrows, const int *__restrict__ ubscan /*nrows+1*/,
                                                     const int *__restrict__ ub, const int *__restrict__ biglist,
                                                     Expander ex, int *__restrict__ rownnz,
                                                     const int *__restrict__ CI, int *__restrict__ CJ, double *__restrict__ Ca, int big_hash = 0)
{
    __shared__ u64 key[NMAX];
    __shared__ double val[NUMERIC ? NMAX : 1];
    __shared__ int sh[8];
    __shared__ int s_r0, s_r1;
    int r0, r1, base, count;
    if (BIG) {
        r0 = biglist[blockIdx.x]; r1 = r0 + 1; base = 0; count = ub[r0];
    } else {
        if (threadIdx.x == 0) {
            s_r0 = lower_bound_dev(ubscan, nrows, (int)blockIdx.x * kCap);
            s_r1 = lower_bound_dev(ubscan, nrows, ((int)blockIdx.x + 1) * kCap);
        }
        __syncthreads();
        r0 = s_r0; r1 = s_r1;
        if (r0 >= r1) return;
        base = ubscan[r0]; count = ubscan[r1] - base;
    }
    if (count == 0) return;
    if constexpr (BIG && NUMERIC && Expander::kFlat && NMAX == 8192) {
        // One long row, without sorting its products (up to 8 192 of them: 91 barrier-separated stages over (key, value) pairs -- 0.56-0.74 ms
        // of every hierarchy build for a few hundred such rows): the products are staged 2 048 at a time in their order, ONE wavefront adds them
        // to a hash table of column sums in LDS -- 64 at a time, in ascending lane order (lds_atomics_lane_ordered: the adds of one instruction
        // that meet on a column are applied lowest lane first, the lowest lane wins an empty slot), so a column's sum is the sum of its
        // products in product order whatever the schedule --, then only the distinct columns are sorted.
        const CsrView &A = ex.A, &B = ex.B;
        const int as = A.I[r0], na = A.I[r0 + 1] - as;
        if (big_hash && na <= kFlatUnits) {                                    // (uniform over the workgroup)
            __shared__ int pre[kFlatUnits + 1];
            constexpr int HT = 8192, CH = 2048;
            int *hk = reinterpret_cast<int *>(key);                            // HT slots: column (-1: empty)
            int *pk = hk + HT;                                                 // CH staged columns
            double *pv = reinterpret_cast<double *>(pk + CH);                  // CH staged values
            double *hv = val;                                                  // HT sums
            int *uk = hk + HT;                                                 // afterwards: the distinct columns (over pk / pv and the rest of `key`)
            auto slot_of = [](int col) { return (int)(((unsigned)col * 2654435761u) >> 19); };
            for (int i = threadIdx.x; i < HT; i += kBlock) { hk[i] = -1; hv[i] = 0.0; }
            for (int t = threadIdx.x; t < na; t += kBlock) pre[t + 1] = B.rowlen(A.J[as + t]);
            if (threadIdx.x == 0) pre[0] = 0;
            __syncthreads();
            if (threadIdx.x < 64) {
                int carry = 0;
                for (int b = 0; b < na; b += 64) {
                    const int t = b + (int)threadIdx.x;
                    int v = t < na ? pre[t + 1] : 0;
                    v = wave_incl_scan(v);
                    if (t < na) pre[t + 1] = v + carry;
                    carry += __shfl(v, 63, 64);
                }
            }
            __syncthreads();
            const int total = pre[na];
            for (int c0 = 0; c0 < total; c0 += CH) {
                const int nch = min(CH, total - c0);
                for (int p = threadIdx.x; p < nch; p += kBlock) {
                    const int pp = c0 + p;
                    int lo = 0, hi = na;                                       // the entry whose run holds product pp: pre[lo] <= pp < pre[lo + 1]
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre[mid] <= pp) lo = mid; else hi = mid; }
                    const int j = A.J[as + lo], q = B.rowbeg(j) + (pp - pre[lo]);
                    pk[p] = B.J[q];
                    pv[p] = A.a[as + lo] * B.a[q];
                }
                __syncthreads();
                if (threadIdx.x < 64) {
                    for (int p0 = 0; p0 < nch; p0 += 64) {
                        const int p = p0 + (int)threadIdx.x;
                        const bool on = p < nch;
                        const int col = on ? pk[p] : 0;
                        const double v = on ? pv[p] : 0.0;
                        int h = slot_of(col);
                        if (on)
                            for (;;) {
                                const int prev = atomicCAS(&hk[h], -1, col);
                                if (prev == -1 || prev == col) break;
                                h = (h + 1) & (HT - 1);
                            }
                        __builtin_amdgcn_wave_barrier();
                        if (on) atomicAdd(&hv[h], v);                           // ONE instruction for the 64 products: lanes ascending
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __syncthreads();
            }
            // the distinct columns, compacted (a thread owns HT / kBlock consecutive slots) and sorted
            constexpr int PER = HT / kBlock;
            int cnt = 0;
#pragma unroll 4
            for (int q = 0; q < PER; q++) cnt += hk[threadIdx.x * PER + q] >= 0 ? 1 : 0;
            int nu;
            int off = block_excl_scan(cnt, sh, &nu);
            for (int q = 0; q < PER; q++) {
                const int k = hk[threadIdx.x * PER + q];
                if (k >= 0) uk[off++] = k;
            }
            int npow = 64;
            while (npow < nu) npow <<= 1;
            __syncthreads();
            for (int i = nu + (int)threadIdx.x; i < npow; i += kBlock) uk[i] = 0x7fffffff;
            __syncthreads();
            for (int size = 2; size <= npow; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = threadIdx.x; t < (npow >> 1); t += kBlock) {
                        const int lo = 2 * t - (t & (stride - 1));
                        const int hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const int a0 = uk[lo], b0 = uk[hi];
                        if ((a0 > b0) == up) { uk[lo] = b0; uk[hi] = a0; }
                    }
                    bitonic_stage_sync(size, stride, npow);
                }
            const int cbeg = CI[r0];
            for (int i = threadIdx.x; i < nu; i += kBlock) {
                const int col = uk[i];
                int h = slot_of(col);
                while (hk[h] != col) h = (h + 1) & (HT - 1);
                CJ[cbeg + i] = col;
                Ca[cbeg + i] = hv[h];
            }
            if (rownnz && threadIdx.x == 0) rownnz[r0] = nu;
            return;
        }
    }
    int n2 = 64;
    while (n2 < count) n2 <<= 1;
    // expand: one wavefront per row, round-robin; the single long row of a BIG workgroup by all of its threads where the expander can
    bool flat_done = false;
    if constexpr (BIG && Expander::kFlat) {
        __shared__ int pre[kFlatUnits + 1];
        flat_done = matmul_expand_flat<NUMERIC>(ex.A, ex.B, r0, key, val, pre);
    }
    const int w = threadIdx.x >> 6;
    if (!BIG && r1 - r0 >= 32) {
        // many short rows: sixteen lanes per row, four rows per wavefront at a time (a row costs its group four dependent loads whatever its length)
        for (int row = r0 + ((int)threadIdx.x >> 4); row < r1; row += kBlock / 16) {
            const int off = ubscan[row] - base;
            if (ub[row] > 0) ex.template expand<NUMERIC, 16>(row, (unsigned)(row - r0), off, key, val);
        }
    } else if (!flat_done)
        for (int row = r0 + w; row < r1; row += kBlock / 64) {
            const int off = BIG ? 0 : (ubscan[row] - base);
            if (BIG || ub[row] > 0) ex.template expand<NUMERIC>(row, (unsigned)(row - r0), off, key, val);
        }
    if constexpr (!BIG && NUMERIC && NMAX == 2 * kCap) {
        // A chunk of rows the same way (see the BIG branch above): the chunk's products sit in key / val in product order row after row; one
        // wavefront adds them to a table of (row, column) sums, the distinct (row, column) pairs are sorted, a pair's place in C is its row's
        // offset plus its rank among the row's pairs.
        if (big_hash) {
            constexpr int HT = NMAX;                                            // at most NMAX products, hence at most NMAX distinct pairs
            __shared__ u64 hk[HT];
            __shared__ double hv[HT];
            static_assert(HT == 4096, "the slot is the top 12 bits of the mixed key");
            auto slot_of = [](u64 k) { return (int)((((unsigned)k + (unsigned)(k >> 32) * 0x9E3779B1u) * 2654435761u) >> 20); };
            for (int i = threadIdx.x; i < HT; i += kBlock) { hk[i] = ~0ull; hv[i] = 0.0; }
            __syncthreads();
            if (threadIdx.x < 64) {
                for (int p0 = 0; p0 < count; p0 += 64) {
                    const int p = p0 + (int)threadIdx.x;
                    const u64 kk = p < count ? key[p] : ~0ull;
                    const bool on = kk != ~0ull;                                // (an expander may leave unused places of its upper bound)
                    const double v = on ? val[p] : 0.0;
                    int h = slot_of(kk);
                    if (on)
                        for (;;) {
                            const u64 prev = atomicCAS(reinterpret_cast<unsigned long long *>(&hk[h]), ~0ull, (unsigned long long)kk);
                            if (prev == ~0ull || prev == kk) break;
                            h = (h + 1) & (HT - 1);
                        }
                    __builtin_amdgcn_wave_barrier();
                    if (on) atomicAdd(&hv[h], v);                               // ONE instruction for the 64 products: lanes ascending
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __syncthreads();
            constexpr int PER = HT / kBlock;
            int cnt = 0;
            for (int q = 0; q < PER; q++) cnt += hk[threadIdx.x * PER + q] != ~0ull ? 1 : 0;
            int nu;
            int off = block_excl_scan(cnt, sh, &nu);
            for (int q = 0; q < PER; q++) {
                const u64 k = hk[threadIdx.x * PER + q];
                if (k != ~0ull) key[off++] = k;                                 // (the products are in the table: `key` holds the distinct pairs now)
            }
            int npow = 64;
            while (npow < nu) npow <<= 1;
            __syncthreads();
            for (int i = nu + (int)threadIdx.x; i < npow; i += kBlock) key[i] = ~0ull;
            __syncthreads();
            for (int size = 2; size <= npow; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int t = threadIdx.x; t < (npow >> 1); t += kBlock) {
                        const int lo = 2 * t - (t & (stride - 1));
                        const int hi = lo + stride;
                        const bool up = ((lo & size) == 0);
                        const u64 a0 = key[lo], b0 = key[hi];
                        if ((a0 > b0) == up) { key[lo] = b0; key[hi] = a0; }
                    }
                    bitonic_stage_sync(size, stride, npow);
                }
            for (int i = threadIdx.x; i < nu; i += kBlock) {
                const u64 kk = key[i];
                const unsigned rl = (unsigned)(kk >> 32);
                const u64 rowkey = (u64)rl << 32;
                int lo = 0, hi = i;                                             // first sorted pair of row rl (<= i)
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] < rowkey) lo = mid + 1; else hi = mid; }
                int h = slot_of(kk);
                while (hk[h] != kk) h = (h + 1) & (HT - 1);
                const int pos = CI[r0 + (int)rl] + (i - lo);
                CJ[pos] = (int)(kk & 0xffffffffu);
                Ca[pos] = hv[h];
                // (single pass: CI holds upper-bound offsets, the rows' entry counts are recorded on the way -- by the row's last pair: a row lies
                //  in one chunk, and hundreds of atomics on the three counters of a three-row chunk took 40 us)
                if (rownnz && (i + 1 == nu || (unsigned)(key[i + 1] >> 32) != rl)) rownnz[r0 + (int)rl] = i - lo + 1;
            }
            return;
        }
    }
    for (int i = count + threadIdx.x; i < n2; i += kBlock) key[i] = ~0ull;
    __syncthreads();
    bitonic_sort_lds<NUMERIC>(key, val, n2);
    // heads of equal-key runs; each thread owns a contiguous slice
    const int per = n2 / kBlock > 0 ? n2 / kBlock : 1;
    const int beg = threadIdx.x * per;
    int nheads = 0;
    if (beg < n2)
        for (int i = beg; i < beg + per; i++) {
            const u64 kk = key[i];
            if (kk != ~0ull && (i == 0 || key[i - 1] != kk)) nheads++;
        }
    if (!NUMERIC) {
        // symbolic: distinct columns per row
        if (beg < n2)
            for (int i = beg; i < beg + per; i++) {
                const u64 kk = key[i];
                if (kk != ~0ull && (i == 0 || key[i - 1] != kk)) atomicAdd(&rownnz[r0 + (int)(kk >> 32)], 1);
            }
        return;
    }
    int tot;
    int hidx = block_excl_scan(nheads, sh, &tot);
    if (BIG && rownnz && threadIdx.x == 0) rownnz[r0] = tot;
    // Output position of a head = CI[its row] + its rank among the heads of that row.  (The rows of a chunk need not be
    // adjacent in C: rows handled by the wave-per-row kernel may lie between them.)  rank = hidx - (heads before the
    // row's first sorted entry); the latter from the per-thread head offsets.
    __shared__ int hstart[kBlock];
    hstart[threadIdx.x] = hidx;
    __syncthreads();
    if (beg < n2)
        for (int i = beg; i < beg + per; i++) {
            const u64 kk = key[i];
            if (kk != ~0ull && (i == 0 || key[i - 1] != kk)) {
                double s = val[i];
                for (int q = i + 1; q < n2 && key[q] == kk; q++) s += val[q];
                const unsigned rl = (unsigned)(kk >> 32);
                int lo = 0, hi = i;   // first sorted index of row rl (<= i)
                const u64 rowkey = (u64)rl << 32;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] < rowkey) lo = mid + 1; else hi = mid; }
                const int t0 = lo / per;
                int h0 = hstart[t0];
                for (int q = t0 * per; q < lo; q++) {
                    const u64 kq = key[q];
                    if (kq != ~0ull && (q == 0 || key[q - 1] != kq)) h0++;
                }
                const int pos = CI[r0 + (int)rl] + (hidx - h0);
                CJ[pos] = (int)(kk & 0xffffffffu);
                Ca[pos] = s;
                hidx++;
                // single pass (rownnz given with NUMERIC): CI holds upper-bound offsets, the rows' entry counts are recorded on the way
                if (rownnz && !BIG) atomicAdd(&rownnz[r0 + (int)rl], 1);
            }
        }
}

__global__ void ell_build_kernel(int nrows, int W, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a, int *__restrict__ EJ,
                                 double *__restrict__ Ea)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nrows * W) return;
    const int r = (int)(i / W), q = (int)(i % W);
    const int s = I[r], len = I[r + 1] - s;
    EJ[i] = q < len ? J[s + q] : -1;
    Ea[i] = q < len ? a[s + q] : 0.0;
}

// the whole plan of A * B for a W-slot right factor: a row's products are bounded by W times its length, so the upper-bound layout is W
// times A's row pointers -- no gather, no scan
__global__ void ell_plan_kernel(int nrows, int W, const int *__restrict__ I, int *__restrict__ ub, int *__restrict__ ubw, int *__restrict__ UBI)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row > nrows) return;
    UBI[row] = W * I[row];
    if (row < nrows) { const int u = W * (I[row + 1] - I[row]); ub[row] = u; ubw[row] = u; }
}

__global__ void matmul_ub_kernel(int nrows, CsrView A, CsrView B, int *__restrict__ ub)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    int s = 0;
    for (int k = A.I[row]; k < A.I[row + 1]; k++) s += B.rowlen(A.J[k]);
    ub[row] = s;
}

__global__ void add_ub_kernel(int nrows, CsrView A, CsrView B, int *__restrict__ ub)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    ub[row] = (A.I[row + 1] - A.I[row]) + (B.I[row + 1] - B.I[row]);
}

__global__ void rowlen_kernel(int nrows, const int *__restrict__ I, int *__restrict__ ub)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < nrows) ub[row] = I[row + 1] - I[row];
}

// split ub into chunkable part (ub <= kCap) and a list of big rows
constexpr int kWaveCap = 256;   // rows expanding to at most this many products are handled by one wavefront in registers

__global__ void split_big_kernel(int nrows, const int *__restrict__ ub, int *__restrict__ ub_small, int *__restrict__ biglist,
                                 int *__restrict__ nbig, int *__restrict__ too_big, int wave_cap)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    const int u = ub[row];
    if (u > kCap) {
        ub_small[row] = 0;
        if (u > kBigCap) atomicAdd(too_big, 1);
        else biglist[atomicAdd(nbig, 1)] = row;
    } else ub_small[row] = (u <= wave_cap) ? 0 : u;   // wave rows are not part of the chunked LDS path
}

// ---- wave-per-row path ------------------------------------------------------------------------------------------------
// Most rows of the products on the V-cycle path are short (A P: ~28 products per row, P^T(AP): ~120, fused level
// operators alike).  Sorting them -- in LDS with barriers, or in registers with shuffles -- is what the time goes to
// (measured: both ~10 ms per product at cube456 x5).  Here ONE wavefront handles a row without sorting the products:
//   1. expand the row's products into the wave's LDS slice (storage order = hypre's accumulation order),
//   2. mark first occurrences of every column: all lanes walk the product list together (LDS broadcast reads),
//   3. compact the distinct columns; rank each among them (<= 64R, again broadcast reads) => sorted output position,
//   4. the lane owning an output entry walks the product list once more and adds the matching products in storage
//      order -- deterministic, no atomics, and the same summation order as the reference's marker loop.
// O(products) broadcast steps per row instead of O(products log^2) exchanges.
// Single-pass use (NUMERIC with rownnz != nullptr): CI then holds the rows' offsets in an upper-bound layout (room for every
// product of the row), the distinct entries are written there and their number recorded in rownnz; esc_compact_kernel moves
// them to their final place once the scan of rownnz is known.  This replaces the symbolic launch of the two-pass scheme.
template <int G, int R, bool NUMERIC, class Expander>
__global__ __launch_bounds__(kBlock) void esc_wave_kernel(int nrows, const int *__restrict__ ub, int lo, int hi, Expander ex,
                                                          int *__restrict__ rownnz, const int *__restrict__ CI, int *__restrict__ CJ,
                                                          double *__restrict__ Ca, const int *__restrict__ rowlist = nullptr, int nlist = 0)
{
    // a group of G lanes per row (G = 16 / 32: several rows per wavefront; G = 64: one row), R products per lane.
    // The groups of a wavefront walk their product lists in step (broadcast reads at the same offset of each group's slice):
    // the slices are one entry longer than needed so that they start in different LDS banks (SQ_LDS_BANK_CONFLICT was 40 % of
    // the LDS cycles with 256-byte slices, profiles/r02_rap_pmc.txt).
    constexpr int N = G * R;
    constexpr int NP = (G < 64) ? N + 1 : N;
    constexpr int NG = kBlock / G;
    __shared__ u64 key[NG * NP];
    __shared__ double val[NUMERIC ? NG * NP : 1];
    __shared__ int ucol[NG * NP];
    const int g = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    // rowlist: the rows of this launch's class, gathered beforehand (a launch over all rows spends most of its waves on
    // rows of other classes); without it every row is looked at and the (lo, hi] test decides
    const int gi = blockIdx.x * NG + g;
    const int row = rowlist ? (gi < nlist ? rowlist[gi] : nrows) : gi;
    const int u = row < nrows ? ub[row] : 0;
    const bool active = u > lo && u <= hi;          // (lo, hi]: this launch's class of rows
    u64 *kw = key + g * NP;
    double *vw = NUMERIC ? val + g * NP : val;
    int *uc = ucol + g * NP;
    if (active) ex.template expand<NUMERIC, G>(row, 0u, 0, kw, vw);
    __syncthreads();
    int E = 0;
    if (active) {
        int ck[R];
        bool first[R];
#pragma unroll
        for (int r = 0; r < R; r++) { const int e = r * G + lane; ck[r] = e < u ? (int)(kw[e] & 0xffffffffu) : -1; first[r] = e < u; }
        for (int q = 0; q < u; q++) {
            const int cq = (int)(kw[q] & 0xffffffffu);
#pragma unroll
            for (int r = 0; r < R; r++)
                if (q < r * G + lane && cq == ck[r]) first[r] = false;
        }
        const int gshift = (int)(threadIdx.x & 63) & ~(G - 1);   // first lane of this group inside its wavefront
#pragma unroll
        for (int r = 0; r < R; r++) {
            unsigned long long m = __ballot(first[r]);
            if (G < 64) m = (m >> gshift) & ((1ull << (G & 63)) - 1ull);
            if (first[r]) uc[E + __popcll(m & ((1ull << lane) - 1ull))] = ck[r];
            E += __popcll(m);
        }
        if ((!NUMERIC || rownnz) && lane == 0) rownnz[row] = E;
    }
    if (!NUMERIC) return;
    __syncthreads();                                 // the distinct-column lists are complete
    if (!active) return;
    const int base = CI[row];
    for (int eo = lane; eo < E; eo += G) {
        const int c = uc[eo];
        int rank = 0;
        for (int t = 0; t < E; t++) rank += (uc[t] < c) ? 1 : 0;
        double sum = 0.0;
        for (int q = 0; q < u; q++)
            if ((int)(kw[q] & 0xffffffffu) == c) sum += vw[q];
        CJ[base + rank] = c;
        Ca[base + rank] = sum;
    }
}

// ---- hash path -------------------------------------------------------------------------------------------------------
// The wave path above spends 587 VALU + 285 SALU + 99 LDS instructions per wavefront on 15 memory reads (profiles/r02_rap_pmc.txt): every lane
// compares its products with every product of the row, twice.  Here a group of G lanes owns a row and a small open-addressing table
// in LDS (S slots >= the row's products): the lanes fetch the row's units together (an entry of the left factor's row and the bounds of
// the right factor's row it selects) and R products each, then insert them ONE LANE AFTER THE OTHER, in storage order -- first product
// of a column stored, later ones added: the sums are the wave path's (and the reference's marker loop's) bit for bit, and no atomics
// are involved.  The distinct columns are then compacted, ranked among themselves (E^2 / G comparisons instead of products^2 / G) and
// written in ascending order into the upper-bound layout, like the wave path's single pass.
template <int G, int R, int S, int NU, int BLOCK, class Expander>
__global__ __launch_bounds__(BLOCK) void esc_hash_kernel(int nrows, Expander ex, int *__restrict__ rownnz, const int *__restrict__ CI,
                                                         int *__restrict__ CJ, double *__restrict__ Ca, const int *__restrict__ rowlist, int nlist)
{
    constexpr int NG = BLOCK / G;
    constexpr int NGW = 64 / G;              // lane groups (rows) per wavefront
    constexpr int LOG2S = S == 16 ? 4 : S == 32 ? 5 : S == 64 ? 6 : S == 128 ? 7 : 8;
    static_assert((1 << LOG2S) == S, "table sizes are 16 ... 256");
    __shared__ int key[NG * S];
    __shared__ double val[NG * S];
    const int g = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    const int gi = blockIdx.x * NG + g;
    const bool active = gi < nlist;
    const int row = active ? rowlist[gi] : 0;
    // The tables of a wavefront's NGW rows are INTERLEAVED: slot s of row-group gw lives at s * NGW + gw.  During the serialised inserts one lane
    // per group is active and each goes to a random slot of its own table: with a table per group stored contiguously (g * S + h) the bank is
    // h mod 32 -- random, 58 % of this kernel's LDS cycles were bank conflicts (profiles/r03_rap_pmc_final.txt) and the LDS pipeline was busy 65 %
    // of its time.  Interleaved, the bank of (gw, s) is gw + NGW * (s mod 32 / NGW) for the 4-byte keys and 2 gw + ... for the 8-byte sums:
    // the groups of a half-wavefront never meet on a bank, whatever their slots.
    int *kt = key + (g / NGW) * (NGW * S) + (g % NGW);
    double *vt = val + (g / NGW) * (NGW * S) + (g % NGW);
    for (int s = lane; s < S; s += G) kt[s * NGW] = -1;
    int ubeg = 0, nu = 0;
    if (active) ex.units(row, ubeg, nu);
    const int gshift = (int)(threadIdx.x & 63) & ~(G - 1);          // first lane of this group inside its wavefront
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gshift;
    // NU units per lane and pass, R products per unit: everything a pass needs is requested before the first insert, so a row of up to
    // G * NU units costs four memory round trips (row bounds, units, their product runs' bounds, the products) however long it is --
    // with one unit per lane and pass the A P rows of cube456 x5 (15 units) paid sixteen: 327 us for the product, 2 x that per RAP
    for (int u0 = 0; __ballot(u0 < nu) & gmask; u0 += G * NU) {
        int pbeg[NU], plen[NU];
        double scale[NU];
        const int *J[NU];
        const double *a[NU];
#pragma unroll
        for (int t = 0; t < NU; t++) {
            const int u = u0 + t * G + lane;
            pbeg[t] = 0; plen[t] = 0; scale[t] = 0.0; J[t] = nullptr; a[t] = nullptr;
            if (u < nu) ex.unit(row, ubeg, u, pbeg[t], plen[t], scale[t], J[t], a[t]);
        }
        int pmax = 0;
#pragma unroll
        for (int t = 0; t < NU; t++) pmax = plen[t] > pmax ? plen[t] : pmax;
        for (int q0 = 0; __ballot(q0 < pmax) & gmask; q0 += R) {
            int c[NU][R];
            double v[NU][R];
#pragma unroll
            for (int t = 0; t < NU; t++)
#pragma unroll
                for (int r = 0; r < R; r++) {
                    c[t][r] = -1; v[t][r] = 0.0;
                    if (q0 + r < plen[t]) { c[t][r] = J[t][pbeg[t] + q0 + r]; v[t][r] = a[t][pbeg[t] + q0 + r]; }
                }
            // one lane after the other, unit after unit: the insertion order is the storage order of the products (exactly so when a
            // unit's products fit one pass of R; a longer unit's later products follow the other units' first R -- still one fixed order).
            // The turns are separated by wave barriers: the compiler must not merge or reorder the mutually exclusive bodies (the lanes of a
            // group read and write the same table; a wavefront executes its LDS operations in order).
#pragma unroll
            for (int t = 0; t < NU; t++) {
                for (int p = 0; p < G; p++) {
                    if (lane == p) {
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            if (c[t][r] >= 0) {
                                const double pv = scale[t] * v[t][r];
                                unsigned h = ((unsigned)c[t][r] * 2654435761u) >> (32 - LOG2S);
                                while (true) {
                                    const int kk = kt[h * NGW];
                                    if (kk == c[t][r]) { vt[h * NGW] += pv; break; }
                                    if (kk < 0) { kt[h * NGW] = c[t][r]; vt[h * NGW] = pv; break; }
                                    h = (h + 1) & (S - 1);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    // compact the occupied slots to the front of the table, chunk of G slots by chunk (every lane reads its slot before any writes)
    int E = 0;
    for (int s0 = 0; s0 < S; s0 += G) {
        const int kk = kt[(s0 + lane) * NGW];
        const double vv = vt[(s0 + lane) * NGW];
        unsigned long long m = __ballot(kk >= 0) & gmask;
        m >>= gshift;
        __builtin_amdgcn_wave_barrier();
        if (kk >= 0) { const int e = E + __popcll(m & ((1ull << lane) - 1ull)); kt[e * NGW] = kk; vt[e * NGW] = vv; }
        E += __popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    if (!active) return;
    if (lane == 0) rownnz[row] = E;
    const int base = CI[row];
    for (int i = lane; i < E; i += G) {
        const int ci = kt[i * NGW];
        int rank = 0;
        for (int t = 0; t < E; t++) rank += (kt[t * NGW] < ci) ? 1 : 0;
        CJ[base + rank] = ci;
        Ca[base + rank] = vt[i * NGW];
    }
}

// ---- hash path for a W-slot right factor: ordered parallel inserts, software-pipelined ------------------------------------------
// Where esc_hash_kernel's time goes on A * ELL(B) (cube456 x5, profiles/r04_rap_pmc_*.txt): 840 VALU + 825 SALU + 220 LDS instructions per
// wavefront of 16 rows -- the lanes of a group insert ONE AFTER THE OTHER, so every insert is a wave-wide instruction sequence for 16 useful
// lanes, and a wave64 VALU instruction holds its SIMD for four cycles: the kernel is bound by instruction issue (a first software-pipelined
// version with the same serialised inserts hid the memory latency and was SLOWER, 545 against 420 us: fewer waves, same instructions).
// Here the lanes of a group insert TOGETHER.  Lane l of the group holds slot l % LPU of unit l / LPU of the step at hand, so ascending lanes are
// ascending products in storage order; the key is claimed with an LDS compare-and-swap and the value added with the LDS f64 atomic.  Lanes
// of ONE instruction that meet on an address are served in ascending lane order on gfx950 -- measured (tools/lds_atomic_order_probe.cpp: 504 000
// groups of 2 ... 64 lanes, every one the left-to-right sum) and CHECKED once per process before this kernel is ever chosen
// (lds_atomics_lane_ordered) -- and a wavefront's LDS instructions execute in order, so a column's products are added in storage order:
// the bits of the serial insertion, at one instruction sequence per G products instead of one per product.
// The loads are software-pipelined over the rows a group walks: while row m is inserted, the slots of row m+1, the entries of row m+2, the
// bounds of row m+3 and the number of row m+4 are in flight.
// LPU = min(W, R) lanes per unit: where esc_hash_kernel takes R < W products per unit and turn, a unit's later slots follow every unit's first
// R (pass after pass) -- the same order here, so that every route of a product gives the same bits.
template <int G, int NSP, int W, int R, int S, int BLOCK>
__global__ __launch_bounds__(BLOCK) void ell_hash_kernel(CsrView A, const int *__restrict__ EJ, const double *__restrict__ Ea, int *__restrict__ rownnz,
                                                        int *__restrict__ CJ, double *__restrict__ Ca, const int *__restrict__ rowlist,
                                                        const int *__restrict__ nlist_dev, int nlist_host, int nlists, int list_cap, int lo,
                                                        int *__restrict__ ovf_list, int *__restrict__ ovf_count, int ovf_nlists, int ovf_cap)
{
    // Rows: the list `rowlist` (its length on the device, nlist_dev: the rows an earlier launch passed on) or, without a list, rows 0 .. nlist_host - 1.
    // The launch takes the rows of more than `lo` and at most NSP * UPS entries; a longer row is appended to ovf_list for a wider launch, or only
    // counted, or -- without ovf_count -- left alone like the short ones (another launch over the same rows takes it).
    // Lists come in sets of `nlists` (list l at rowlist + l * list_cap, its length at nlist_dev[32 l]; workgroup b reads list b mod nlists):
    // a sixth of the rows of A at cube456 x5 are too long for the first launch, and that many atomics on ONE counter cost 2.3 ms (5.9 ns each) --
    // a wavefront reserves room for its rows with one atomic, on the counter of the list it is assigned (128 bytes apart: other channels).
    constexpr int NG = BLOCK / G;
    constexpr int LPU = W < R ? W : R;          // lanes per unit
    constexpr int NPASS = W / LPU;              // passes over the units (slots [q * LPU, (q + 1) * LPU) in pass q)
    constexpr int UPS = G / LPU;                // units per step
    constexpr int NS = NSP * NPASS;             // steps per row
    constexpr int MAXU = NSP * UPS;             // entries per row at most
    constexpr int LOG2S = S == 16 ? 4 : S == 32 ? 5 : S == 64 ? 6 : S == 128 ? 7 : 8;
    constexpr int LOG2MIN = G <= 16 ? 4 : 5;    // a row's table: the smallest power of two >= its products, at least 16 slots and a slot per lane
    static_assert((1 << LOG2S) == S, "table sizes are 16 ... 256");
    static_assert(W % LPU == 0 && G % LPU == 0 && W * MAXU <= S, "slots and lanes split evenly over the units; a full row fits the table");
    // the tables of a half-wavefront's groups start G banks apart (a table of 32 k entries per group puts the lanes l of ALL its groups on bank l:
    // every initialisation, compaction and ranking access was a 4-way conflict, 56 % of the LDS cycles -- profiles/r04_rap_pmc_ordered.txt)
    constexpr int TS = S + (G < 32 ? G : 0);
    __shared__ int key[NG * TS];
    __shared__ double val[NG * TS];
    const int g = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    int *kt = key + g * TS;
    double *vt = val + g * TS;
    const int gshift = (int)(threadIdx.x & 63) & ~(G - 1);
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gshift;
    const int lu = lane / LPU, ls = lane % LPU;
    int nlist = nlist_dev ? *nlist_dev : nlist_host;
    long long tg = (long long)gridDim.x * NG;                        // groups of the launch: trip m takes list entries [m * tg, (m + 1) * tg)
    long long first = (long long)blockIdx.x * NG + g;
    if (nlists > 1) {
        const int l = (int)(blockIdx.x % (unsigned)nlists);
        rowlist += (size_t)l * list_cap;
        nlist = nlist_dev[32 * l];
        tg = (long long)(gridDim.x / (unsigned)nlists) * NG;
        first = (long long)(blockIdx.x / (unsigned)nlists) * NG + g;
    }
    const int trips = (int)((nlist + tg - 1) / tg);
    // pipeline registers: stage 0 = row number, stage 1 = row bounds, stage 2 = A's entries, stage 3 = products (already scaled)
    int row0 = -1;
    int row1 = -1, ub1 = 0, nu1 = 0;
    int row2 = -1, ub2 = 0, nu2 = 0, j2[NSP];
    double s2[NSP];
    int row3 = -1, ub3 = 0, nu3 = 0, c3[NS];
    double v3[NS];
#pragma unroll
    for (int t = 0; t < NSP; t++) { j2[t] = -1; s2[t] = 0.0; }
#pragma unroll
    for (int t = 0; t < NS; t++) { c3[t] = -1; v3[t] = 0.0; }
    for (int m = -4; m < trips; m++) {
        // ---- issue this trip's loads (each uses what the previous trip's loads brought) ----
        int row0n = -1;
        {
            const long long li = first + (long long)(m + 4) * tg;
            if (m + 4 < trips && li < nlist) row0n = rowlist ? rowlist[li] : (int)li;
        }
        int ub1n = 0, nu1n = 0;
        if (row0 >= 0) { ub1n = A.I[row0]; nu1n = A.I[row0 + 1] - ub1n; }
        {                                                   // too long for this launch: pass the row on (one atomic per wavefront and trip)
            const bool ovf = nu1 > MAXU;                       // (uniform over the group)
            const unsigned long long om = __ballot(ovf && lane == 0);
            if (om && ovf_count) {
                const int wl = ovf_nlists > 1 ? (int)((blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) % (unsigned)ovf_nlists) : 0;
                const int leader = __ffsll((long long)om) - 1;
                int basek = 0;
                if ((int)(threadIdx.x & 63) == leader) basek = atomicAdd(ovf_count + 32 * wl, __popcll(om));
                basek = __shfl(basek, leader, 64);
                if (ovf && lane == 0 && ovf_list) {
                    const int k = basek + __popcll(om & ((1ull << (threadIdx.x & 63)) - 1ull));
                    if (k < ovf_cap) ovf_list[(size_t)wl * ovf_cap + k] = row1;            // (k >= ovf_cap cannot happen: the caller sized the lists; the count would tell)
                }
            }
            if (ovf) { row1 = -1; nu1 = 0; }
        }
        if (nu1 <= lo) { row1 = -1; nu1 = 0; }              // (another launch's row; an empty row: its count stays the memset's zero)
        int j2n[NSP];
        double s2n[NSP];
#pragma unroll
        for (int t = 0; t < NSP; t++) {
            const int u = t * UPS + lu;
            j2n[t] = -1; s2n[t] = 0.0;
            if (u < nu1) { j2n[t] = A.J[ub1 + u]; s2n[t] = A.a[ub1 + u]; }
        }
        int c3n[NS];
        double v3n[NS];
#pragma unroll
        for (int t = 0; t < NS; t++) {
            const int tu = t % NSP, slot = (t / NSP) * LPU + ls;
            c3n[t] = -1; v3n[t] = 0.0;
            if (j2[tu] >= 0) { c3n[t] = EJ[(unsigned)(j2[tu] * W + slot)]; v3n[t] = Ea[(unsigned)(j2[tu] * W + slot)]; }
        }
        // ---- the row whose products arrived during the previous trip ----
        if (row3 >= 0) {
            int l2 = 32 - __clz(W * nu3 - 1);                // table of this row: 2^l2 >= its products
            l2 = l2 < LOG2MIN ? LOG2MIN : l2;
            const int Srow = 1 << l2;
            for (int s = lane; s < Srow; s += G) { kt[s] = -1; vt[s] = 0.0; }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < NS; t++) {
                if (c3[t] >= 0) {
                    unsigned h = ((unsigned)c3[t] * 2654435761u) >> (32 - l2);
                    while (true) {
                        const int prev = atomicCAS(&kt[h], -1, c3[t]);
                        if (prev == -1 || prev == c3[t]) break;
                        h = (h + 1) & (Srow - 1);
                    }
                    atomicAdd(&vt[h], v3[t]);
                }
                __builtin_amdgcn_wave_barrier();            // (a step's LDS atomics stay in front of the next step's: compiler fence, no instruction)
            }
            int E = 0;
            for (int s0 = 0; s0 < Srow; s0 += G) {
                const int kk = kt[s0 + lane];
                const double vv = vt[s0 + lane];
                unsigned long long mk = __ballot(kk >= 0) & gmask;
                mk >>= gshift;
                __builtin_amdgcn_wave_barrier();
                if (kk >= 0) { const int e = E + __popcll(mk & ((1ull << lane) - 1ull)); kt[e] = kk; vt[e] = vv; }
                E += __popcll(mk);
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) rownnz[row3] = E;
            const unsigned base = (unsigned)ub3 * W;                 // the upper-bound layout of A * ELL(B): W times A's row pointers
            for (int i = lane; i < E; i += G) {
                const int ci = kt[i];
                int rank = 0;
                for (int t = 0; t < E; t++) rank += (kt[t] < ci) ? 1 : 0;
                CJ[base + rank] = ci;
                Ca[base + rank] = vt[i];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- rotate ----
        row3 = row2; ub3 = ub2; nu3 = nu2;
#pragma unroll
        for (int t = 0; t < NS; t++) { c3[t] = c3n[t]; v3[t] = s2[t % NSP] * v3n[t]; }
        row2 = row1; ub2 = ub1; nu2 = nu1;
#pragma unroll
        for (int t = 0; t < NSP; t++) { j2[t] = j2n[t]; s2[t] = s2n[t]; }
        row1 = row0; ub1 = ub1n; nu1 = nu1n;
        row0 = row0n;
    }
}

// one instruction's same-address LDS f64 atomic adds are applied in ascending lane order (and the lowest lane wins a same-slot compare-and-swap):
// what ell_hash_kernel's bits rest on, checked on the device the first time it matters
__global__ void lds_order_probe_kernel(const double *__restrict__ v, const int *__restrict__ slot, double *__restrict__ out, int *__restrict__ winner)
{
    // 256 slots: lane t of the one wavefront adds v[t] to slot[t] and bids for it -- the host chooses the collision patterns (groups of four, all
    // 64 lanes on one slot, random multiplicities, the W-slot interleaving of ell_hash_kernel, slots a whole number of banks apart)
    __shared__ double acc[256];
    __shared__ int key[256];
    const int t = threadIdx.x;
    for (int q = t; q < 256; q += 64) { acc[q] = 0.0; key[q] = -1; }
    __syncthreads();
    const int sl = slot[blockIdx.x * 64 + t];
    atomicAdd(&acc[sl], v[blockIdx.x * 64 + t]);
    (void)atomicCAS(&key[sl], -1, t);
    __syncthreads();
    for (int q = t; q < 256; q += 64) { out[blockIdx.x * 256 + q] = acc[q]; winner[blockIdx.x * 256 + q] = key[q]; }
}

// ---- hash path, unit after unit ---------------------------------------------------------------------------------------
// The same tables, for rows whose units carry LONG product runs (P^T (A P): a unit is a row of A P, 8 products on average; a sum of two
// matrices: a unit is a whole row).  The kernel above gives every lane a unit and lets the lanes insert one after the other -- 135
// serialised inserts for a row of 15 units x 9 products, with 1 / G of the lanes active: 22-25 k cycles per wavefront
// (profiles/r03_rap_pmc.txt).  Here the UNITS take turns and the lanes of the group insert a unit's products together: the columns of one
// unit are distinct (a CSR row), so no two lanes meet in a slot during a turn -- keys are claimed with an LDS compare-and-swap, sums go
// through the LDS f64 atomic so that even a non-canonical row with a repeated column stays a sum -- and a column's products still arrive
// in storage order, unit by unit: the same bits as the serial insertion.  A wavefront executes its LDS operations in order, so the turns
// need no barrier.  Every unit's first RP * G products are requested before the first turn (three memory round trips per G units);
// longer units fetch the rest during their turn.
template <int G, int S, int RP, int BLOCK, class Expander>
__global__ __launch_bounds__(BLOCK) void esc_hashu_kernel(int nrows, Expander ex, int *__restrict__ rownnz, const int *__restrict__ CI,
                                                          int *__restrict__ CJ, double *__restrict__ Ca, const int *__restrict__ rowlist, int nlist)
{
    constexpr int NG = BLOCK / G;
    constexpr int LOG2S = S == 16 ? 4 : S == 32 ? 5 : S == 64 ? 6 : S == 128 ? 7 : 8;
    static_assert((1 << LOG2S) == S, "table sizes are 16 ... 256");
    constexpr int TS = S + (G < 32 ? G : 0);                         // (tables G banks apart: see ell_hash_kernel)
    __shared__ int key[NG * TS];
    __shared__ double val[NG * TS];
    const int g = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    const int gi = blockIdx.x * NG + g;
    const bool active = gi < nlist;
    const int row = active ? rowlist[gi] : 0;
    int *kt = key + g * TS;
    double *vt = val + g * TS;
    for (int s = lane; s < S; s += G) { kt[s] = -1; vt[s] = 0.0; }
    int ubeg = 0, nu = 0;
    if (active) ex.units(row, ubeg, nu);
    const int gshift = (int)(threadIdx.x & 63) & ~(G - 1);          // first lane of this group inside its wavefront
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gshift;
    auto insert = [&](int c, double pv) {
        unsigned h = ((unsigned)c * 2654435761u) >> (32 - LOG2S);
        while (true) {
            const int prev = atomicCAS(&kt[h], -1, c);
            if (prev == -1 || prev == c) break;
            h = (h + 1) & (S - 1);
        }
        atomicAdd(&vt[h], pv);
    };
    for (int u0 = 0; __ballot(u0 < nu) & gmask; u0 += G) {
        int pbeg = 0, plen = 0;
        double scale = 0.0;
        const int *J = nullptr;
        const double *a = nullptr;
        if (u0 + lane < nu) ex.unit(row, ubeg, u0 + lane, pbeg, plen, scale, J, a);
        int c[G][RP];
        double v[G][RP];
#pragma unroll
        for (int t = 0; t < G; t++) {
            const int pb = __shfl(pbeg, t, G), pl = __shfl(plen, t, G);
            const int *Jt = reinterpret_cast<const int *>(__shfl((unsigned long long)J, t, G));
            const double *at = reinterpret_cast<const double *>(__shfl((unsigned long long)a, t, G));
#pragma unroll
            for (int r = 0; r < RP; r++) {
                const int q = lane + r * G;
                c[t][r] = -1; v[t][r] = 0.0;
                if (q < pl) { c[t][r] = Jt[pb + q]; v[t][r] = at[pb + q]; }
            }
        }
#pragma unroll
        for (int t = 0; t < G; t++) {
            const double sc = __shfl(scale, t, G);
            const int pl = __shfl(plen, t, G);
#pragma unroll
            for (int r = 0; r < RP; r++)
                if (c[t][r] >= 0) insert(c[t][r], sc * v[t][r]);
            if (pl > G * RP) {                                      // (uniform over the group)
                const int pb = __shfl(pbeg, t, G);
                const int *Jt = reinterpret_cast<const int *>(__shfl((unsigned long long)J, t, G));
                const double *at = reinterpret_cast<const double *>(__shfl((unsigned long long)a, t, G));
                for (int q = G * RP + lane; q < pl; q += G) insert(Jt[pb + q], sc * at[pb + q]);
            }
            __builtin_amdgcn_wave_barrier();                        // (a turn's LDS atomics stay in front of the next turn's: compiler fence, no instruction)
        }
    }
    // compact the occupied slots to the front of the table, chunk of G slots by chunk (every lane reads its slot before any writes)
    int E = 0;
    for (int s0 = 0; s0 < S; s0 += G) {
        const int kk = kt[s0 + lane];
        const double vv = vt[s0 + lane];
        unsigned long long m = __ballot(kk >= 0) & gmask;
        m >>= gshift;
        __builtin_amdgcn_wave_barrier();
        if (kk >= 0) { const int e = E + __popcll(m & ((1ull << lane) - 1ull)); kt[e] = kk; vt[e] = vv; }
        E += __popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    if (!active) return;
    if (lane == 0) rownnz[row] = E;
    const int base = CI[row];
    for (int i = lane; i < E; i += G) {
        const int ci = kt[i];
        int rank = 0;
        for (int t = 0; t < E; t++) rank += (kt[t] < ci) ? 1 : 0;
        CJ[base + rank] = ci;
        Ca[base + rank] = vt[i];
    }
}

// ---- hash path, unit after unit, without a plan -------------------------------------------------------------------------------
// esc_hashu_kernel for the second product of a triple product whose first stayed in its layout (csr_rap_ell): the rows are addressed directly
// (or through the list an earlier launch passed on, its length on the device), a row's upper bound `ub` and offset `UBI` come from one
// kernel and one scan -- no classes, no lists, no numbers the host would have to wait for.  A row of more than S products is appended to
// ovf_list for the next, wider launch (or only counted); a row's table is the smallest power of two that holds its products.
template <int G, int S, int RP, int BLOCK, class Expander>
__global__ __launch_bounds__(BLOCK) void esc_hashd_kernel(Expander ex, const int *__restrict__ ub, const int *__restrict__ UBI, int *__restrict__ rownnz,
                                                          int *__restrict__ CJ, double *__restrict__ Ca, const int *__restrict__ rowlist,
                                                          const int *__restrict__ nlist_dev, int nlist_host, int *__restrict__ ovf_list, int *__restrict__ ovf_count)
{
    constexpr int NG = BLOCK / G;
    constexpr int LOG2MIN = G <= 16 ? 4 : 5;
    constexpr int TS = S + (G < 32 ? G : 0);                         // (tables G banks apart: see ell_hash_kernel)
    __shared__ int key[NG * TS];
    __shared__ double val[NG * TS];
    const int g = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    int *kt = key + g * TS;
    double *vt = val + g * TS;
    const int gshift = (int)(threadIdx.x & 63) & ~(G - 1);          // first lane of this group inside its wavefront
    const unsigned long long gmask = (G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull)) << gshift;
    const int nlist = nlist_dev ? *nlist_dev : nlist_host;
    for (long long li = (long long)blockIdx.x * NG + g; li < nlist; li += (long long)gridDim.x * NG) {
        const int row = rowlist ? rowlist[li] : (int)li;
        const int u = ub[row];
        if (u == 0) continue;                                            // (its count stays the memset's zero)
        if (u > S) {                                                     // (left to a wider launch -- the caller listed it -- or, there, counted)
            if (ovf_count && lane == 0) { const int k = atomicAdd(ovf_count, 1); if (ovf_list) ovf_list[k] = row; }
            continue;
        }
        int l2 = 32 - __clz(u - 1);
        l2 = l2 < LOG2MIN ? LOG2MIN : l2;
        const int Srow = 1 << l2;
        for (int s = lane; s < Srow; s += G) { kt[s] = -1; vt[s] = 0.0; }
        int ubeg = 0, nu = 0;
        ex.units(row, ubeg, nu);
        auto insert = [&](int c, double pv) {
            unsigned h = ((unsigned)c * 2654435761u) >> (32 - l2);
            while (true) {
                const int prev = atomicCAS(&kt[h], -1, c);
                if (prev == -1 || prev == c) break;
                h = (h + 1) & (Srow - 1);
            }
            atomicAdd(&vt[h], pv);
        };
        for (int u0 = 0; u0 < nu; u0 += G) {
            int pbeg = 0, plen = 0;
            double scale = 0.0;
            const int *J = nullptr;
            const double *a = nullptr;
            if (u0 + lane < nu) ex.unit(row, ubeg, u0 + lane, pbeg, plen, scale, J, a);
            int c[G][RP];
            double v[G][RP];
#pragma unroll
            for (int t = 0; t < G; t++) {
                const int pb = __shfl(pbeg, t, G), pl = __shfl(plen, t, G);
                const int *Jt = reinterpret_cast<const int *>(__shfl((unsigned long long)J, t, G));
                const double *at = reinterpret_cast<const double *>(__shfl((unsigned long long)a, t, G));
#pragma unroll
                for (int r = 0; r < RP; r++) {
                    const int q = lane + r * G;
                    c[t][r] = -1; v[t][r] = 0.0;
                    if (q < pl) { c[t][r] = Jt[pb + q]; v[t][r] = at[pb + q]; }
                }
            }
#pragma unroll
            for (int t = 0; t < G; t++) {
                const double sc = __shfl(scale, t, G);
                const int pl = __shfl(plen, t, G);
#pragma unroll
                for (int r = 0; r < RP; r++)
                    if (c[t][r] >= 0) insert(c[t][r], sc * v[t][r]);
                if (pl > G * RP) {                                      // (uniform over the group)
                    const int pb = __shfl(pbeg, t, G);
                    const int *Jt = reinterpret_cast<const int *>(__shfl((unsigned long long)J, t, G));
                    const double *at = reinterpret_cast<const double *>(__shfl((unsigned long long)a, t, G));
                    for (int q = G * RP + lane; q < pl; q += G) insert(Jt[pb + q], sc * at[pb + q]);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        int E = 0;
        for (int s0 = 0; s0 < Srow; s0 += G) {
            const int kk = kt[s0 + lane];
            const double vv = vt[s0 + lane];
            unsigned long long m = __ballot(kk >= 0) & gmask;
            m >>= gshift;
            __builtin_amdgcn_wave_barrier();
            if (kk >= 0) { const int e = E + __popcll(m & ((1ull << lane) - 1ull)); kt[e] = kk; vt[e] = vv; }
            E += __popcll(m);
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) rownnz[row] = E;
        const int base = UBI[row];
        for (int i = lane; i < E; i += G) {
            const int ci = kt[i];
            int rank = 0;
            for (int t = 0; t < E; t++) rank += (kt[t] < ci) ? 1 : 0;
            CJ[base + rank] = ci;
            Ca[base + rank] = vt[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the fused triple product ---------------------------------------------------------------------------------------------------
// C = R A P in ONE pass, no stored A P: hypre's own formulation of the Galerkin product (hypre_BoomerAMGBuildCoarseOperator, the reference's call
// at src/linalg/solver_ops/ParELAG_Hierarchy.cpp:365: for each coarse row ic, each entry i of R's row, each entry j of A's row i, each entry jc of
// P's row j, accumulate (r * a) * p into C(ic, jc)) for a right factor with at most 4 entries per row (its W-slot copy).  The two-step route
// pays the per-row work of a sparse product -- table, compaction, ranking, output -- once for every FINE row of A P (2.5 M of them at cube456 x5)
// and once more for every coarse row, writes A P in a layout with room for every product and reads it back; this one pays it for the coarse
// rows only (0.33 M) and reads A once per parent of a row.  Where A P is wanted anyway (S = A P - W^-1 P of a bandwidth-bound level) the
// two-step route stays (rap_ell with AP_out).
// One wavefront per coarse row.  The products of the row are numbered through: the entries of A's rows i, child after child in the storage
// order of R's row, W slots each; a step of the wavefront takes 64 / W consecutive entries (lane l: entry l / W, slot l % W), whatever child
// they belong to -- every lane works whatever the children's row lengths.  The child of an entry comes from the children's offsets (held
// one child per lane, exclusive scan of their row lengths): the first entry's by one ballot, the few boundaries inside a step by
// v_readlane.  Inserts are the ordered parallel inserts of ell_hash_kernel (ascending lanes = ascending products: the sums are those of
// the sequential triple loop, bit for bit -- tests/test_gpu_parity.py compares them with the oracle's restatement of hypre's loop).
// Everything is software-pipelined: steps go through  [A's entries] -> [P's slots] -> [insert]  B at a time and one trip apart, across row
// boundaries; rows go through  [R's row bounds] -> [R's entries] -> [A's row bounds]  one row apart.  Two sets of stage registers alternate,
// so that nothing is copied from trip to trip, and which lanes of a step hold a product is a wave-uniform mask (scalar registers).
// A row of more than 64 children, or with more than 64 distinct columns, counts as failed: the caller takes the two-step route then.
template <int B>
struct FusedStageA {       // a batch of B steps whose entries of A are in flight / have arrived
    int row = -1;
    bool first = false, last = false;
    unsigned long long valid[B];    // lanes of the step that hold a product (wave-uniform: scalar registers)
    int j[B];
    double av[B], rv[B];
};
template <int B>
struct FusedStageB {       // ... whose slots of P are in flight / have arrived
    int row = -1;
    bool first = false, last = false;
    unsigned long long valid[B];
    int c[B];
    double pv[B], sc[B];
};

template <int W, int B, int S, int BLOCK>
__global__ __launch_bounds__(BLOCK) void rap_fused_kernel(CsrView R, CsrView A, const int *__restrict__ EJ, const double *__restrict__ Ea, int nrows,
                                                         int *__restrict__ rownnz, int *__restrict__ CJ, double *__restrict__ Ca, int out_stride,
                                                         int *__restrict__ fail_count)
{
    constexpr int EPS = 64 / W;                  // entries of A per step
    constexpr int NW = BLOCK / 64;
    constexpr int LOG2S = S == 64 ? 6 : S == 128 ? 7 : 8;
    static_assert((1 << LOG2S) == S, "table sizes are 64 ... 256");
    __shared__ int key[NW * S];
    __shared__ double val[NW * S];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;     // (the wavefront's number is uniform: rows, bounds and flags stay scalar)
    int *kt = key + wave * S;
    double *vt = val + wave * S;
    const int le = lane / W, ls = lane % W;
    const int tw = (int)gridDim.x * NW, wid = (int)blockIdx.x * NW + wave;
    // ---- row pipeline: q0 = row bounds in flight, q1 = children in flight, q2 = children's bounds in A in flight, g = the row being generated ----
    int m_next = 0;
    int q0_row = -1, q0_kb = 0, q0_ke = 0;
    int q1_row = -1, q1_nc = 0, q1_ci = 0;
    double q1_rv = 0.0;
    int q2_row = -1, q2_nc = 0, q2_ab = 0, q2_ae = 0;
    double q2_rv = 0.0;
    int g_row = -1, g_nc = 0, g_T = 0, g_ns = 0, g_s = 0, g_pre = 0, g_d = 0;      // g_d: a child's first entry in A minus its offset in the row's numbering
    double g_rv = 0.0;
    int claimed = 0;
    bool failed = false;
    bool more_rows = true;
    // One trip: generate a batch (entries of A requested) into `na`, request the slots of P for the batch in `sa` into `nb`, insert the batch in `sb`.
    auto trip = [&](FusedStageA<B> &sa, FusedStageB<B> &sb, FusedStageA<B> &na, FusedStageB<B> &nb) -> bool {
        // ================= generator: the next B steps of the row at hand, or the next row =================
        if (g_s >= g_ns) {
            // ---- the row whose children's bounds arrived becomes the row at hand ----
            g_row = q2_row; g_nc = q2_nc;
            const int al = (lane < g_nc) ? q2_ae - q2_ab : 0;
            g_rv = q2_rv;
            int incl = al;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
            g_pre = incl - al;
            g_d = q2_ab - g_pre;
            g_T = __builtin_amdgcn_readlane(incl, 63);
            g_ns = (g_T + EPS - 1) / EPS;
            g_s = 0;
            // ---- the children's bounds in A of the row whose children arrived ----
            q2_row = q1_row; q2_nc = q1_nc; q2_rv = q1_rv; q2_ab = 0; q2_ae = 0;
            if (lane < q1_nc) { q2_ab = A.I[q1_ci]; q2_ae = A.I[q1_ci + 1]; }
            // ---- the children of the row whose bounds arrived ----
            q1_row = q0_row; q1_ci = 0; q1_rv = 0.0;
            {
                const int kb = __builtin_amdgcn_readfirstlane(q0_kb), ke = __builtin_amdgcn_readfirstlane(q0_ke);
                int nc = ke - kb;
                if (nc > 64) {                                  // more children than lanes: not this kernel's row
                    if (lane == 0) atomicAdd(fail_count, 1);
                    nc = 0;
                }
                q1_nc = nc;
                if (lane < nc) { q1_ci = R.J[kb + lane]; q1_rv = R.a[kb + lane]; }
            }
            // ---- the bounds of the next row of this wavefront ----
            {
                const int row = wid + m_next * tw;
                m_next++;
                q0_row = -1; q0_kb = 0; q0_ke = 0;
                if (more_rows && row < nrows) { q0_row = row; q0_kb = R.I[row]; q0_ke = R.I[row + 1]; }
                else more_rows = false;
            }
        }
        na.row = -1; na.first = false; na.last = false;
        if (g_s < g_ns) {
            na.row = g_row; na.first = g_s == 0; na.last = g_s + B >= g_ns;
#pragma unroll
            for (int b = 0; b < B; b++) {
                const int st = g_s + b;
                na.valid[b] = 0ull;
                if (st < g_ns) {                                // (uniform)
                    const int e_first = st * EPS, e_last = min(e_first + EPS, g_T) - 1, e = e_first + le;
                    // the child of the step's first entry: the last one whose offset is <= it; the boundaries inside the step one by one
                    const int c_lo = __popcll(__ballot(lane < g_nc && g_pre <= e_first)) - 1;
                    int c = c_lo;
                    for (int cn = c_lo + 1; cn < g_nc; cn++) {
                        const int pn = __builtin_amdgcn_readlane(g_pre, cn);
                        if (pn > e_last) break;
                        c += (e >= pn) ? 1 : 0;
                    }
                    const bool ok = e <= e_last;
                    na.valid[b] = __ballot(ok);
                    // (every lane loads: the lanes past the row's end read the step's first entry again and are masked out of the inserts)
                    const int cc = ok ? c : c_lo;
                    const int q = __shfl(g_d, cc, 64) + (ok ? e : e_first);
                    na.rv[b] = __shfl(g_rv, cc, 64);
                    na.j[b] = A.J[q]; na.av[b] = A.a[q];
                }
            }
            g_s += B;
        }
        // ================= P's slots of the steps whose entries of A arrived =================
        nb.row = sa.row; nb.first = sa.first; nb.last = sa.last;
        if (sa.row >= 0) {
#pragma unroll
            for (int b = 0; b < B; b++) {
                nb.valid[b] = sa.valid[b];
                if (sa.valid[b]) {                              // (uniform)
                    nb.sc[b] = sa.rv[b] * sa.av[b];             // (r * a first, then * p: hypre's order)
                    nb.c[b] = EJ[(unsigned)(sa.j[b] * W + ls)]; nb.pv[b] = Ea[(unsigned)(sa.j[b] * W + ls)];
                }
            }
        }
        // ================= inserts of the steps whose slots arrived =================
        if (sb.row >= 0) {
            if (sb.first) {
                for (int s = lane; s < S; s += 64) { kt[s] = -1; vt[s] = 0.0; }
                claimed = 0; failed = false;
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int b = 0; b < B; b++) {
                if (sb.valid[b]) {                              // (uniform)
                    if (!failed && claimed > S - 64) failed = true;    // (the table could fill up during this step: more distinct columns than this kernel takes)
                    if (!failed) {
                        bool fresh = false;
                        if (((sb.valid[b] >> lane) & 1ull) && sb.c[b] >= 0) {
                            unsigned h = ((unsigned)sb.c[b] * 2654435761u) >> (32 - LOG2S);
                            while (true) {
                                const int prev = atomicCAS(&kt[h], -1, sb.c[b]);
                                if (prev == -1) { fresh = true; break; }
                                if (prev == sb.c[b]) break;
                                h = (h + 1) & (S - 1);
                            }
                            atomicAdd(&vt[h], sb.sc[b] * sb.pv[b]);
                        }
                        claimed += __popcll(__ballot(fresh));
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (sb.last) {
                int E = 0;
                if (!failed) {
                    for (int s0 = 0; s0 < S; s0 += 64) {
                        const int kk = kt[s0 + lane];
                        const double vv = vt[s0 + lane];
                        const unsigned long long mk = __ballot(kk >= 0);
                        __builtin_amdgcn_wave_barrier();
                        if (kk >= 0) { const int e = E + __popcll(mk & ((1ull << lane) - 1ull)); kt[e] = kk; vt[e] = vv; }
                        E += __popcll(mk);
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (E > out_stride) failed = true;
                }
                if (failed) {
                    if (lane == 0) atomicAdd(fail_count, 1);
                } else {
                    if (lane == 0) rownnz[sb.row] = E;
                    const long long base = (long long)sb.row * out_stride;
                    for (int i = lane; i < E; i += 64) {
                        const int ci = kt[i];
                        int rank = 0;
                        for (int t = 0; t < E; t++) rank += (kt[t] < ci) ? 1 : 0;
                        CJ[base + rank] = ci;
                        Ca[base + rank] = vt[i];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return !more_rows && g_s >= g_ns && q0_row < 0 && q1_row < 0 && q2_row < 0 && na.row < 0 && nb.row < 0;
    };
    FusedStageA<B> a0, a1;
    FusedStageB<B> b0, b1;
    while (true) {
        if (trip(a0, b0, a1, b1)) break;
        if (trip(a1, b1, a0, b0)) break;
    }
}

// upper bound of a row's products, 8 lanes per row (matmul_ub_kernel walks a row with one thread: 27 us for the 325 k rows of P^T at cube456 x5),
// packed for ONE scan of two quantities: the bound in the low word, in the high word 1 for a row beyond `wide` products (the rows the first
// launch of esc_hashd_kernel leaves to the second)
__global__ __launch_bounds__(kBlock) void matmul_ub8_kernel(int nrows, CsrView A, CsrView B, int wide, int *__restrict__ ub, long long *__restrict__ packed)
{
    const int row = (int)((blockIdx.x * (long long)kBlock + threadIdx.x) >> 3);
    const int lane = threadIdx.x & 7;
    int s = 0;
    if (row < nrows)
        for (int k = A.I[row] + lane; k < A.I[row + 1]; k += 8) s += B.rowlen(A.J[k]);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 8);
    if (row < nrows && lane == 0) { ub[row] = s; packed[row] = ((long long)(s > wide ? 1 : 0) << 32) | (long long)s; }
}

// the scanned pairs apart: UBI[r] = offset of row r in the upper-bound layout; the rows beyond `wide` products, in ascending order, and their number
__global__ void rap_unpack_kernel(int nrows, const long long *__restrict__ scanned, const int *__restrict__ ub, int wide, int *__restrict__ UBI, int *__restrict__ wide_list,
                                  int *__restrict__ wide_count)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row > nrows) return;
    const long long v = scanned[row];
    UBI[row] = (int)(v & 0xffffffffll);
    if (row == nrows) { *wide_count = (int)(v >> 32); return; }
    if (ub[row] > wide) wide_list[(int)(v >> 32)] = row;
}

// every number the host reads after a csr_rap_ell, in one block: [0..4] the launches' counters as they are, [5] nnz(C), [6] longest row of C,
// [7] nnz(A P), [8] longest row of A P
__global__ void rap_gather_kernel(int *__restrict__ ctl, const int *__restrict__ nnzC, const int *__restrict__ longC, const int *__restrict__ nnzAP, const int *__restrict__ longAP)
{
    if (threadIdx.x == 0) { ctl[5] = nnzC ? *nnzC : 0; ctl[6] = longC ? *longC : 0; ctl[7] = nnzAP ? *nnzAP : 0; ctl[8] = longAP ? *longAP : 0; }
}

// ub of the rows the wave path handles (others 0): its scan gives the rows' offsets in the upper-bound layout
__global__ void wave_ub_kernel(int nrows, const int *__restrict__ ub, int wave_cap, int *__restrict__ ubw, unsigned long long *__restrict__ total, int all_rows)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    int v = 0;
    if (row < nrows) { const int u = ub[row]; v = (u > 0 && (u <= wave_cap || all_rows)) ? u : 0; ubw[row] = v; }
    // 64-bit total (the int scan of ubw is only used when this fits); only asked for when n * wave_cap could pass 2^31
    if (total) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(total, (unsigned long long)v);
    }
}

// rows written in the upper-bound layout -> their final place (8 lanes per row; rows of the chunked path have ubw == 0)
__global__ __launch_bounds__(kBlock) void esc_compact_kernel(int nrows, const int *__restrict__ ubw, const int *__restrict__ UBI,
                                                             const int *__restrict__ CI, const int *__restrict__ Ju, const double *__restrict__ au,
                                                             int *__restrict__ CJ, double *__restrict__ Ca, int src_scale = 1)
{
    const int row = (int)((blockIdx.x * (long long)kBlock + threadIdx.x) >> 3);
    const int lane = threadIdx.x & 7;
    if (row >= nrows || (ubw && ubw[row] == 0)) return;
    const int src = (UBI ? UBI[row] : row) * src_scale, dst = CI[row], cnt = CI[row + 1] - dst;
    for (int e = lane; e < cnt; e += 8) { CJ[dst + e] = Ju[src + e]; Ca[dst + e] = au[src + e]; }
}

// rows of the wave path sorted into their classes by number of products: class c holds the rows with bounds[c] < ub <= bounds[c+1].
// A workgroup reserves one range per class (one atomic each) and writes its rows there in ascending order.
constexpr int kNumClasses = 5;
struct ClassBounds { int b[kNumClasses + 1]; };

__global__ __launch_bounds__(kBlock) void esc_classify_kernel(int nrows, const int *__restrict__ ub, ClassBounds cb, int *__restrict__ counters /*kNumClasses*/,
                                                              int *__restrict__ lists /*kNumClasses x nrows*/, int fill)
{
    // A workgroup owns a contiguous span of rows: it counts its rows per class, reserves one range per class (ONE atomic each for the whole
    // span -- with a reservation per 256 rows the 50 000 atomics on five addresses were 100 of this kernel's 116 us at cube456 x5) and
    // writes its rows there in ascending order.
    __shared__ int wcount[kNumClasses][kBlock / 64];
    __shared__ int base[kNumClasses];
    __shared__ int running[kNumClasses];
    const int span = (int)((((long long)nrows + gridDim.x - 1) / gridDim.x + kBlock - 1) / kBlock) * kBlock;
    const int r0 = (int)blockIdx.x * span, r1 = min(nrows, r0 + span);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    auto cls_of = [&](int row) {
        const int u = row < r1 ? ub[row] : 0;
        int cls = -1;
#pragma unroll
        for (int c = 0; c < kNumClasses; c++)
            if (u > cb.b[c] && u <= cb.b[c + 1]) cls = c;
        return cls;
    };
    int mine[kNumClasses] = {0, 0, 0, 0, 0};
    for (int c0 = r0; c0 < r1; c0 += kBlock) {
        const int cls = cls_of(c0 + (int)threadIdx.x);
#pragma unroll
        for (int c = 0; c < kNumClasses; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (lane == 0) mine[c] += __popcll(m);
        }
    }
    if (lane == 0)
        for (int c = 0; c < kNumClasses; c++) wcount[c][w] = mine[c];
    __syncthreads();
    if (threadIdx.x < kNumClasses) {
        int tot = 0;
        for (int i = 0; i < kBlock / 64; i++) tot += wcount[threadIdx.x][i];
        base[threadIdx.x] = tot ? atomicAdd(&counters[threadIdx.x], tot) : 0;
        running[threadIdx.x] = 0;
    }
    __syncthreads();
    if (!fill) return;
    for (int c0 = r0; c0 < r1; c0 += kBlock) {
        const int row = c0 + (int)threadIdx.x;
        const int cls = cls_of(row);
        int before = 0;
#pragma unroll
        for (int c = 0; c < kNumClasses; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcount[c][w] = __popcll(m);
        }
        __syncthreads();
        if (cls >= 0) {
            int off = base[cls] + running[cls] + before;
            for (int i = 0; i < w; i++) off += wcount[cls][i];
            lists[(size_t)cls * nrows + off] = row;
        }
        __syncthreads();
        if (threadIdx.x < kNumClasses) {
            int tot = 0;
            for (int i = 0; i < kBlock / 64; i++) tot += wcount[threadIdx.x][i];
            running[threadIdx.x] += tot;
        }
        __syncthreads();
    }
}

inline bool single_pass_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_ESC_SINGLE_PASS"); v = e ? atoi(e) : 1; }   // 0 = two-pass (A/B runs)
    return v != 0;
}

__global__ void esc_gather_kernel(int *__restrict__ counters, const int *__restrict__ tot_small, const int *__restrict__ tot_wave, const int *__restrict__ ccount,
                                  const int *__restrict__ extra, int extra_n)
{
    const int t = threadIdx.x;
    if (t == 0) counters[2] = *tot_small;
    if (t == 1) counters[3] = tot_wave ? *tot_wave : 0;
    if (t >= 2 && t < 2 + kNumClasses) counters[2 + t] = ccount ? ccount[t - 2] : 0;
    // a caller's array, sampled: the sum of 64 entries spread evenly over its extra_n (the mean row length of a product left in the
    // upper-bound layout -- all the next product wants from it is which insertion scheme to take)
    int v = (extra && extra_n > 0) ? extra[(long long)t * extra_n / 64] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (t == 0) counters[4 + kNumClasses] = v;
}

inline int hash_classes()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_ESC_HASH"); v = e ? atoi(e) : 31; }     // bit mask of wave-path classes (A/B runs); 0 = wave path only
    return v;
}

inline bool hashu_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_ESC_HASHU"); v = e ? atoi(e) : 1; }   // 0 = serial insertion only (A/B runs)
    return v != 0;
}

bool lds_atomics_lane_ordered()
{
    static int ordered = -1;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (ordered >= 0) return ordered == 1;
    // The patterns the product kernels produce, not only the easy one: groups of 4 lanes on 16 slots; all 64 lanes on ONE slot;
    // 2 ... 33 slots with random multiplicities; lane l on slot l mod W (the W-slot interleaving, W = 2 ... 32: every slot met by 64 / W lanes
    // that are W apart); colliding slots 32 / 64 words apart (all of them in the same banks); runs of random length.  Values of very
    // different magnitude, so that any other order of a slot's additions shows in the last bits.
    constexpr int NB = 384;
    std::vector<double> v((size_t)NB * 64), out((size_t)NB * 256);
    std::vector<int> slot((size_t)NB * 64), win((size_t)NB * 256);
    unsigned long long x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto &d : v) { const unsigned long long r = rnd(); d = std::ldexp((double)(r % 2000001ull) / 1e6 - 1.0, (int)((r >> 32) % 60) - 30); }
    for (int b = 0; b < NB; b++) {
        int *sl = &slot[(size_t)b * 64];
        const int kind = b % 6;
        if (kind == 0) for (int t = 0; t < 64; t++) sl[t] = t >> 2;
        else if (kind == 1) { const int one = (int)(rnd() % 256); for (int t = 0; t < 64; t++) sl[t] = one; }
        else if (kind == 2) { static const int ks[] = {2, 3, 5, 8, 16, 33}; const int K = ks[(b / 6) % 6]; for (int t = 0; t < 64; t++) sl[t] = (int)(rnd() % (unsigned)K) * 7 % 256; }
        else if (kind == 3) { const int W = 2 << ((b / 6) % 5); for (int t = 0; t < 64; t++) sl[t] = t % W; }
        else if (kind == 4) { const int K = 2 + (b / 6) % 7, step = (b / 6) % 2 ? 64 : 32; for (int t = 0; t < 64; t++) sl[t] = ((int)(rnd() % (unsigned)K) * step) % 256; }
        else { int t = 0, cur = (int)(rnd() % 256); while (t < 64) { int len = 1 + (int)(rnd() % 9); while (len-- && t < 64) sl[t++] = cur; cur = (int)(rnd() % 256); } }
    }
    double *dv = nullptr, *dout = nullptr;
    int *dw = nullptr, *ds = nullptr;
    hipStream_t st = ctx().stream;
    dv = dev_alloc<double>(v.size()); dout = dev_alloc<double>(out.size()); dw = dev_alloc<int>(win.size()); ds = dev_alloc<int>(slot.size());
    stage_h2d(dv, v.data(), sizeof(double) * v.size());
    stage_h2d(ds, slot.data(), sizeof(int) * slot.size());
    hipLaunchKernelGGL(lds_order_probe_kernel, dim3(NB), dim3(64), 0, st, (const double *)dv, (const int *)ds, dout, dw);
    stage_d2h(out.data(), dout, sizeof(double) * out.size());
    stage_d2h(win.data(), dw, sizeof(int) * win.size());
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    dev_free(dv); dev_free(dout); dev_free(dw); dev_free(ds);
    bool ok = true;
    for (int b = 0; b < NB && ok; b++) {
        double sum[256];
        int first[256];
        for (int q = 0; q < 256; q++) { sum[q] = 0.0; first[q] = -1; }
        for (int t = 0; t < 64; t++) {                       // ascending lanes: the sequential loop
            const int q = slot[(size_t)b * 64 + t];
            sum[q] += v[(size_t)b * 64 + t];
            if (first[q] < 0) first[q] = t;
        }
        for (int q = 0; q < 256; q++)
            if (memcmp(&sum[q], &out[(size_t)b * 256 + q], sizeof(double)) != 0 || win[(size_t)b * 256 + q] != first[q]) { ok = false; break; }
    }
    ordered = ok ? 1 : 0;
    if (!ok && !getenv("ELAG_HIP_QUIET"))
        fprintf(stderr, "[elag_hip] this device does not apply the LDS atomics of one instruction in lane order: sparse products keep the serialised inserts\n");
    return ok;
}

// long single rows of a product through the hash table of column sums (esc_kernel, BIG): needs the lane order of the LDS atomics
inline int esc_big_hash()
{
    const char *e = getenv("ELAG_HIP_BIG_HASH");       // (read at every product: A/B runs and a test switch it)
    return (!(e && !atoi(e)) && lds_atomics_lane_ordered()) ? 1 : 0;
}

inline bool ell_pipe_enabled()
{
    const char *e = getenv("ELAG_HIP_ELL_PIPE");       // (read at every product: A/B runs and a test switch it); 0 = esc_hash_kernel for the W-slot right factor too
    return !(e && !atoi(e)) && lds_atomics_lane_ordered();
}

// slots per row of the W-slot copy of a right factor with rows of at most `max_row` entries: 2 or 4 (the widths ell_hash_kernel is built for)
inline int ell_width(int max_row) { return max_row <= 2 ? 2 : 4; }

inline bool wave_path_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_ESC_WAVE"); v = e ? atoi(e) : 1; }   // 0 = chunked LDS path only (A/B runs)
    return v != 0;
}

inline unsigned grid1(long long n, int b = kBlock) { return (unsigned)((n + b - 1) / b < 1 ? 1 : (n + b - 1) / b); }

struct EscPlan {
    int nrows;
    Temp<int> ub, ub_small, ubscan, biglist, counters;
    // single-pass wave path: per-row upper bounds of the wave rows, their scan, and the class row lists -- launched in finish() with the
    // rest of the plan, so that ONE drain returns every number the host needs (three drains per sparse product in round 2)
    Temp<int> ubw, UBI, ccount, lists;
    int total_small = 0, nbig = 0;
    int wave_cap = 0;   // > 0: rows with at most this many products go to esc_wave_kernel
    bool wave_ready = false;          // ubw / UBI / class lists are in place and total_ub / hc are known
    bool all_rows = false;            // ... and the upper-bound layout has room for EVERY row (chunked and one-row-per-workgroup rows too): no symbolic launches at all
    unsigned long long total_ub = 0;
    int hc[kNumClasses] = {0, 0, 0, 0, 0};
    const int *extra = nullptr;       // an array of the caller's (extra_n entries), sampled at 64 places with the plan's own read-back (extra_value = their sum)
    int extra_n = 0, extra_value = 0;
    double unit_run = 0.0;            // products per unit on average (a product: the right factor's entries per row): picks the insertion scheme of the hash path
    int max_unit = 0;                 // ... and at most, where the right factor knows its longest row (0: unknown)
    explicit EscPlan(int n)
        : nrows(n), ub((size_t)n), ub_small((size_t)n), ubscan((size_t)n + 1), biglist((size_t)n), counters(5 + kNumClasses), ubw((size_t)n),
          UBI((size_t)n + 1), ccount(kNumClasses), lists((size_t)kNumClasses * (size_t)n) {}
    void finish()
    {
        hipStream_t st = ctx().stream;
        zero_async(counters, 2 * sizeof(int));
        hipLaunchKernelGGL(split_big_kernel, dim3(grid1(nrows)), dim3(kBlock), 0, st, nrows, ub.p, ub_small.p, biglist.p,
                           counters.p, counters.p + 1, wave_cap);
        exclusive_scan(ub_small, ubscan, nrows);
        // an upper-bound layout indexed with ints: only when nrows * wave_cap cannot overflow them (the caller's fallback handles the rest)
        const bool wave = wave_cap > 0 && single_pass_enabled() && (unsigned long long)nrows * (unsigned long long)wave_cap < (1ull << 31);
        if (wave) {
            all_rows = (unsigned long long)nrows * (unsigned long long)kBigCap < (1ull << 31);        // (every row's products fit the int layout whatever their class)
            hipLaunchKernelGGL(wave_ub_kernel, dim3(grid1(nrows)), dim3(kBlock), 0, st, nrows, (const int *)ub.p, wave_cap, ubw.p, (unsigned long long *)nullptr, all_rows ? 1 : 0);
            exclusive_scan(ubw, UBI, nrows);
            const ClassBounds cb{{0, 16, 32, 64, 128, kWaveCap}};
            zero_async(ccount, sizeof(int) * kNumClasses);
            hipLaunchKernelGGL(esc_classify_kernel, dim3(std::min(grid1(nrows), 1024u)), dim3(kBlock), 0, st, nrows, (const int *)ub.p, cb, ccount.p, lists.p, 1);
            ELAG_LAUNCH_CHECK();
        }
        // every number the host needs, gathered into one block and read back by ONE copy into pinned memory (four small copies into
        // pageable variables cost ~25 us of host time each: profiles/r03_rap_timeline.txt)
        hipLaunchKernelGGL(esc_gather_kernel, dim3(1), dim3(64), 0, st, counters.p, (const int *)(ubscan.p + nrows), wave ? (const int *)(UBI.p + nrows) : (const int *)nullptr,
                           wave ? (const int *)ccount.p : (const int *)nullptr, extra, extra_n);
        int *hp = reinterpret_cast<int *>(ctx().h_scalar + 10);          // (doubles 0 and 8 of the pinned block belong to the reductions)
        fetch_ints(counters.p, 5 + kNumClasses, hp);
        const int h[2] = {hp[0], hp[1]}, tot = hp[2], totw = hp[3];
        for (int c = 0; c < kNumClasses; c++) hc[c] = hp[4 + c];
        extra_value = hp[4 + kNumClasses];
        if (h[1] > 0)
            fail(ELAG_HIP_ERR_ARG, "sparse product: %d rows expand to more than %d intermediate products (unsupported)", h[1], kBigCap);
        nbig = h[0];
        total_small = tot;
        if (getenv("ELAG_HIP_ESC_VERBOSE"))
            fprintf(stderr, "[elag_hip] sparse product plan: %d rows, %d one-row workgroups (> %d products), %d products in %d chunks, wave path %llu\n", nrows, nbig, kCap,
                    total_small, nchunks(), wave ? (unsigned long long)totw : 0ull);
        if (wave) { wave_ready = true; total_ub = (unsigned long long)totw; }
    }
    // The plan of A * B when B has W slots per row (MatmulEllExpander) and no row of A is longer than kWaveCap / W: every row is a
    // hash-path row, the layout is W * A.I, and the only number the host waits for is how many rows each class has.
    void finish_ell(const int *AI, int W, long long nnzA)
    {
        hipStream_t st = ctx().stream;
        hipLaunchKernelGGL(ell_plan_kernel, dim3(grid1((long long)nrows + 1)), dim3(kBlock), 0, st, nrows, W, AI, ub.p, ubw.p, UBI.p);
        const ClassBounds cb{{0, 16, 32, 64, 128, kWaveCap}};
        zero_async(ccount, sizeof(int) * kNumClasses);
        hipLaunchKernelGGL(esc_classify_kernel, dim3(std::min(grid1(nrows), 1024u)), dim3(kBlock), 0, st, nrows, (const int *)ub.p, cb, ccount.p, lists.p, 1);
        ELAG_LAUNCH_CHECK();
        int *hp = reinterpret_cast<int *>(ctx().h_scalar + 10);
        fetch_ints(ccount.p, kNumClasses, hp);
        for (int c = 0; c < kNumClasses; c++) hc[c] = hp[c];
        nbig = 0; total_small = 0;
        wave_ready = true; all_rows = false;
        total_ub = (unsigned long long)W * (unsigned long long)nnzA;
    }
    int nchunks() const { return (total_small + kCap - 1) / kCap; }
};

// `keep`: the caller's arrays for a product that stays in the upper-bound layout (esc_layout_fits said yes): entries, values and the rows'
// entry counts are written there, nothing else happens (no scan of the counts, no host round trip, no compaction) and C is not touched.
struct EscLayout {
    int *J = nullptr;
    double *a = nullptr;
    int *rownnz = nullptr;
};
inline bool esc_layout_fits(const EscPlan &plan)
{
    const bool in_range = plan.total_ub < (1ull << 31) && plan.total_ub * 12ull <= (16ull << 30);
    const bool single_all = plan.wave_cap > 0 && single_pass_enabled() && plan.wave_ready && plan.all_rows && in_range;
    return plan.wave_cap > 0 && single_pass_enabled() && plan.wave_ready && in_range && (single_all || (plan.nchunks() == 0 && plan.nbig == 0));
}

template <class Expander>
void esc_run(EscPlan &plan, const Expander &ex, int ncols, DevCSR &C, bool spmv_plan = true, const EscLayout *keep = nullptr)
{
    hipStream_t st = ctx().stream;
    const int n = plan.nrows;
    Temp<int> rownnz_own(keep ? 1 : (size_t)n);
    struct { int *p; operator int *() const { return p; } } rownnz{keep ? keep->rownnz : rownnz_own.p};
    zero_async(rownnz, sizeof(int) * (size_t)n);
    if (keep) ELAG_REQUIRE(esc_layout_fits(plan), "internal: esc_run asked to keep a layout that does not hold every row");
    const int nch = plan.nchunks();
    // the symbolic launches of the chunked / one-row-per-workgroup rows -- unless the single pass below has room for those rows too
    const bool single_all = plan.wave_cap > 0 && single_pass_enabled() && plan.wave_ready && plan.all_rows &&
                            plan.total_ub < (1ull << 31) && plan.total_ub * 12ull <= (16ull << 30);
    if (nch && !single_all)
        hipLaunchKernelGGL((esc_kernel<2 * kCap, false, false, Expander>), dim3(nch), dim3(kBlock), 0, st, n, plan.ubscan.p,
                           plan.ub_small.p, (const int *)nullptr, ex, rownnz.p, (const int *)nullptr, (int *)nullptr, (double *)nullptr);
    if (plan.nbig && !single_all)
        hipLaunchKernelGGL((esc_kernel<kBigCap, true, false, Expander>), dim3(plan.nbig), dim3(kBlock), 0, st, n, plan.ubscan.p,
                           plan.ub.p, plan.biglist.p, ex, rownnz.p, (const int *)nullptr, (int *)nullptr, (double *)nullptr);
    // ---- single pass for the wave rows: numeric launch into an upper-bound layout, counts recorded on the way, then a
    //      compaction (the symbolic launches of these rows -- a third of a Galerkin product's time -- are not needed) ----
    if (plan.wave_cap > 0 && single_pass_enabled()) {
        Temp<int> &ubw = plan.ubw, &UBI = plan.UBI, &lists = plan.lists;
        unsigned long long total_ub = plan.total_ub;
        int hc[kNumClasses];
        const int *ubp = plan.ub.p;
        if (plan.wave_ready) {
            for (int c = 0; c < kNumClasses; c++) hc[c] = plan.hc[c];
        } else {
            // (plans whose nrows * wave_cap could overflow the int layout: the total first, the layout only if it fits)
            Temp<unsigned long long> tot64(1);
            zero_async(tot64, sizeof(unsigned long long));
            hipLaunchKernelGGL(wave_ub_kernel, dim3(grid1(n)), dim3(kBlock), 0, st, n, (const int *)plan.ub.p, plan.wave_cap, ubw.p, tot64.p, 0);
            ELAG_LAUNCH_CHECK();
            fetch_ints(reinterpret_cast<const int *>(tot64.p), 2, reinterpret_cast<int *>(&total_ub));
            if (total_ub < (1ull << 31)) {
                exclusive_scan(ubw, UBI, n);
                const ClassBounds cb{{0, 16, 32, 64, 128, kWaveCap}};
                zero_async(plan.ccount, sizeof(int) * kNumClasses);
                hipLaunchKernelGGL(esc_classify_kernel, dim3(std::min(grid1(n), 1024u)), dim3(kBlock), 0, st, n, ubp, cb, plan.ccount.p, lists.p, 1);
                ELAG_LAUNCH_CHECK();
                fetch_ints(plan.ccount.p, kNumClasses, hc);
            }
        }
        // the upper-bound layout is indexed with ints and lives in the scratch arena: at most 2^31 - 1 products and 16 GiB
        if (total_ub < (1ull << 31) && total_ub * 12ull <= (16ull << 30)) {
            Temp<int> Ju_own(keep ? 1 : (size_t)std::max<long long>(total_ub, 1));
            Temp<double> au_own(keep ? 1 : (size_t)std::max<long long>(total_ub, 1));
            struct { int *p; } Ju{keep ? keep->J : Ju_own.p};
            struct { double *p; } au{keep ? keep->a : au_own.p};
            // row classes by number of products, each with the smallest lane group that still expands a row in one pass (the
            // entries of the left factor's row are what the lanes of a group share out: 15 for A, 2-15 for P^T; the products
            // only set how many slots a lane owns, R), each launched over the list of ITS rows
            auto cls_list = [&](int c) { return (const int *)(lists.p + (size_t)c * n); };
            const int hm = hash_classes();         // bit c: class c goes to the hash path (ELAG_HIP_ESC_HASH, default all)
            bool done[kNumClasses] = {false, false, false, false, false};
            if constexpr (Expander::kUnits) {
#define ELAG_HASH_LAUNCH(c, G, R, S, NU, BLOCK)                                                                                                  \
    do {                                                                                                                                        \
        hipLaunchKernelGGL((esc_hash_kernel<G, R, S, NU, BLOCK, Expander>), dim3(grid1((long long)hc[c] * G, BLOCK)), dim3(BLOCK), 0, st, n, ex, \
                           rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(c), hc[c]);                                                       \
        done[c] = true;                                                                                                                         \
    } while (0)
                // (lanes per row, products per lane and pass, table slots, units per lane and pass, workgroup size) per class.  Fewer lanes
                // per row -- (2,2,32,8), (4,8,128,4) -- were measured slower on cube456 x5 (A P 395 against 298 us): the loads of a row
                // coalesce worse, and the kernel is bound by memory and LDS latency, not by the serialised inserts' instructions
#define ELAG_HASHU_LAUNCH(c, G, S, RP, BLOCK)                                                                                                    \
    do {                                                                                                                                        \
        hipLaunchKernelGGL((esc_hashu_kernel<G, S, RP, BLOCK, Expander>), dim3(grid1((long long)hc[c] * G, BLOCK)), dim3(BLOCK), 0, st, n, ex,   \
                           rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(c), hc[c]);                                                       \
        done[c] = true;                                                                                                                         \
    } while (0)
                // long product runs per unit (plan.unit_run: the right factor's entries per row): the units take turns and the lanes insert
                // a unit's products together (esc_hashu_kernel); lanes per row = about the run length
                const double run = hashu_enabled() ? plan.unit_run : 0.0;
                if (run >= 12.0) {
                    if (hc[0] && (hm & 1)) ELAG_HASHU_LAUNCH(0, 16, 16, 1, 256);
                    if (hc[1] && (hm & 2)) ELAG_HASHU_LAUNCH(1, 16, 32, 2, 256);
                    if (hc[2] && (hm & 4)) ELAG_HASHU_LAUNCH(2, 16, 64, 2, 256);
                    if (hc[3] && (hm & 8)) ELAG_HASHU_LAUNCH(3, 16, 128, 2, 256);
                    if (hc[4] && (hm & 16)) ELAG_HASHU_LAUNCH(4, 16, 256, 2, 256);
                } else if (run >= 5.0) {
                    if (hc[0] && (hm & 1)) ELAG_HASHU_LAUNCH(0, 8, 16, 2, 256);
                    if (hc[1] && (hm & 2)) ELAG_HASHU_LAUNCH(1, 8, 32, 2, 256);
                    if (hc[2] && (hm & 4)) ELAG_HASHU_LAUNCH(2, 8, 64, 2, 256);
                    if (hc[3] && (hm & 8)) ELAG_HASHU_LAUNCH(3, 8, 128, 2, 256);
                    if (hc[4] && (hm & 16)) ELAG_HASHU_LAUNCH(4, 16, 256, 2, 256);
                }
#undef ELAG_HASHU_LAUNCH
                // (a right factor known to hold 3 or 4 entries per row: four products per unit and turn, so that a unit's products are inserted
                // together -- storage order, the order of csr_rap_ell's kernels on the same product: every route gives the same bits)
                const bool r4 = plan.max_unit > 2 && plan.max_unit <= 4;
                if (hc[0] && !done[0] && (hm & 1)) { if (r4) ELAG_HASH_LAUNCH(0, 4, 4, 16, 2, 256); else ELAG_HASH_LAUNCH(0, 4, 2, 16, 2, 256); }
                if (hc[1] && !done[1] && (hm & 2)) { if (r4) ELAG_HASH_LAUNCH(1, 4, 4, 32, 4, 256); else ELAG_HASH_LAUNCH(1, 4, 2, 32, 4, 256); }
                if (hc[2] && !done[2] && (hm & 4)) ELAG_HASH_LAUNCH(2, 8, 4, 64, 2, 256);
                if (hc[3] && !done[3] && (hm & 8)) ELAG_HASH_LAUNCH(3, 8, 8, 128, 2, 256);
                if (hc[4] && !done[4] && (hm & 16)) ELAG_HASH_LAUNCH(4, 16, 8, 256, 1, 256);
#undef ELAG_HASH_LAUNCH
            }
            if (hc[0] && !done[0])
                hipLaunchKernelGGL((esc_wave_kernel<16, 1, true, Expander>), dim3(grid1((long long)hc[0] * 16)), dim3(kBlock), 0, st, n, ubp, 0, 16, ex,
                                   rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(0), hc[0]);
            if (hc[1] && !done[1])
                hipLaunchKernelGGL((esc_wave_kernel<16, 2, true, Expander>), dim3(grid1((long long)hc[1] * 16)), dim3(kBlock), 0, st, n, ubp, 16, 32, ex,
                                   rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(1), hc[1]);
            if (hc[2] && !done[2])
                hipLaunchKernelGGL((esc_wave_kernel<32, 2, true, Expander>), dim3(grid1((long long)hc[2] * 32)), dim3(kBlock), 0, st, n, ubp, 32, 64, ex,
                                   rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(2), hc[2]);
            if (hc[3] && !done[3])
                hipLaunchKernelGGL((esc_wave_kernel<64, 2, true, Expander>), dim3(grid1((long long)hc[3] * 64)), dim3(kBlock), 0, st, n, ubp, 64, 128, ex,
                                   rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(3), hc[3]);
            if (hc[4] && !done[4])
                hipLaunchKernelGGL((esc_wave_kernel<64, 4, true, Expander>), dim3(grid1((long long)hc[4] * 64)), dim3(kBlock), 0, st, n, ubp, 128, kWaveCap, ex,
                                   rownnz.p, (const int *)UBI.p, Ju.p, au.p, cls_list(4), hc[4]);
            if (single_all) {
                // the chunked and the one-row-per-workgroup rows write into the upper-bound layout too and record their counts: one
                // numeric launch each instead of a symbolic and a numeric one (cube456 x3, the flattened cycle's G1 G0: 0.67 + 0.38 ms)
                if (nch)
                    hipLaunchKernelGGL((esc_kernel<2 * kCap, false, true, Expander>), dim3(nch), dim3(kBlock), 0, st, n, plan.ubscan.p,
                                       plan.ub_small.p, (const int *)nullptr, ex, rownnz.p, (const int *)UBI.p, Ju.p, au.p, esc_big_hash());
                if (plan.nbig)
                    hipLaunchKernelGGL((esc_kernel<kBigCap, true, true, Expander>), dim3(plan.nbig), dim3(kBlock), 0, st, n, plan.ubscan.p,
                                       plan.ub.p, plan.biglist.p, ex, rownnz.p, (const int *)UBI.p, Ju.p, au.p, esc_big_hash());
            }
            ELAG_LAUNCH_CHECK();
            if (keep) return;                        // the product stays where its kernels wrote it
            int *CI = dev_alloc<int>((size_t)n + 1);
            int *hnnz = reinterpret_cast<int *>(ctx().h_scalar + 10);
            Temp<int> longest(2);                    // [0] the longest row, [1] the number of entries
            zero_async(longest.p, 2 * sizeof(int));
            exclusive_scan(rownnz, CI, n, spmv_plan ? longest.p : nullptr, longest.p + 1);          // (both ride along with the scan: no kernel + drain for them afterwards)
            fetch_ints(longest.p, 2, hnnz);
            const int nnz = hnnz[1];
            C.release();
            C.nrows = n; C.ncols = ncols; C.I = CI;
            C.alloc_entries(nnz);
            if (spmv_plan) C.known_max_row = hnnz[0];
            if (nch && !single_all)
                hipLaunchKernelGGL((esc_kernel<2 * kCap, false, true, Expander>), dim3(nch), dim3(kBlock), 0, st, n, plan.ubscan.p,
                                   plan.ub_small.p, (const int *)nullptr, ex, (int *)nullptr, (const int *)C.I, C.J, C.a, esc_big_hash());
            if (plan.nbig && !single_all)
                hipLaunchKernelGGL((esc_kernel<kBigCap, true, true, Expander>), dim3(plan.nbig), dim3(kBlock), 0, st, n, plan.ubscan.p,
                                   plan.ub.p, plan.biglist.p, ex, (int *)nullptr, (const int *)C.I, C.J, C.a, esc_big_hash());
            hipLaunchKernelGGL(esc_compact_kernel, dim3(grid1((long long)n * 8)), dim3(kBlock), 0, st, n, (const int *)ubw.p, (const int *)UBI.p,
                               (const int *)C.I, (const int *)Ju.p, (const double *)au.p, C.J, C.a);
            ELAG_LAUNCH_CHECK();
            // (Ju / au go back to the arena when this scope ends: the arena recycles in stream order, no drain needed)
            if (spmv_plan) C.choose_spmv_shape();
            return;
        }
    }
    // row classes by number of products: (0,16] 16 lanes per row, (16,32] 32, (32,64] a wavefront, (64,256] 4 per lane
    if (plan.wave_cap > 0) {
        const int *ubp = plan.ub.p;
        hipLaunchKernelGGL((esc_wave_kernel<16, 1, false, Expander>), dim3(grid1((long long)n * 16)), dim3(kBlock), 0, st, n, ubp, 0, 16, ex, rownnz.p,
                           (const int *)nullptr, (int *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL((esc_wave_kernel<32, 1, false, Expander>), dim3(grid1((long long)n * 32)), dim3(kBlock), 0, st, n, ubp, 16, 32, ex, rownnz.p,
                           (const int *)nullptr, (int *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL((esc_wave_kernel<64, 1, false, Expander>), dim3(grid1((long long)n * 64)), dim3(kBlock), 0, st, n, ubp, 32, 64, ex, rownnz.p,
                           (const int *)nullptr, (int *)nullptr, (double *)nullptr);
        hipLaunchKernelGGL((esc_wave_kernel<64, 4, false, Expander>), dim3(grid1((long long)n * 64)), dim3(kBlock), 0, st, n, ubp, 64, kWaveCap, ex, rownnz.p,
                           (const int *)nullptr, (int *)nullptr, (double *)nullptr);
    }
    ELAG_LAUNCH_CHECK();
    int *CI = dev_alloc<int>((size_t)n + 1);
    exclusive_scan(rownnz, CI, n);
    const int nnz = fetch_int(CI + n);
    C.release();
    C.nrows = n; C.ncols = ncols; C.I = CI;
    C.alloc_entries(nnz);
    if (nch)
        hipLaunchKernelGGL((esc_kernel<2 * kCap, false, true, Expander>), dim3(nch), dim3(kBlock), 0, st, n, plan.ubscan.p,
                           plan.ub_small.p, (const int *)nullptr, ex, (int *)nullptr, (const int *)C.I, C.J, C.a, esc_big_hash());
    if (plan.nbig)
        hipLaunchKernelGGL((esc_kernel<kBigCap, true, true, Expander>), dim3(plan.nbig), dim3(kBlock), 0, st, n, plan.ubscan.p,
                           plan.ub.p, plan.biglist.p, ex, (int *)nullptr, (const int *)C.I, C.J, C.a, esc_big_hash());
    if (plan.wave_cap > 0) {
        const int *ubp = plan.ub.p;
        hipLaunchKernelGGL((esc_wave_kernel<16, 1, true, Expander>), dim3(grid1((long long)n * 16)), dim3(kBlock), 0, st, n, ubp, 0, 16, ex, (int *)nullptr,
                           (const int *)C.I, C.J, C.a);
        hipLaunchKernelGGL((esc_wave_kernel<32, 1, true, Expander>), dim3(grid1((long long)n * 32)), dim3(kBlock), 0, st, n, ubp, 16, 32, ex, (int *)nullptr,
                           (const int *)C.I, C.J, C.a);
        hipLaunchKernelGGL((esc_wave_kernel<64, 1, true, Expander>), dim3(grid1((long long)n * 64)), dim3(kBlock), 0, st, n, ubp, 32, 64, ex, (int *)nullptr,
                           (const int *)C.I, C.J, C.a);
        hipLaunchKernelGGL((esc_wave_kernel<64, 4, true, Expander>), dim3(grid1((long long)n * 64)), dim3(kBlock), 0, st, n, ubp, 64, kWaveCap, ex, (int *)nullptr,
                           (const int *)C.I, C.J, C.a);
    }
    ELAG_LAUNCH_CHECK();
    if (spmv_plan) C.choose_spmv_shape();
}

}  // namespace

namespace {
// A * B may go through the W-slot copy of B (MatmulEllExpander + EscPlan::finish_ell): B's rows hold at most 4 entries, every row of the
// product stays within the hash path's 256 products, and the upper-bound layout fits its int indices and the scratch arena
bool ell_applies(const DevCSR &A, const DevCSR &B)
{
    const char *e_ell = getenv("ELAG_HIP_RAP_ELL");                 // (read at every call: a test switches it)
    const int use_ell = e_ell ? atoi(e_ell) : 1;
    const int W = ell_width(B.max_row_nnz);
    return use_ell && B.max_row_nnz >= 1 && B.max_row_nnz <= 4 && B.nnz > 0 && (long long)B.nrows * W < (1ll << 31) && hash_classes() == 31 && wave_path_enabled() &&
           single_pass_enabled() && A.max_row_nnz > 0 && (long long)A.max_row_nnz * W <= kWaveCap && (long long)A.nnz * W < (1ll << 31) &&
           (unsigned long long)A.nnz * W * 12ull <= (16ull << 30);
}
}  // namespace

namespace {
// the W-slot copy of a right factor, built once and kept with the matrix (DevCSR::ellJ / ella)
void ensure_ell(const DevCSR &B, int W)
{
    if (B.ellW == W && B.ellJ && B.ella) return;
    dev_free(B.ellJ); dev_free(B.ella);
    B.ellJ = dev_alloc<int>((size_t)B.nrows * W);
    B.ella = dev_alloc<double>((size_t)B.nrows * W);
    B.ellW = W;
    hipLaunchKernelGGL(ell_build_kernel, dim3(grid1((long long)B.nrows * W)), dim3(kBlock), 0, ctx().stream, B.nrows, W, (const int *)B.I, (const int *)B.J,
                       (const double *)B.a, B.ellJ, B.ella);
    ELAG_LAUNCH_CHECK();
}

unsigned stride_grid(long long groups_wanted, int groups_per_wg, long long max_wgs)
{
    long long wgs = (groups_wanted + groups_per_wg - 1) / groups_per_wg;
    if (wgs > max_wgs) wgs = max_wgs;
    return (unsigned)(wgs < 1 ? 1 : wgs);
}

// A P and, with R, C = R (A P) for a right factor P of at most 4 entries per row (the interpolators of the lowest-order sequence), without a plan:
//   * A P: the W-slot copy of P makes the layout analytic (row i at W * A.I[i], room for every product) -- ell_hash_kernel over ALL rows, the
//     few rows too long for it passed on to a wider launch through a list whose length stays on the device;
//   * R (A P): the rows' upper bounds (one kernel), their scan (the layout), esc_hashd_kernel over all rows, its long rows passed on likewise;
//   * the counts' scan, and ONE read-back after the last kernel: nnz(C), nnz(A P) when the caller wants A P as a matrix too, the longest rows
//     (for the SpMV plans) and whether any row outgrew the widest launch -- in which case nothing has been touched and the caller takes the
//     general route.
// No class lists, no numbers the host waits for in the middle: ~25 dispatches and one drain where the planned route took ~55 and three
// (profiles/r03_rap_timeline_final.txt against profiles/r04_rap_timeline.txt).  The scratch for C's upper-bound layout is sized by the host's
// bound W x (entries of the layout of A P): the route is taken only where that fits kRapEllScratch.
constexpr unsigned long long kRapEllScratch = 24ull << 30;
bool rap_ell(const DevCSR *R, const DevCSR &A, const DevCSR &P, DevCSR *C, DevCSR *AP_out, bool spmv_plan)
{
    hipStream_t st = ctx().stream;
    const int W = ell_width(P.max_row_nnz);
    const int n = A.nrows, nr = R ? R->nrows : 0;
    const unsigned long long total1 = (unsigned long long)W * (unsigned long long)A.nnz;           // entries of the layout of A P
    const unsigned long long bound2 = R ? total1 * (unsigned long long)W : 0ull;                   // sum of the rows' upper bounds of R (A P), at most
    if (total1 >= (1ull << 31) || bound2 >= (1ull << 31) || (total1 + bound2) * 12ull > kRapEllScratch) return false;
    if (R && ((long long)R->nnz <= 0 || nr == 0)) return false;
    ensure_ell(P, W);
    Temp<int> Jap((size_t)std::max<unsigned long long>(total1, 1)), len((size_t)n), ovfA((size_t)n), ovfB((size_t)n), ctl(16);
    Temp<double> aap((size_t)std::max<unsigned long long>(total1, 1));
    zero_async(ctl, 16 * sizeof(int));
    zero_async(len, sizeof(int) * (size_t)n);
    const CsrView a{A.I, A.J, A.a};
    // (lanes per row, steps per pass, slots, products per unit and turn as esc_hash_kernel's class table, table size): rows of at most
    // steps x lanes / min(W, R) entries; the order of insertion is esc_hash_kernel's for the same row, so every route gives the same bits
    // (lanes per row, steps per pass, slots, products per unit and turn, table size): rows of at most steps x lanes / W entries.
    //   launch 1: every row, those of up to 16 (W = 2) / 8 (W = 4) entries are taken, the longer ones passed on through kLists lists;
    //   launch 2: the lists, rows of up to 32 / 16 entries -- every row of the examples' operators is done here;
    //   launches 3, 4: what is longer still (rare; one list), up to 64 | 128 / 32 | 64 entries; beyond that the caller takes the general route.
    constexpr int kLists = 256;
    const long long cus = ctx().num_cus;
    const unsigned grid1a = stride_grid(((long long)n + 7) / 8, 32, 1ll << 30);
    const int capA = (int)(((long long)grid1a * 4 + kLists - 1) / kLists) * (int)(((long long)n + (long long)grid1a * 32 - 1) / ((long long)grid1a * 32)) * 8;   // waves per list x trips x rows per wave and trip
    Temp<int> listsA((size_t)kLists * (size_t)std::max(capA, 1)), cntA(32 * kLists);
    zero_async(cntA, sizeof(int) * 32 * kLists);
    const int *none = nullptr;
    int *nolist = nullptr;
#define ELAG_ELL_LAUNCH(G, NSP, WW, RR, S, GRID, LIST, NDEV, NHOST, NLISTS, CAP, LO, OVF, OVFC, ONL, OCAP)                                         \
    hipLaunchKernelGGL((ell_hash_kernel<G, NSP, WW, RR, S, 256>), dim3(GRID), dim3(256), 0, st, a, (const int *)P.ellJ, (const double *)P.ella, len.p, Jap.p, aap.p, \
                       LIST, NDEV, NHOST, NLISTS, CAP, LO, OVF, OVFC, ONL, OCAP)
    const unsigned grid1b = (unsigned)(kLists * std::max<long long>(1, std::min<long long>(8, ((long long)n / 6 / 8 / 16 + kLists - 1) / kLists)));   // ~n / 6 rows, 8 trips, 16 groups
    if (W == 2) {
        ELAG_ELL_LAUNCH(8, 4, 2, 2, 32, grid1a, none, none, n, 0, 0, 0, listsA.p, cntA.p, kLists, capA);
        ELAG_ELL_LAUNCH(16, 4, 2, 2, 64, grid1b, (const int *)listsA.p, (const int *)cntA.p, n, kLists, capA, 16, ovfA.p, ctl.p + 0, 1, n);
        ELAG_ELL_LAUNCH(16, 8, 2, 2, 128, (unsigned)(2 * cus), (const int *)ovfA.p, (const int *)(ctl.p + 0), n, 0, 0, 32, ovfB.p, ctl.p + 1, 1, n);
        ELAG_ELL_LAUNCH(32, 8, 2, 2, 256, (unsigned)(2 * cus), (const int *)ovfB.p, (const int *)(ctl.p + 1), n, 0, 0, 64, nolist, ctl.p + 2, 1, n);
    } else {
        ELAG_ELL_LAUNCH(8, 4, 4, 4, 32, grid1a, none, none, n, 0, 0, 0, listsA.p, cntA.p, kLists, capA);
        ELAG_ELL_LAUNCH(16, 4, 4, 4, 64, grid1b, (const int *)listsA.p, (const int *)cntA.p, n, kLists, capA, 8, ovfA.p, ctl.p + 0, 1, n);
        ELAG_ELL_LAUNCH(16, 8, 4, 4, 128, (unsigned)(2 * cus), (const int *)ovfA.p, (const int *)(ctl.p + 0), n, 0, 0, 16, ovfB.p, ctl.p + 1, 1, n);
        ELAG_ELL_LAUNCH(32, 8, 4, 4, 256, (unsigned)(2 * cus), (const int *)ovfB.p, (const int *)(ctl.p + 1), n, 0, 0, 32, nolist, ctl.p + 2, 1, n);
    }
#undef ELAG_ELL_LAUNCH
    ELAG_LAUNCH_CHECK();
    int *h = reinterpret_cast<int *>(ctx().h_scalar + 10);              // (doubles 0 and 8 of the pinned block belong to the reductions)
    // ---- A P as a matrix of its own: the scan of the rows' counts (its entries move once their total is known) ----
    int *API = nullptr;
    Temp<int> longAP(1);
    if (AP_out) {
        API = dev_alloc<int>((size_t)n + 1);
        exclusive_scan(len, API, n);
        longest_row(n, API, longAP.p);
    }
    if (!R) {
        hipLaunchKernelGGL(rap_gather_kernel, dim3(1), dim3(64), 0, st, ctl.p, (const int *)nullptr, (const int *)nullptr, (const int *)(API + n), (const int *)longAP.p);
        fetch_ints(ctl.p, 9, h);
        if (h[2] > 0) { dev_free(API); return false; }
    } else {
        // ---- C = R (A P) ----
        CsrView r{R->I, R->J, R->a}, ap{A.I, Jap.p, aap.p, len.p, W};
        Temp<int> ub2((size_t)nr), UBI2((size_t)nr + 1), rn2((size_t)nr), ovfC((size_t)nr), longC(1);
        Temp<long long> pk((size_t)nr), pks((size_t)nr + 1);
        Temp<int> Ju((size_t)std::max<unsigned long long>(bound2, 1));
        Temp<double> au((size_t)std::max<unsigned long long>(bound2, 1));
        // the rows' bounds, and in the same scan the list of the rows beyond 128 products (a sixth of them at cube456 x5: through one counter that
        // many atomics cost more than the launch they feed)
        hipLaunchKernelGGL(matmul_ub8_kernel, dim3(grid1((long long)nr * 8)), dim3(kBlock), 0, st, nr, r, ap, 128, ub2.p, pk.p);
        exclusive_scan64(pk, pks, nr);
        hipLaunchKernelGGL(rap_unpack_kernel, dim3(grid1((long long)nr + 1)), dim3(kBlock), 0, st, nr, (const long long *)pks.p, (const int *)ub2.p, 128, UBI2.p, ovfC.p, ctl.p + 3);
        zero_async(rn2, sizeof(int) * (size_t)nr);
        MatmulExpander ex2{r, ap};
        hipLaunchKernelGGL((esc_hashd_kernel<8, 128, 2, 256, MatmulExpander>), dim3(stride_grid(nr, 32, 1ll << 30)), dim3(256), 0, st, ex2, (const int *)ub2.p,
                           (const int *)UBI2.p, rn2.p, Ju.p, au.p, (const int *)nullptr, (const int *)nullptr, nr, (int *)nullptr, (int *)nullptr);
        hipLaunchKernelGGL((esc_hashd_kernel<16, 256, 2, 256, MatmulExpander>), dim3(stride_grid(nr, 16, 4 * cus)), dim3(256), 0, st, ex2, (const int *)ub2.p,
                           (const int *)UBI2.p, rn2.p, Ju.p, au.p, (const int *)ovfC.p, (const int *)(ctl.p + 3), nr, (int *)nullptr, ctl.p + 4);
        ELAG_LAUNCH_CHECK();
        int *CI = dev_alloc<int>((size_t)nr + 1);
        exclusive_scan(rn2, CI, nr);
        if (spmv_plan) longest_row(nr, CI, longC.p);
        hipLaunchKernelGGL(rap_gather_kernel, dim3(1), dim3(64), 0, st, ctl.p, (const int *)(CI + nr), spmv_plan ? (const int *)longC.p : (const int *)nullptr,
                           AP_out ? (const int *)(API + n) : (const int *)nullptr, AP_out ? (const int *)longAP.p : (const int *)nullptr);
        fetch_ints(ctl.p, 9, h);
        if (h[2] > 0 || h[4] > 0) { dev_free(API); dev_free(CI); return false; }        // a row beyond the widest launch: the general route
        C->release();
        C->nrows = nr; C->ncols = P.ncols; C->I = CI;
        C->alloc_entries(h[5]);
        if (spmv_plan) C->known_max_row = h[6];
        hipLaunchKernelGGL(esc_compact_kernel, dim3(grid1((long long)nr * 8)), dim3(kBlock), 0, st, nr, (const int *)ub2.p, (const int *)UBI2.p, (const int *)C->I,
                           (const int *)Ju.p, (const double *)au.p, C->J, C->a, 1);
        ELAG_LAUNCH_CHECK();
        if (spmv_plan) C->choose_spmv_shape();
    }
    if (AP_out) {
        AP_out->release();
        AP_out->nrows = n; AP_out->ncols = P.ncols; AP_out->I = API;
        AP_out->alloc_entries(h[7]);
        AP_out->known_max_row = h[8];
        hipLaunchKernelGGL(esc_compact_kernel, dim3(grid1((long long)n * 8)), dim3(kBlock), 0, st, n, (const int *)nullptr, (const int *)A.I, (const int *)AP_out->I,
                           (const int *)Jap.p, (const double *)aap.p, AP_out->J, AP_out->a, W);
        ELAG_LAUNCH_CHECK();
    }
    return true;
}

__global__ void zero_ints_kernel(long long n, int *__restrict__ p)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0;
}

// C = R A P by rap_fused_kernel; false (nothing touched) where the route does not apply or a row outgrew it
bool rap_fused(const DevCSR &R, const DevCSR &A, const DevCSR &P, DevCSR &C, bool spmv_plan)
{
    const char *e = getenv("ELAG_HIP_RAP_FUSED");                     // (read at every call: A/B runs and a test switch it)
    if (e && !atoi(e)) return false;
    constexpr int kStride = 64;
    hipStream_t st = ctx().stream;
    const int W = ell_width(P.max_row_nnz);
    const int nr = R.nrows;
    if (nr == 0 || R.nnz == 0 || A.nnz == 0) return false;
    if ((unsigned long long)nr * kStride >= (1ull << 31) || (unsigned long long)nr * kStride * 12ull > kRapEllScratch) return false;
    if (R.max_row_nnz > 64) return false;                               // (known to hold a row of more than 64 children)
    ensure_ell(P, W);
    // the rows' counters and the 16 control words behind them in one block, zeroed by ONE kernel launch (two hipMemsetAsync calls cost the host
    // 8-10 us each before the product kernel could be enqueued: it started 22 us after the second fill had ended)
    Temp<int> rnctl((size_t)nr + 16), Ju((size_t)nr * kStride);
    Temp<double> au((size_t)nr * kStride);
    struct { int *p; } rn{rnctl.p}, ctl{rnctl.p + nr};
    hipLaunchKernelGGL(zero_ints_kernel, dim3(std::min(grid1((long long)nr + 16), 2048u)), dim3(kBlock), 0, st, (long long)nr + 16, rnctl.p);
    const CsrView r{R.I, R.J, R.a}, a{A.I, A.J, A.a};
    // a wavefront walks ~8 rows (the pipeline takes four rows to fill), four wavefronts per workgroup
    const unsigned grid = stride_grid(((long long)nr + 7) / 8, 4, 1ll << 30);
    // (steps per trip: 2 -- 0.78-0.96 ms per product at cube456 x5 -- against 1.0-1.07 with 3 or 4 and 1.5 with 8: the registers of a deeper batch cost more
    // wavefronts than its loads in flight buy, ELAG_HIP_FUSED_B=4 for A/B runs; and holding the compiler to 6 / 7 / 8 wavefronts per SIMD
    // -- amdgpu_waves_per_eu: 80 / 72 / 64 registers with 6 / 8 / 16 of them spilled -- costs 0.85 / 0.92 / 1.37 ms against 0.78 at its own 5)
    static int fb = -1;
    if (fb < 0) { const char *eb = getenv("ELAG_HIP_FUSED_B"); fb = eb ? atoi(eb) : 2; }
#define ELAG_FUSED_LAUNCH(WW, BB) hipLaunchKernelGGL((rap_fused_kernel<WW, BB, 128, 256>), dim3(grid), dim3(256), 0, st, r, a, (const int *)P.ellJ, (const double *)P.ella, nr, rn.p, Ju.p, au.p, kStride, ctl.p + 2)
    if (W == 2 && fb == 4) ELAG_FUSED_LAUNCH(2, 4);
    else if (W == 2) ELAG_FUSED_LAUNCH(2, 2);
    else ELAG_FUSED_LAUNCH(4, 2);
#undef ELAG_FUSED_LAUNCH
    ELAG_LAUNCH_CHECK();
    int *CI = dev_alloc<int>((size_t)nr + 1);
    // the rows' offsets, with nnz(C) -> ctl[5] and the longest row -> ctl[6] by the scan's own kernels (ctl was zeroed above; three scan kernels,
    // a fill, a row-length maximum and a gather kernel before: 35 of the ~125 us between the product kernel and the result)
    exclusive_scan(rn.p, CI, nr, spmv_plan ? ctl.p + 6 : nullptr, ctl.p + 5);
    int *h = reinterpret_cast<int *>(ctx().h_scalar + 10);
    fetch_ints(ctl.p, 9, h);
    if (h[2] > 0) { dev_free(CI); return false; }
    C.release();
    C.nrows = nr; C.ncols = P.ncols; C.I = CI;
    C.alloc_entries(h[5]);
    if (spmv_plan) C.known_max_row = h[6];
    hipLaunchKernelGGL(esc_compact_kernel, dim3(grid1((long long)nr * 8)), dim3(kBlock), 0, st, nr, (const int *)nullptr, (const int *)nullptr, (const int *)C.I,
                       (const int *)Ju.p, (const double *)au.p, C.J, C.a, kStride);
    ELAG_LAUNCH_CHECK();
    if (spmv_plan) C.choose_spmv_shape();
    return true;
}
}  // namespace

void csr_spgemm(const DevCSR &A, const DevCSR &B, DevCSR &C, bool spmv_plan)
{
    ELAG_REQUIRE(A.ncols == B.nrows, "spgemm: inner dimensions differ (%d vs %d)", A.ncols, B.nrows);
    if (A.nrows == 0) { C.set_empty(0, B.ncols); return; }
    if (ell_applies(A, B) && ell_pipe_enabled() && rap_ell(nullptr, A, B, nullptr, &C, spmv_plan)) {
        if (spmv_plan) C.choose_spmv_shape();
        else { C.max_row_nnz = C.known_max_row; C.known_max_row = 0; }
        return;
    }
    if (ell_applies(A, B)) {
        hipStream_t st = ctx().stream;
        const int W = ell_width(B.max_row_nnz);
        Temp<int> EJ((size_t)B.nrows * W);
        Temp<double> Ea((size_t)B.nrows * W);
        hipLaunchKernelGGL(ell_build_kernel, dim3(grid1((long long)B.nrows * W)), dim3(kBlock), 0, st, B.nrows, W, (const int *)B.I, (const int *)B.J,
                           (const double *)B.a, EJ.p, Ea.p);
        EscPlan plan(A.nrows);
        plan.wave_cap = kWaveCap;
        plan.unit_run = (double)W;
        plan.max_unit = W;
        plan.finish_ell(A.I, W, (long long)A.nnz);
        MatmulEllExpander ex{CsrView{A.I, A.J, A.a}, EJ.p, Ea.p, W};
        esc_run(plan, ex, B.ncols, C, spmv_plan);
        return;
    }
    EscPlan plan(A.nrows);
    CsrView a{A.I, A.J, A.a}, b{B.I, B.J, B.a};
    hipLaunchKernelGGL(matmul_ub_kernel, dim3(grid1(A.nrows)), dim3(kBlock), 0, ctx().stream, A.nrows, a, b, plan.ub.p);
    plan.wave_cap = wave_path_enabled() ? kWaveCap : 0;
    plan.unit_run = B.nrows ? (double)B.nnz / (double)B.nrows : 0.0;
    plan.max_unit = B.max_row_nnz;
    plan.finish();
    MatmulExpander ex{a, b};
    esc_run(plan, ex, B.ncols, C, spmv_plan);
}

void csr_rap(const DevCSR &R, const DevCSR &A, const DevCSR &P, DevCSR &C, bool spmv_plan)
{
    ELAG_REQUIRE(A.ncols == P.nrows && R.ncols == A.nrows, "rap: shapes do not chain");
    const char *e_loose = getenv("ELAG_HIP_RAP_LOOSE");             // (read at every call: a test switches it)
    const int loose = e_loose ? atoi(e_loose) : 1;
    hipStream_t st = ctx().stream;
    if (loose && A.nrows > 0 && R.nrows > 0) {
        CsrView a{A.I, A.J, A.a}, p{P.I, P.J, P.a};
        const int W = ell_width(P.max_row_nnz);
        if (ell_applies(A, P) && ell_pipe_enabled() && (rap_fused(R, A, P, C, spmv_plan) || rap_ell(&R, A, P, &C, nullptr, spmv_plan))) return;
        if (ell_applies(A, P)) {
            // the right factor with W slots per row (MatmulEllExpander): valid when every row of A P takes the hash path
            Temp<int> EJ((size_t)P.nrows * W);
            Temp<double> Ea((size_t)P.nrows * W);
            hipLaunchKernelGGL(ell_build_kernel, dim3(grid1((long long)P.nrows * W)), dim3(kBlock), 0, st, P.nrows, W, (const int *)P.I, (const int *)P.J,
                               (const double *)P.a, EJ.p, Ea.p);
            EscPlan plan(A.nrows);
            plan.wave_cap = kWaveCap;
            plan.unit_run = (double)W;
            plan.max_unit = W;
            plan.finish_ell(A.I, W, (long long)A.nnz);
            if (esc_layout_fits(plan) && plan.nchunks() == 0 && plan.nbig == 0) {
                MatmulEllExpander exl{a, EJ.p, Ea.p, W};
                Temp<int> Jap((size_t)std::max<unsigned long long>(plan.total_ub, 1)), len((size_t)A.nrows);
                Temp<double> aap((size_t)std::max<unsigned long long>(plan.total_ub, 1));
                const EscLayout lay{Jap.p, aap.p, len.p};
                DevCSR unused;
                esc_run(plan, exl, P.ncols, unused, false, &lay);
                EscPlan plan2(R.nrows);
                CsrView r{R.I, R.J, R.a}, ap{plan.UBI.p, Jap.p, aap.p, len.p};
                hipLaunchKernelGGL(matmul_ub_kernel, dim3(grid1(R.nrows)), dim3(kBlock), 0, st, R.nrows, r, ap, plan2.ub.p);
                plan2.wave_cap = kWaveCap;
                plan2.extra = len.p; plan2.extra_n = A.nrows;
                plan2.finish();
                plan2.unit_run = (double)plan2.extra_value / 64.0;
                MatmulExpander ex2{r, ap};
                esc_run(plan2, ex2, P.ncols, C, spmv_plan);
                return;
            }
            // (a row of A with more than 256 / W entries: the general route below)
        }
        EscPlan plan(A.nrows);
        hipLaunchKernelGGL(matmul_ub_kernel, dim3(grid1(A.nrows)), dim3(kBlock), 0, st, A.nrows, a, p, plan.ub.p);
        plan.wave_cap = wave_path_enabled() ? kWaveCap : 0;
        plan.unit_run = P.nrows ? (double)P.nnz / (double)P.nrows : 0.0;
        plan.max_unit = P.max_row_nnz;
        plan.finish();
        MatmulExpander ex{a, p};
        if (esc_layout_fits(plan)) {
            // A P in the upper-bound layout: rows at plan.UBI, their lengths in `len`; every array is a Temp of this frame, released (in
            // reverse order) when the second product is done
            Temp<int> Jap((size_t)std::max<unsigned long long>(plan.total_ub, 1)), len((size_t)A.nrows);
            Temp<double> aap((size_t)std::max<unsigned long long>(plan.total_ub, 1));
            const EscLayout lay{Jap.p, aap.p, len.p};
            DevCSR unused;
            esc_run(plan, ex, P.ncols, unused, false, &lay);
            EscPlan plan2(R.nrows);
            CsrView r{R.I, R.J, R.a}, ap{plan.UBI.p, Jap.p, aap.p, len.p};
            hipLaunchKernelGGL(matmul_ub_kernel, dim3(grid1(R.nrows)), dim3(kBlock), 0, st, R.nrows, r, ap, plan2.ub.p);
            plan2.wave_cap = wave_path_enabled() ? kWaveCap : 0;
            plan2.extra = len.p; plan2.extra_n = A.nrows;   // (the mean row length of A P, sampled, comes back with the second plan's own numbers)
            plan2.finish();
            plan2.unit_run = (double)plan2.extra_value / 64.0;
            MatmulExpander ex2{r, ap};
            esc_run(plan2, ex2, P.ncols, C, spmv_plan);
            return;
        }
        DevCSR AP;
        esc_run(plan, ex, P.ncols, AP, false);
        csr_spgemm(R, AP, C, spmv_plan);
        return;
    }
    DevCSR AP;
    csr_spgemm(A, P, AP, false);
    csr_spgemm(R, AP, C, spmv_plan);
}

void preload_setup_kernels()
{
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(matmul_ub_kernel));
}

void csr_add(double ca, const DevCSR &A, double cb, const DevCSR &B, DevCSR &C)
{
    ELAG_REQUIRE(A.nrows == B.nrows && A.ncols == B.ncols, "add: incompatible matrix dimensions");
    if (A.nrows == 0) { C.set_empty(0, A.ncols); return; }
    EscPlan plan(A.nrows);
    CsrView a{A.I, A.J, A.a}, b{B.I, B.J, B.a};
    hipLaunchKernelGGL(add_ub_kernel, dim3(grid1(A.nrows)), dim3(kBlock), 0, ctx().stream, A.nrows, a, b, plan.ub.p);
    plan.wave_cap = wave_path_enabled() ? kWaveCap : 0;
    plan.unit_run = (double)(A.nnz + B.nnz) / (2.0 * (double)A.nrows);          // (a unit is a whole row of A or of B)
    plan.finish();
    AddExpander ex{a, b, ca, cb};
    esc_run(plan, ex, A.ncols, C);
}

namespace {
// Rows of at most G entries: one G-lane group per row sorts (column, value) pairs in registers with a shuffle bitonic
// network -- no LDS, no barriers, in place (row lengths do not change, so no count / scan / fill passes either).
template <int G>
__global__ __launch_bounds__(kBlock) void sort_rows_group_kernel(int nrows, const int *__restrict__ I, int *__restrict__ J, double *__restrict__ a)
{
    const int row = (blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    if (row >= nrows) return;
    const int s = I[row], len = I[row + 1] - s;
    if (len < 2) return;
    int key = 0x7fffffff;
    double v = 0.0;
    if (lane < len) { key = J[s + lane]; v = a[s + lane]; }
#pragma unroll
    for (int k = 2; k <= G; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int pk = __shfl_xor(key, j, 64);
            const double pv = __shfl_xor(v, j, 64);
            const bool up = ((lane & k) == 0);
            const bool lower = ((lane & j) == 0);
            // the lower lane of an ascending pair keeps the smaller key (ties: everybody keeps its own)
            const bool take = (lower == up) ? (pk < key) : (pk > key);
            if (take) { key = pk; v = pv; }
        }
    }
    if (lane < len) { J[s + lane] = key; a[s + lane] = v; }
}
}  // namespace

namespace {
// The same for rows of at most 16 / 32 entries by RANKING: a lane counts the keys of its row that come before its own (ties by lane) and
// stores its pair at that position -- the other keys pass by through DPP row rotations (vector-ALU moves; the second half of a 32-lane group
// through one shuffle), where the bitonic network above exchanges 3 words through the LDS crossbar 15 times per lane: that kernel was
// bound by the LDS pipe, 628 us for the 37.7 M entries of the renumbered cube456 x5 operator.
template <int N>
struct RankSteps {
    static __device__ __forceinline__ void run(int key, int other, int lane16, bool other_is_mine, bool other_before, int &rank)
    {
        constexpr int n = 16 - N;                                             // rotation 0 (only for the other half) ... 15
        if (n > 0 || !other_is_mine) {
            const int k = n == 0 ? other : __builtin_amdgcn_update_dpp(0, other, 0x120 + (n == 0 ? 1 : n), 0xf, 0xf, false);   // row_ror:n -- lane i reads lane (i - n) & 15
            const int src = (lane16 - n) & 15;
            const bool before = other_is_mine ? (k < key || (k == key && src < lane16)) : (k < key || (k == key && other_before));
            rank += before ? 1 : 0;
        }
        RankSteps<N - 1>::run(key, other, lane16, other_is_mine, other_before, rank);
    }
};
template <>
struct RankSteps<0> {
    static __device__ __forceinline__ void run(int, int, int, bool, bool, int &) {}
};

template <int G>
__global__ __launch_bounds__(kBlock) void sort_rows_rank_kernel(int nrows, const int *__restrict__ I, int *__restrict__ J, double *__restrict__ a)
{
    static_assert(G == 16 || G == 32, "rank sort: groups of 16 or 32 lanes");
    const int row = (blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1), lane16 = threadIdx.x & 15;
    // (no early exit: DPP and shuffles read the lanes of rows past the end and of short rows too)
    int s = 0, len = 0;
    if (row < nrows) { s = I[row]; len = I[row + 1] - s; }
    int key = 0x7fffffff;
    double v = 0.0;
    if (lane < len) { key = J[s + lane]; v = a[s + lane]; }
    int rank = 0;
    RankSteps<16>::run(key, key, lane16, true, false, rank);                   // the 15 other lanes of this lane's DPP row
    if (G == 32) {
        const int other = __shfl_xor(key, 16, 64);                             // the key of the same position in the other half
        RankSteps<16>::run(key, other, lane16, false, (lane & 16) != 0, rank);    // all 16 of them (ties: the lower half comes first)
    }
    if (len >= 2 && lane < len) { J[s + rank] = key; a[s + rank] = v; }
}
}  // namespace

namespace {
// Row permutation, column renumbering and the sort of the renumbered rows in ONE pass (rows of at most G entries): the permuted operator was
// written unsorted and read again by the rank sort -- at cube456 x5 253 + 469 us for the level-0 operator alone, 0.6 ms of perm_fill_kernel per build.
template <int G>
__global__ __launch_bounds__(kBlock) void perm_sort_rank_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                                                const int *__restrict__ order, const int *__restrict__ colrank, const int *__restrict__ OI,
                                                                int *__restrict__ OJ, double *__restrict__ Oa)
{
    static_assert(G == 16 || G == 32, "rank sort: groups of 16 or 32 lanes");
    const int row = (blockIdx.x * kBlock + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1), lane16 = threadIdx.x & 15;
    int s = 0, len = 0, d = 0;
    if (row < n) { const int o = order[row]; s = I[o]; len = I[o + 1] - s; d = OI[row]; }
    int key = 0x7fffffff;
    double v = 0.0;
    if (lane < len) { const int c = J[s + lane]; key = colrank ? colrank[c] : c; v = a[s + lane]; }
    int rank = 0;
    RankSteps<16>::run(key, key, lane16, true, false, rank);
    if (G == 32) {
        const int other = __shfl_xor(key, 16, 64);
        RankSteps<16>::run(key, other, lane16, false, (lane & 16) != 0, rank);
    }
    if (lane < len) { OJ[d + rank] = key; Oa[d + rank] = v; }
}
bool rank_sort_enabled()
{
    static int rank_sort = -1;
    if (rank_sort < 0) { const char *e = getenv("ELAG_HIP_RANK_SORT"); rank_sort = e ? atoi(e) : 1; }
    return rank_sort != 0;
}
}  // namespace

void csr_sort_rows(DevCSR &A)
{
    if (A.nrows == 0 || A.nnz == 0) return;
    dev_free(A.J16);                                                           // (a copy of J in its old order)
    if (A.block_row == nullptr || A.max_row_nnz <= 0) build_stream_plan(A);   // max_row_nnz
    if (A.max_row_nnz <= 64) {
        hipStream_t st = ctx().stream;
        const int *I = A.I;
        const bool rank_sort = rank_sort_enabled();
        if (rank_sort && A.max_row_nnz <= 16) hipLaunchKernelGGL((sort_rows_rank_kernel<16>), dim3(grid1((long long)A.nrows * 16)), dim3(kBlock), 0, st, A.nrows, I, A.J, A.a);
        else if (rank_sort && A.max_row_nnz <= 32) hipLaunchKernelGGL((sort_rows_rank_kernel<32>), dim3(grid1((long long)A.nrows * 32)), dim3(kBlock), 0, st, A.nrows, I, A.J, A.a);
        else if (A.max_row_nnz <= 16) hipLaunchKernelGGL((sort_rows_group_kernel<16>), dim3(grid1((long long)A.nrows * 16)), dim3(kBlock), 0, st, A.nrows, I, A.J, A.a);
        else if (A.max_row_nnz <= 32) hipLaunchKernelGGL((sort_rows_group_kernel<32>), dim3(grid1((long long)A.nrows * 32)), dim3(kBlock), 0, st, A.nrows, I, A.J, A.a);
        else hipLaunchKernelGGL((sort_rows_group_kernel<64>), dim3(grid1((long long)A.nrows * 64)), dim3(kBlock), 0, st, A.nrows, I, A.J, A.a);
        ELAG_LAUNCH_CHECK();
        A.known_max_row = A.max_row_nnz;               // (sorting a row does not change its length: no second look for the longest one)
        A.choose_spmv_shape();
        return;
    }
    EscPlan plan(A.nrows);
    hipLaunchKernelGGL(rowlen_kernel, dim3(grid1(A.nrows)), dim3(kBlock), 0, ctx().stream, A.nrows, (const int *)A.I, plan.ub.p);
    plan.finish();
    SortExpander ex{CsrView{A.I, A.J, A.a}};
    DevCSR out;
    esc_run(plan, ex, A.ncols, out);  // duplicates (none expected) would be summed
    A = std::move(out);
}

// ------------------------------------------------------------------------------------------------
// transpose: column histogram -> scan -> atomic fill -> per-row sort (deterministic result)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void col_hist_kernel(long long nnz, const int *__restrict__ J, int *__restrict__ cnt)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x)
        atomicAdd(&cnt[J[k]], 1);
}
__global__ void transpose_fill_kernel(int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                      int *__restrict__ cursor, int *__restrict__ TJ, double *__restrict__ Ta)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    for (int k = I[row]; k < I[row + 1]; k++) {
        const int p = atomicAdd(&cursor[J[k]], 1);
        TJ[p] = row;
        Ta[p] = a[k];
    }
}
}  // namespace

void csr_transpose(const DevCSR &A, DevCSR &AT)
{
    hipStream_t st = ctx().stream;
    AT.alloc(A.ncols, A.nrows, A.nnz);
    Temp<int> cnt((size_t)A.ncols);
    zero_async(cnt, sizeof(int) * (size_t)(A.ncols > 0 ? A.ncols : 1));
    if (A.nnz) {
        long long g = (A.nnz + kBlock - 1) / kBlock; if (g > 4096) g = 4096;
        hipLaunchKernelGGL(col_hist_kernel, dim3((unsigned)g), dim3(kBlock), 0, st, (long long)A.nnz, (const int *)A.J, cnt.p);
    }
    exclusive_scan(cnt, AT.I, A.ncols);
    if (A.nnz) {
        Temp<int> cursor((size_t)A.ncols);
        dev_copy(cursor, AT.I, sizeof(int) * (size_t)A.ncols, st);
        hipLaunchKernelGGL(transpose_fill_kernel, dim3(grid1(A.nrows)), dim3(kBlock), 0, st, A.nrows, (const int *)A.I,
                           (const int *)A.J, (const double *)A.a, cursor.p, AT.J, AT.a);
        ELAG_LAUNCH_CHECK();
        csr_sort_rows(AT);
    } else {
        AT.choose_spmv_shape();
    }
}

// ------------------------------------------------------------------------------------------------
// delete zeros / scale rows / copy
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void count_keep_kernel(int nrows, const int *__restrict__ I, const double *__restrict__ a, double tol, int *__restrict__ cnt)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    int c = 0;
    for (int k = I[row]; k < I[row + 1]; k++) c += (fabs(a[k]) > tol);
    cnt[row] = c;
}
__global__ void fill_keep_kernel(int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a, double tol,
                                 const int *__restrict__ OI, int *__restrict__ OJ, double *__restrict__ Oa)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    int p = OI[row];
    for (int k = I[row]; k < I[row + 1]; k++)
        if (fabs(a[k]) > tol) { OJ[p] = J[k]; Oa[p] = a[k]; p++; }
}
__global__ void scale_rows_kernel(int nrows, const int *__restrict__ I, double *__restrict__ a, const double *__restrict__ d)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nrows) return;
    const double s = d[row];
    for (int k = I[row]; k < I[row + 1]; k++) a[k] *= s;
}
}  // namespace

void csr_delete_zeros(const DevCSR &A, double tol, DevCSR &out)
{
    hipStream_t st = ctx().stream;
    const int n = A.nrows;
    Temp<int> cnt((size_t)n);
    int *OI = dev_alloc<int>((size_t)n + 1);
    if (n) hipLaunchKernelGGL(count_keep_kernel, dim3(grid1(n)), dim3(kBlock), 0, st, n, (const int *)A.I, (const double *)A.a, tol, cnt.p);
    exclusive_scan(cnt, OI, n);
    const int nnz = fetch_int(OI + n);
    out.release();
    out.nrows = n; out.ncols = A.ncols; out.I = OI;
    out.alloc_entries(nnz);
    if (n) hipLaunchKernelGGL(fill_keep_kernel, dim3(grid1(n)), dim3(kBlock), 0, st, n, (const int *)A.I, (const int *)A.J,
                              (const double *)A.a, tol, (const int *)OI, out.J, out.a);
    ELAG_LAUNCH_CHECK();
    out.choose_spmv_shape();
}

namespace {
__global__ void scale_cols_kernel(long long nnz, const int *__restrict__ J, double *__restrict__ a, const double *__restrict__ d)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) a[k] *= d[J[k]];
}
}  // namespace

void csr_scale_cols(DevCSR &A, const double *d)
{
    A.release_sign();
    if (A.nnz == 0) return;
    long long g = (A.nnz + kBlock - 1) / kBlock; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)g), dim3(kBlock), 0, ctx().stream, (long long)A.nnz, (const int *)A.J, A.a, d);
    ELAG_LAUNCH_CHECK();
}

void csr_scale_rows(DevCSR &A, const double *d)
{
    A.release_sign();
    if (A.nrows == 0) return;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid1(A.nrows)), dim3(kBlock), 0, ctx().stream, A.nrows, (const int *)A.I, A.a, d);
    ELAG_LAUNCH_CHECK();
}

namespace {
__global__ void perm_len_kernel(int n, const int *__restrict__ I, const int *__restrict__ order, int *__restrict__ len)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { const int o = order[r]; len[r] = I[o + 1] - I[o]; }
}
__global__ void perm_fill_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                 const int *__restrict__ order, const int *__restrict__ rank, const int *__restrict__ OI,
                                 int *__restrict__ OJ, double *__restrict__ Oa)
{
    // 8 lanes per output row
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int lane = threadIdx.x & 7;
    if (r >= n) return;
    const int o = order[r];
    const int s = I[o], len = I[o + 1] - s, d = OI[r];
    for (int q = lane; q < len; q += 8) { OJ[d + q] = rank ? rank[J[s + q]] : J[s + q]; Oa[d + q] = a[s + q]; }
}
}  // namespace

void csr_permute(const DevCSR &A, const int *d_order_rows, const int *d_rank_cols, int ncols_out, DevCSR &out, bool sort_rows)
{
    hipStream_t st = ctx().stream;
    const int n = A.nrows;
    out.release();
    out.nrows = n; out.ncols = ncols_out;
    out.I = dev_alloc<int>((size_t)n + 1);
    {
        Temp<int> len((size_t)n);
        if (n) hipLaunchKernelGGL(perm_len_kernel, dim3(grid1(n)), dim3(kBlock), 0, st, n, (const int *)A.I, d_order_rows, len.p);
        exclusive_scan(len, out.I, n);
    }
    out.alloc_entries(A.nnz);
    static const bool fused_sort = !(getenv("ELAG_HIP_PERM_SORT") && !atoi(getenv("ELAG_HIP_PERM_SORT")));
    if (n && A.nnz && sort_rows && fused_sort && rank_sort_enabled() && A.max_row_nnz > 0 && A.max_row_nnz <= 32) {
        if (A.max_row_nnz <= 16)
            hipLaunchKernelGGL((perm_sort_rank_kernel<16>), dim3(grid1((long long)n * 16)), dim3(kBlock), 0, st, n, (const int *)A.I, (const int *)A.J,
                               (const double *)A.a, d_order_rows, d_rank_cols, (const int *)out.I, out.J, out.a);
        else
            hipLaunchKernelGGL((perm_sort_rank_kernel<32>), dim3(grid1((long long)n * 32)), dim3(kBlock), 0, st, n, (const int *)A.I, (const int *)A.J,
                               (const double *)A.a, d_order_rows, d_rank_cols, (const int *)out.I, out.J, out.a);
        ELAG_LAUNCH_CHECK();
        out.known_max_row = A.max_row_nnz;
        out.max_row_nnz = A.max_row_nnz;
        out.choose_spmv_shape();
        return;
    }
    if (n && A.nnz)
        hipLaunchKernelGGL(perm_fill_kernel, dim3(grid1((long long)n * 8)), dim3(kBlock), 0, st, n, (const int *)A.I, (const int *)A.J,
                           (const double *)A.a, d_order_rows, d_rank_cols, (const int *)out.I, out.J, out.a);
    ELAG_LAUNCH_CHECK();
    out.known_max_row = A.max_row_nnz;                 // (a permutation of the rows keeps the longest one)
    if (A.nnz && sort_rows) csr_sort_rows(out);
    else out.choose_spmv_shape();
}

namespace {
__global__ void parent_key_kernel(int nf, int nc, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                  const int *__restrict__ rank_c, int *__restrict__ KI, int *__restrict__ KJ, double *__restrict__ Ka)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > nf) return;
    KI[i] = i;
    if (i == nf) return;
    int best = -1;
    double bv = -1.0;
    for (int k = I[i]; k < I[i + 1]; k++) {
        const double v = fabs(a[k]);
        if (v > bv) { bv = v; best = J[k]; }
    }
    KJ[i] = best < 0 ? nc : rank_c[best];
    Ka[i] = 1.0;
}
__global__ void inverse_perm_kernel(int n, const int *__restrict__ order, int *__restrict__ rank)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) rank[order[r]] = r;
}
}  // namespace

void child_order(const DevCSR &P, const int *d_rank_c, int *d_order_out)
{
    const int nf = P.nrows, nc = P.ncols;
    if (nf == 0) return;
    hipStream_t st = ctx().stream;
    DevCSR K, KT;
    K.alloc(nf, nc + 1, nf);
    hipLaunchKernelGGL(parent_key_kernel, dim3(grid1(nf + 1)), dim3(kBlock), 0, st, nf, nc, (const int *)P.I, (const int *)P.J, (const double *)P.a, d_rank_c,
                       K.I, K.J, K.a);
    ELAG_LAUNCH_CHECK();
    csr_transpose(K, KT);            // row p of KT: the children of the coarse dof of rank p, ascending
    dev_copy(d_order_out, KT.J, sizeof(int) * (size_t)nf, st);
    // (no drain: K / KT go back to the pool, which hands them out again in stream order)
}

void inverse_permutation(int n, const int *d_order, int *d_rank)
{
    if (n == 0) return;
    hipLaunchKernelGGL(inverse_perm_kernel, dim3(grid1(n)), dim3(kBlock), 0, ctx().stream, n, d_order, d_rank);
    ELAG_LAUNCH_CHECK();
}

namespace {
__global__ void diagonal_fill_kernel(int n, const double *__restrict__ d, double scale, int *__restrict__ I, int *__restrict__ J, double *__restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { I[i] = i; J[i] = i; a[i] = d ? scale * d[i] : scale; }
    if (i == n) I[n] = n;
}
}  // namespace

void csr_diagonal(int n, const double *d, double scale, DevCSR &out)
{
    out.alloc(n, n, n);
    hipLaunchKernelGGL(diagonal_fill_kernel, dim3(grid1(n + 1)), dim3(kBlock), 0, ctx().stream, n, d, scale, out.I, out.J, out.a);
    ELAG_LAUNCH_CHECK();
    out.choose_spmv_shape();
}

namespace {
__global__ void combine_inplace_kernel(int nrows, const int *__restrict__ CI, const int *__restrict__ CJ, double *__restrict__ Ca, double c_scale,
                                       const double *__restrict__ c_row, const double *__restrict__ c_col, const int *__restrict__ BI,
                                       const int *__restrict__ BJ, const double *__restrict__ Ba, double b_scale, const double *__restrict__ b_row,
                                       int b_row_divide, const double *__restrict__ diag, double d_scale, int *__restrict__ missing)
{
    // 8 lanes per row of C: every entry of C is written by exactly one lane (scaled, plus the other operand's entry of that column if
    // there is one -- a binary search in B's ascending row --, plus the diagonal term); the entries found are counted against B's row
    const int row = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 3);
    const int lane = threadIdx.x & 7;
    if (row >= nrows) return;
    const int cs = CI[row], ce = CI[row + 1];
    const int bs = BI ? BI[row] : 0, be = BI ? BI[row + 1] : 0;
    const double rs = c_scale * (c_row ? c_row[row] : 1.0);
    double bs_row = b_scale;
    if (b_row) bs_row = b_row_divide ? b_scale / b_row[row] : b_scale * b_row[row];
    int found = 0, has_diag = 0;
    for (int k = cs + lane; k < ce; k += 8) {
        const int j = CJ[k];
        double v = rs * Ca[k] * (c_col ? c_col[j] : 1.0);
        int lo = bs, hi = be;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (BJ[mid] < j) lo = mid + 1; else hi = mid; }
        if (lo < be && BJ[lo] == j) { v += bs_row * Ba[lo]; found++; }
        if (diag && j == row) { v += d_scale * diag[row]; has_diag = 1; }
        Ca[k] = v;
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { found += __shfl_xor(found, o, 8); has_diag += __shfl_xor(has_diag, o, 8); }
    if (lane == 0 && (found != be - bs || (diag && !has_diag))) atomicAdd(missing, 1);
}
}  // namespace

bool csr_combine_inplace(DevCSR &C, const CsrCombine &op)
{
    if (C.nrows == 0) return true;
    if (op.B) ELAG_REQUIRE(op.B->nrows == C.nrows && op.B->ncols == C.ncols, "internal: csr_combine_inplace shapes");
    hipStream_t st = ctx().stream;
    C.release_sign();
    Temp<int> missing(1);
    zero_async(missing, sizeof(int));
    hipLaunchKernelGGL(combine_inplace_kernel, dim3(grid1((long long)C.nrows * 8)), dim3(kBlock), 0, st, C.nrows, (const int *)C.I, (const int *)C.J, C.a, op.c_scale,
                       op.c_row, op.c_col, op.B ? (const int *)op.B->I : nullptr, op.B ? (const int *)op.B->J : nullptr,
                       op.B ? (const double *)op.B->a : nullptr, op.b_scale, op.b_row, op.b_row_divide, op.diag, op.d_scale, missing.p);
    ELAG_LAUNCH_CHECK();
    int *hm = reinterpret_cast<int *>(ctx().h_scalar + 10);
    fetch_ints(missing.p, 1, hm);
    return *hm == 0;
}

void csr_copy(const DevCSR &A, DevCSR &out)
{
    hipStream_t st = ctx().stream;
    out.alloc(A.nrows, A.ncols, A.nnz);
    dev_copy(out.I, A.I, sizeof(int) * ((size_t)A.nrows + 1), st);
    if (A.nnz) {
        dev_copy(out.J, A.J, sizeof(int) * (size_t)A.nnz, st);
        dev_copy(out.a, A.a, sizeof(double) * (size_t)A.nnz, st);
    }
    out.choose_spmv_shape();
}

}  // namespace k
}  // namespace elag
