// CSR SpMV family, vector kernels and row statistics for gfx950 (wave64).
//
// SpMV design (HBM-bound, no MFMA): a group of LPR lanes (power of two, 1..64, chosen from nnz/row)
// owns one row, so a 64-lane wavefront covers 64/LPR consecutive rows and its loads of J/a sweep one
// contiguous span of the CSR arrays (coalesced 4-/8-byte accesses); partial sums are combined with
// cross-lane shuffles inside the group and lane 0 applies the fused epilogue.  The x gathers are served
// from L2 / Infinity Cache.  Algorithmic bytes per launch: 12 nnz + 4 (n+1) + 8 ncols + 8 n (+8 n per extra
// epilogue vector), see DESIGN.md.
#include "kernels.hpp"
#include "row_group.hpp"

#include <algorithm>

namespace elag {
namespace k {

namespace {

constexpr int kBlock = 256;

// the sum of a group's LPR values on every lane of the group (a butterfly of ds_bpermute exchanges)
template <int LPR>
__device__ __forceinline__ double group_reduce_all(double v)
{
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
using namespace rowgroup;      // dpp_shifted, lanes_up, group_reduce<LPR> (lane 0 of the group: the butterfly's value bit for bit), kRowU

// one row by a group of LPR lanes (all of them active): products in lane order, group reduction, fused epilogue by lane 0
template <int LPR, bool HAS_OFFD, bool HAS_CS>
__device__ __forceinline__ void spmv_fused_row(int row, int lane, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                               const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa, const SpmvArgs &p)
{
    int os = 0, oe = 0;
    if (HAS_OFFD || p.skip_rows_with_offd) {
        if (oI) { os = oI[row]; oe = oI[row + 1]; }
        if (p.skip_rows_with_offd && oe > os) return;
    }

    const int s = I[row], e = I[row + 1];
    // These kernels serve the launch/latency-bound levels, where the time is the length of the dependent chain of memory
    // round trips, not the bytes: the epilogue operands are requested up front (they only need `row`), and the first
    // entry of the diag and of the offd stream travel together, as do their two gathers.
    double f0 = 0.0, y0 = 0.0, sc0 = 1.0, sc20 = 1.0;
    if (lane == 0) {
        if (p.s2) sc20 = p.s2[row];
        if (p.f) f0 = p.f[row];
        if (p.s) sc0 = p.s[row];
        if (p.yin) y0 = p.yin[row];
    }
    const double *__restrict__ x = p.x;
    const double *__restrict__ cs = p.cs;
    const double *__restrict__ xg = p.xg;
    // kRowU entries per lane and pass, diag and offd together: all the column / value loads of a pass are requested before the first
    // gather, all the gathers before the first product.  The operators these kernels serve sit in L2 / Infinity Cache, where a
    // launch costs (entries) / (entries in flight per CU) round trips.  (What this buys depends on the size of the launch: the
    // 0.2-0.9 M-entry products of the level-by-level cycles sit on the ~4 us floor of a dependent launch either way -- H(div) x3
    // 0.139 -> 0.134 ms, Darcy 0.0886 -> 0.086 ms --, the 1.2 M-entry stages of the flattened cycle went from 6.5 to 5.3 us.)
    constexpr int U = kRowU;
    double sum = 0.0;
    const int nd_ = e - s, no_ = HAS_OFFD ? oe - os : 0;
    for (int q0 = lane; q0 < nd_ || q0 < no_; q0 += U * LPR) {
        int jd[U], jo[U];
        double ad[U], ao[U], xd[U], xo[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + u * LPR;
            jd[u] = 0; ad[u] = 0.0;
            if (q < nd_) { jd[u] = J[s + q]; ad[u] = a[s + q]; }
            if (HAS_OFFD) {
                jo[u] = 0; ao[u] = 0.0;
                if (q < no_) { jo[u] = oJ[os + q]; ao[u] = oa[os + q]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int q = q0 + u * LPR;
            xd[u] = 0.0;
            if (q < nd_) { xd[u] = x[jd[u]]; if (HAS_CS) xd[u] *= cs[jd[u]]; }
            if (HAS_OFFD) {
                xo[u] = 0.0;
                if (q < no_) { xo[u] = xg[jo[u]]; if (HAS_CS && p.csg) xo[u] *= p.csg[jo[u]]; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            sum += ad[u] * xd[u];
            if (HAS_OFFD) sum += ao[u] * xo[u];
        }
    }
    sum = group_reduce<LPR>(sum);
    if (lane == 0) {
        double t = p.alpha * sum;
        t *= sc20;
        t += p.gamma * f0;
        t *= sc0;
        t += p.beta * y0;
        p.out[p.out_perm ? p.out_perm[row] : row] = p.out_add ? p.out_add[p.out_perm[row]] + t : t;
        if (p.out2) p.out2[row] = HAS_CS ? x[row] * p.cs[row] : x[row];
    }
}

template <int LPR, bool HAS_OFFD, bool HAS_CS>
__global__ __launch_bounds__(kBlock) void spmv_fused_kernel(
    int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
    const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa, SpmvArgs p)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int grp = gid / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int nsel = p.rows ? p.nrows_sel : nrows;
    if (grp >= nsel) return;
    const int row = p.rows ? p.rows[grp] : grp;
    spmv_fused_row<LPR, HAS_OFFD, HAS_CS>(row, lane, I, J, a, oI, oJ, oa, p);
}

template <int LPR>
void launch_lpr(const DevCSR &A, const DevCSR *offd, const SpmvArgs &p, hipStream_t st)
{
    const int nsel = p.rows ? p.nrows_sel : A.nrows;
    if (nsel == 0) return;
    const long long threads = (long long)nsel * LPR;
    const unsigned grid = (unsigned)((threads + kBlock - 1) / kBlock);
    const bool has_offd = offd && offd->nnz > 0 && !p.skip_rows_with_offd;
    const bool has_cs = p.cs != nullptr;
    const int *oI = offd ? offd->I : nullptr;
    const int *oJ = offd ? offd->J : nullptr;
    const double *oa = offd ? offd->a : nullptr;
    if (has_offd) {
        if (has_cs)
            hipLaunchKernelGGL((spmv_fused_kernel<LPR, true, true>), dim3(grid), dim3(kBlock), 0, st, A.nrows, A.I, A.J, A.a, oI, oJ, oa, p);
        else
            hipLaunchKernelGGL((spmv_fused_kernel<LPR, true, false>), dim3(grid), dim3(kBlock), 0, st, A.nrows, A.I, A.J, A.a, oI, oJ, oa, p);
    } else {
        if (has_cs)
            hipLaunchKernelGGL((spmv_fused_kernel<LPR, false, true>), dim3(grid), dim3(kBlock), 0, st, A.nrows, A.I, A.J, A.a, oI, oJ, oa, p);
        else
            hipLaunchKernelGGL((spmv_fused_kernel<LPR, false, false>), dim3(grid), dim3(kBlock), 0, st, A.nrows, A.I, A.J, A.a, oI, oJ, oa, p);
    }
    ELAG_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Sign form (DevCSR::Js): rows of -1 / 0 / +1 entries, 4 bytes per entry.  LPR lanes per row, each lane takes up to four
// entries per pass: the four words are requested together, then the four x values, and only then added -- in storage order
// within a lane, so with LPR = 1 the row sum is the CPU loop's (a product by +-1 is exact).
// ------------------------------------------------------------------------------------------------
template <int LPR, bool HAS_CS>
__global__ __launch_bounds__(kBlock) void spmv_sign_kernel(int nrows, const int *__restrict__ I, const int *__restrict__ Js, SpmvArgs p)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int row = gid / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= nrows) return;
    const int s = I[row], e = I[row + 1];
    double f0 = 0.0, y0 = 0.0, sc0 = 1.0, sc20 = 1.0;
    if (lane == 0) {
        if (p.s2) sc20 = p.s2[row];
        if (p.f) f0 = p.f[row];
        if (p.s) sc0 = p.s[row];
        if (p.yin) y0 = p.yin[row];
    }
    const double *__restrict__ x = p.x;
    const double *__restrict__ cs = p.cs;
    double sum = 0.0;
    for (int k = s + lane; k < e; k += 4 * LPR) {
        int w[4];
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = (k + q * LPR < e) ? Js[k + q * LPR] : 0x40000000;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = w[q] & 0x3fffffff;
            v[q] = (w[q] & 0x40000000) ? 0.0 : (HAS_CS ? x[j] * cs[j] : x[j]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) sum += (w[q] < 0) ? -v[q] : v[q];
    }
    sum = group_reduce<LPR>(sum);
    if (lane == 0) {
        double t = p.alpha * sum;
        t *= sc20;
        t += p.gamma * f0;
        t *= sc0;
        t += p.beta * y0;
        p.out[p.out_perm ? p.out_perm[row] : row] = p.out_add ? p.out_add[p.out_perm[row]] + t : t;
        if (p.out2) p.out2[row] = HAS_CS ? x[row] * p.cs[row] : x[row];
    }
}

template <int LPR>
void launch_sign(const DevCSR &A, const SpmvArgs &p, hipStream_t st)
{
    if (A.nrows == 0) return;
    const unsigned grid = (unsigned)(((long long)A.nrows * LPR + kBlock - 1) / kBlock);
    if (p.cs) hipLaunchKernelGGL((spmv_sign_kernel<LPR, true>), dim3(grid), dim3(kBlock), 0, st, A.nrows, (const int *)A.I, (const int *)A.Js, p);
    else hipLaunchKernelGGL((spmv_sign_kernel<LPR, false>), dim3(grid), dim3(kBlock), 0, st, A.nrows, (const int *)A.I, (const int *)A.Js, p);
    ELAG_LAUNCH_CHECK();
}

// x_out = x_in + D (wa .* (D^T pv)) for a sign-form D and its stored transpose, in ONE launch and without the intermediate vector: the
// auxiliary-space correction of a Hiptmair sweep whose auxiliary smoother is one diagonal sweep from the zero guess (HiptmairSmoother.cpp:61-75:
// cB = D^T defect; relax; X += D correction).  A group of LPR lanes per row of D; a lane walks the transposed row of its entry four words
// at a time.  (Two launches -- D^T with the scaling in its epilogue, then D -- were 2 x 5.8 us of the 130 us H(div) cycle at cube456 x3,
// four times per cycle.)
template <int LPR>
__global__ __launch_bounds__(kBlock) void sign_twohop_kernel(int nrows, const int *__restrict__ DI, const int *__restrict__ DJs, const int *__restrict__ TI,
                                                            const int *__restrict__ TJs, const double *__restrict__ wa, const double *__restrict__ pv,
                                                            const double *__restrict__ xin, double *__restrict__ xout)
{
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int row = gid / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= nrows) return;
    const int s = DI[row], e = DI[row + 1];
    double x0 = 0.0;
    if (lane == 0) x0 = xin[row];
    double sum = 0.0;
    for (int k = s + lane; k < e; k += LPR) {
        const int w = DJs[k];
        if (w & 0x40000000) continue;                       // a stored zero (eliminated essential column)
        const int j = w & 0x3fffffff;
        const int ts = TI[j], te = TI[j + 1];
        const double wj = wa[j];
        double t = 0.0;
        for (int q0 = ts; q0 < te; q0 += 4) {
            int u[4];
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) u[q] = (q0 + q < te) ? TJs[q0 + q] : 0x40000000;
#pragma unroll
            for (int q = 0; q < 4; q++) v[q] = (u[q] & 0x40000000) ? 0.0 : pv[u[q] & 0x3fffffff];
#pragma unroll
            for (int q = 0; q < 4; q++) t += (u[q] < 0) ? -v[q] : v[q];
        }
        t *= wj;
        sum += (w < 0) ? -t : t;
    }
    sum = group_reduce<LPR>(sum);
    if (lane == 0) xout[row] = x0 + sum;
}

__global__ void sign_pack_kernel(long long nnz, const int *__restrict__ J, const double *__restrict__ a, int *__restrict__ Js, int *__restrict__ bad)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x) {
        const double v = a[k];
        int w = J[k];
        if (v == -1.0) w |= (int)0x80000000;
        else if (v == 0.0) w |= 0x40000000;
        else if (v != 1.0) atomicAdd(bad, 1);
        Js[k] = w;
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-staged stream kernel (default).  A 256-thread workgroup owns the rows whose first entry lies in one
// 2048-entry window of the CSR arrays.  Phase 1 streams that window with 16-byte loads (int4 of J, 2 x double2
// of a per lane: up to 160 B in flight per lane), gathers x and parks the products in LDS; phase 2 gives each
// row to one lane, which sums its products in storage order (hypre's order, so the sum is bit-identical to the
// CPU loop) and applies the fused epilogue with coalesced vector accesses.
// ------------------------------------------------------------------------------------------------
typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

template <bool HAS_CS, int VARIANT>
__device__ __forceinline__ void spmv_stream_body(
    int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
    const int *__restrict__ block_row, const int *__restrict__ oI, const SpmvArgs &p)
{
    extern __shared__ double prod[];
    // one 16-byte scalar load: (first row, first entry) of this block and of the next one
    const int4v br = *reinterpret_cast<const int4v *>(block_row + 2 * blockIdx.x);
    const int r0 = br.x, kbeg = br.y, r1 = br.z, kend = br.w;
    if (r0 >= r1) return;
    const int kal = kbeg & ~3;
    const double *__restrict__ x = p.x;
    const double *__restrict__ cs = p.cs;

    // Issue everything that depends only on (r0, kbeg) before waiting on anything: the row extents and the
    // epilogue operands of this lane's first row travel together with the matrix stream, which takes three
    // dependent memory round trips out of the wave's critical path.
    const int row0 = r0 + (int)threadIdx.x;
    const bool has0 = row0 < r1;
    int s0 = 0, e0 = 0;
    double f0 = 0.0, y0 = 0.0, sc0 = 1.0, sc20 = 1.0;
    if (has0) {
        s0 = I[row0]; e0 = I[row0 + 1];
        if (p.f) f0 = p.f[row0];
        if (p.yin) y0 = p.yin[row0];
        if (p.s) sc0 = p.s[row0];
        if (p.s2) sc20 = p.s2[row0];
    }

    auto quad = [&](int k) {
        int4v j4 = *reinterpret_cast<const int4v *>(J + k);
        const double2v a01 = *reinterpret_cast<const double2v *>(a + k);
        const double2v a23 = *reinterpret_cast<const double2v *>(a + k + 2);
        if (VARIANT & 4) {  // DIAGNOSTIC ONLY (wrong results): coalesced x reads instead of the gather
            const int b = (k - kal) + r0;
            j4.x = min(b, nrows - 1) + 0 * j4.x; j4.y = min(b + 1, nrows - 1) + 0 * j4.y; j4.z = min(b + 2, nrows - 1) + 0 * j4.z; j4.w = min(b + 3, nrows - 1) + 0 * j4.w;
        }
        double x0 = x[j4.x], x1 = x[j4.y], x2 = x[j4.z], x3 = x[j4.w];
        if (HAS_CS) { x0 *= cs[j4.x]; x1 *= cs[j4.y]; x2 *= cs[j4.z]; x3 *= cs[j4.w]; }
        double2v p01, p23;
        p01.x = a01.x * x0; p01.y = a01.y * x1; p23.x = a23.x * x2; p23.y = a23.y * x3;
        *reinterpret_cast<double2v *>(prod + (k - kal)) = p01;
        *reinterpret_cast<double2v *>(prod + (k - kal) + 2) = p23;
    };

    for (int k = kal + (int)threadIdx.x * 4; k < kend; k += kBlock * 4) quad(k);
    __syncthreads();
    if (has0 && !(p.skip_rows_with_offd && oI && oI[row0 + 1] > oI[row0])) {
        double sum = 0.0;
        for (int k = s0 - kal; k < e0 - kal; k++) sum += prod[k];
        double t = p.alpha * sum;
        t *= sc20;
        t += p.gamma * f0;
        t *= sc0;
        t += p.beta * y0;
        p.out[p.out_perm ? p.out_perm[row0] : row0] = p.out_add ? p.out_add[p.out_perm[row0]] + t : t;
        if (p.out2) p.out2[row0] = HAS_CS ? x[row0] * p.cs[row0] : ((VARIANT & 1) ? x[row0] * p.o2s[row0] : x[row0]);   // (variant 1: out2 scaled by o2s)
    }
    // blocks of short rows hold more than 256 rows: remaining rows the plain way
    for (int row = row0 + kBlock; row < r1; row += kBlock) {
        if (p.skip_rows_with_offd && oI && oI[row + 1] > oI[row]) continue;
        const int s = I[row] - kal, e = I[row + 1] - kal;
        double sum = 0.0;
        for (int k = s; k < e; k++) sum += prod[k];
        double t = p.alpha * sum;
        if (p.s2) t *= p.s2[row];
        if (p.f) t += p.gamma * p.f[row];
        if (p.s) t *= p.s[row];
        if (p.yin) t += p.beta * p.yin[row];
        p.out[p.out_perm ? p.out_perm[row] : row] = p.out_add ? p.out_add[p.out_perm[row]] + t : t;
        if (p.out2) p.out2[row] = HAS_CS ? x[row] * p.cs[row] : ((VARIANT & 1) ? x[row] * p.o2s[row] : x[row]);
    }
}

template <bool HAS_CS, int VARIANT>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(
    int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
    const int *__restrict__ block_row, const int *__restrict__ oI, SpmvArgs p)
{
    spmv_stream_body<HAS_CS, VARIANT>(nrows, I, J, a, block_row, oI, p);
}

// The same code under another name: the launches of the setup-time measurement (prepare_spmv) must not be counted under
// the production kernel's name by rocprofv3 --stats / --pmc.
__global__ __launch_bounds__(kBlock) void spmv_probe_stream_kernel(
    int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
    const int *__restrict__ block_row, const int *__restrict__ oI, SpmvArgs p)
{
    spmv_stream_body<false, 0>(nrows, I, J, a, block_row, oI, p);
}

// ---- x-cache variant -------------------------------------------------------------------------------------------------
// Same window decomposition and storage-order row sums as spmv_stream_kernel, but the gather is done once per DISTINCT
// column of the window: phase 1 loads x (times cs) for the window's sorted unique columns into LDS (ascending addresses,
// long contiguous runs after the hierarchy renumbering), phase 2 streams the values with a 16-bit LDS index per entry
// (10 bytes per entry from HBM instead of 12, and no global gather in the inner loop).
typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));

// LEAN: the plain product out = A x (no epilogue operand, no permuted output, no second output, alpha = 1): a build of its own -- this kernel's
// time moves by several per cent with what its epilogue has to keep alive (an unused select there cost 7 %)
template <bool HAS_CS, int THREADS, bool LEAN = false>
__device__ __forceinline__ void spmv_xc_body(
    int nrows, const int *__restrict__ I, const double *__restrict__ a, const unsigned short *__restrict__ lidx,
    const int *__restrict__ block_row, const int *__restrict__ uoff, const int *__restrict__ ucols, const int *__restrict__ oI,
    int cap, const SpmvArgs &p)
{
    extern __shared__ double sh[];
    double *prod = sh, *xs = sh + cap;
    // Workgroups are handed to the 8 XCDs round-robin by their id, and every XCD has an L2 of its own: with window = id,
    // neighbouring windows -- which share most of the x lines they gather -- run on different XCDs and each fetches those lines
    // from the fabric itself.  With xcd_chunks the ids of one XCD walk ONE contiguous eighth of the windows instead, so a line
    // fetched for a window is still in that XCD's L2 when the next window needs it.
    int win = blockIdx.x;
    if (p.xcd_chunks) {
        const int nb = (int)gridDim.x, xcd = win & 7, idx = win >> 3, per = nb >> 3, rem = nb & 7;
        win = xcd * per + min(xcd, rem) + idx;
    }
    const int4v br = *reinterpret_cast<const int4v *>(block_row + 2 * win);
    const int r0 = br.x, kbeg = br.y, r1 = br.z, kend = br.w;
    if (r0 >= r1) return;
    const int kal = kbeg & ~3;
    const double *__restrict__ x = p.x;
    const double *__restrict__ cs = p.cs;
    const int u0 = uoff[win], nu = uoff[win + 1] - u0;

    const int row0 = r0 + (int)threadIdx.x;
    const bool has0 = row0 < r1;
    int s0 = 0, e0 = 0;
    double f0 = 0.0, y0 = 0.0, sc0 = 1.0, sc20 = 1.0;
    if (has0) {
        s0 = I[row0]; e0 = I[row0 + 1];
        if (!LEAN) {
            if (p.f) f0 = p.f[row0];
            if (p.yin) y0 = p.yin[row0];
            if (p.s) sc0 = p.s[row0];
            if (p.s2) sc20 = p.s2[row0];
        }
    }
    // the value / position stream of this lane's first two quads is requested before the gather phase, so both
    // memory streams are in flight together
    const int k0 = kal + (int)threadIdx.x * 4, k1 = k0 + THREADS * 4;
    ushort4v l0 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
    double2v a0a = {0, 0}, a0b = {0, 0}, a1a = {0, 0}, a1b = {0, 0};
    if (k0 < kend) { l0 = *reinterpret_cast<const ushort4v *>(lidx + k0); a0a = *reinterpret_cast<const double2v *>(a + k0); a0b = *reinterpret_cast<const double2v *>(a + k0 + 2); }
    if (k1 < kend) { l1 = *reinterpret_cast<const ushort4v *>(lidx + k1); a1a = *reinterpret_cast<const double2v *>(a + k1); a1b = *reinterpret_cast<const double2v *>(a + k1 + 2); }
    for (int u = (int)threadIdx.x; u < nu; u += THREADS) {
        const int c = ucols[u0 + u];
        double v = x[c];
        if (HAS_CS) v *= cs[c];
        xs[u] = v;
    }
    __syncthreads();
    if (k0 < kend) {
        double2v p01, p23;
        p01.x = a0a.x * xs[l0.x]; p01.y = a0a.y * xs[l0.y]; p23.x = a0b.x * xs[l0.z]; p23.y = a0b.y * xs[l0.w];
        *reinterpret_cast<double2v *>(prod + (k0 - kal)) = p01;
        *reinterpret_cast<double2v *>(prod + (k0 - kal) + 2) = p23;
    }
    if (k1 < kend) {
        double2v p01, p23;
        p01.x = a1a.x * xs[l1.x]; p01.y = a1a.y * xs[l1.y]; p23.x = a1b.x * xs[l1.z]; p23.y = a1b.y * xs[l1.w];
        *reinterpret_cast<double2v *>(prod + (k1 - kal)) = p01;
        *reinterpret_cast<double2v *>(prod + (k1 - kal) + 2) = p23;
    }
    for (int k = k1 + THREADS * 4; k < kend; k += THREADS * 4) {
        const ushort4v l4 = *reinterpret_cast<const ushort4v *>(lidx + k);
        const double2v a01 = *reinterpret_cast<const double2v *>(a + k);
        const double2v a23 = *reinterpret_cast<const double2v *>(a + k + 2);
        // (entries of the neighbouring windows inside the first / last quad carry their own window's positions; they
        //  stay below the xs capacity, and their products are never summed)
        double2v p01, p23;
        p01.x = a01.x * xs[l4.x]; p01.y = a01.y * xs[l4.y]; p23.x = a23.x * xs[l4.z]; p23.y = a23.y * xs[l4.w];
        *reinterpret_cast<double2v *>(prod + (k - kal)) = p01;
        *reinterpret_cast<double2v *>(prod + (k - kal) + 2) = p23;
    }
    __syncthreads();
    if (LEAN) {
        if (has0) {
            double sum = 0.0;
            for (int k = s0 - kal; k < e0 - kal; k++) sum += prod[k];
            p.out[row0] = sum;
        }
        for (int row = row0 + THREADS; row < r1; row += THREADS) {
            const int s = I[row] - kal, e = I[row + 1] - kal;
            double sum = 0.0;
            for (int k = s; k < e; k++) sum += prod[k];
            p.out[row] = sum;
        }
        return;
    }
    if (has0 && !(p.skip_rows_with_offd && oI && oI[row0 + 1] > oI[row0])) {
        double sum = 0.0;
        for (int k = s0 - kal; k < e0 - kal; k++) sum += prod[k];
        double t = p.alpha * sum;
        t *= sc20;
        t += p.gamma * f0;
        t *= sc0;
        t += p.beta * y0;
        p.out[p.out_perm ? p.out_perm[row0] : row0] = p.out_add ? p.out_add[p.out_perm[row0]] + t : t;
        if (p.out2) p.out2[row0] = HAS_CS ? x[row0] * p.cs[row0] : x[row0];          // (o2s: not in this kernel)
    }
    for (int row = row0 + THREADS; row < r1; row += THREADS) {
        if (p.skip_rows_with_offd && oI && oI[row + 1] > oI[row]) continue;
        const int s = I[row] - kal, e = I[row + 1] - kal;
        double sum = 0.0;
        for (int k = s; k < e; k++) sum += prod[k];
        double t = p.alpha * sum;
        if (p.s2) t *= p.s2[row];
        if (p.f) t += p.gamma * p.f[row];
        if (p.s) t *= p.s[row];
        if (p.yin) t += p.beta * p.yin[row];
        p.out[p.out_perm ? p.out_perm[row] : row] = p.out_add ? p.out_add[p.out_perm[row]] + t : t;
        if (p.out2) p.out2[row] = HAS_CS ? x[row] * p.cs[row] : x[row];
    }
}

template <bool HAS_CS, int THREADS, bool LEAN = false>
__global__ __launch_bounds__(THREADS) void spmv_xc_kernel(
    int nrows, const int *__restrict__ I, const double *__restrict__ a, const unsigned short *__restrict__ lidx,
    const int *__restrict__ block_row, const int *__restrict__ uoff, const int *__restrict__ ucols, const int *__restrict__ oI,
    int cap, SpmvArgs p)
{
    spmv_xc_body<HAS_CS, THREADS, LEAN>(nrows, I, a, lidx, block_row, uoff, ucols, oI, cap, p);
}

// the same code under another name for the setup-time measurement (see spmv_probe_stream_kernel)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spmv_probe_xc_kernel(
    int nrows, const int *__restrict__ I, const double *__restrict__ a, const unsigned short *__restrict__ lidx,
    const int *__restrict__ block_row, const int *__restrict__ uoff, const int *__restrict__ ucols, const int *__restrict__ oI,
    int cap, SpmvArgs p)
{
    spmv_xc_body<false, THREADS>(nrows, I, a, lidx, block_row, uoff, ucols, oI, cap, p);
}

// One workgroup per window: the window's DISTINCT columns through a hash table in LDS, sorted (bitonic, over the few hundred distinct columns
// only), every entry's position in that list looked up through the table.  (Until round 4 the kernel sorted all the window's ENTRIES, padded to
// the next power of two -- 4 096 keys, 78 barrier-separated stages of 8 compare-exchanges per thread for a 2 304-entry window -- and was 2.9 ms
// of a 16.4 ms hierarchy build at cube456 x5.)  The result -- sorted distinct columns, 16-bit positions -- does not depend on the insertion order.
// fill == 0: a SAMPLE -- workgroup b looks at window b * step and writes ucount[2 b] = its number of distinct columns, ucount[2 b + 1] = its
// entries.  fill == 1: ucols (at uoff[c]) / lidx are written.  fill == 2: the one full pass of xc_build -- ucount[c], lidx and the window's
// columns at ucols + c * step (a slot per window; compacted once the counts are scanned).
// CAP: entries per window the builder handles (window + longest row + 8 <= CAP) = slots of the table; 4096 for 2048-entry windows, 8192 for 4096
template <int CAP>
__global__ __launch_bounds__(kBlock) void xc_build_kernel(const int *__restrict__ block_row, const int *__restrict__ J, int *__restrict__ ucount,
                                                          const int *__restrict__ uoff, int *__restrict__ ucols,
                                                          unsigned short *__restrict__ lidx, int fill, int step)
{
    constexpr int LOG2CAP = CAP == 4096 ? 12 : 13;
    static_assert((1 << LOG2CAP) == CAP, "table sizes are 4096 and 8192");
    __shared__ int hkey[CAP];                    // the table's keys (-1: empty)
    __shared__ unsigned short hval[CAP];         // ... and each key's position in the sorted list
    __shared__ int ukeys[CAP];                   // the distinct columns, sorted in place
    __shared__ int wsum[kBlock / 64];
    const int c = fill == 0 ? (int)blockIdx.x * step : (int)blockIdx.x;
    const int kbeg = block_row[2 * c + 1], kend = block_row[2 * c + 3];
    const int len = kend - kbeg;
    auto slot_of = [](int col) { return (int)(((unsigned)col * 2654435761u) >> (32 - LOG2CAP)); };
    for (int i = threadIdx.x; i < CAP; i += kBlock) hkey[i] = -1;
    __syncthreads();
    for (int e = threadIdx.x; e < len; e += kBlock) {
        const int col = J[kbeg + e];
        int h = slot_of(col);
        for (;;) {
            const int prev = atomicCAS(&hkey[h], -1, col);
            if (prev == -1 || prev == col) break;
            h = (h + 1) & (CAP - 1);
        }
    }
    __syncthreads();
    // the occupied slots, compacted: a thread owns CAP / kBlock consecutive slots
    constexpr int PER = CAP / kBlock;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) cnt += hkey[threadIdx.x * PER + q] >= 0 ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int off = incl - cnt, nu = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) {
        const int v = wsum[w];
        if (w < (int)(threadIdx.x >> 6)) off += v;
        nu += v;
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int k = hkey[threadIdx.x * PER + q];
        if (k >= 0) ukeys[off++] = k;
    }
    int npow = 64;
    while (npow < nu) npow <<= 1;
    for (int i = nu + (int)threadIdx.x; i < npow; i += kBlock) ukeys[i] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= npow; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < (npow >> 1); t += kBlock) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const int a = ukeys[lo], b = ukeys[hi];
                if ((a > b) == up) { ukeys[lo] = b; ukeys[hi] = a; }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        if (fill == 0) { ucount[2 * blockIdx.x] = nu; ucount[2 * blockIdx.x + 1] = len; }
        if (fill == 2) ucount[c] = nu;
    }
    if (!fill) return;
    const long long u0 = fill == 2 ? (long long)c * step : (long long)uoff[c];
    for (int i = threadIdx.x; i < nu; i += kBlock) {
        const int col = ukeys[i];
        ucols[u0 + i] = col;
        int h = slot_of(col);
        while (hkey[h] != col) h = (h + 1) & (CAP - 1);
        hval[h] = (unsigned short)i;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < len; e += kBlock) {
        const int col = J[kbeg + e];
        int h = slot_of(col);
        while (hkey[h] != col) h = (h + 1) & (CAP - 1);
        lidx[kbeg + e] = hval[h];
    }
}

// the windows' column lists out of their slots into one array (uoff = scanned counts)
__global__ void xc_compact_kernel(int nb, int slot, const int *__restrict__ slots, const int *__restrict__ uoff, int *__restrict__ ucols)
{
    for (int c = blockIdx.x; c < nb; c += gridDim.x) {
        const int u0 = uoff[c], n = uoff[c + 1] - u0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) ucols[u0 + i] = slots[(size_t)c * slot + i];
    }
}

__global__ void stream_plan_kernel(int nrows, const int *__restrict__ I, int nblocks, int window, int *__restrict__ block_row)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > nblocks) return;
    if (c == nblocks) { block_row[2 * c] = nrows; block_row[2 * c + 1] = I[nrows]; return; }
    const long long v = (long long)c * window;
    int lo = 0, hi = nrows;  // first row with I[row] >= v
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (I[mid] < v) lo = mid + 1; else hi = mid; }
    block_row[2 * c] = lo;
    block_row[2 * c + 1] = I[lo];
}

__global__ void max_row_kernel(int nrows, const int *__restrict__ I, int *__restrict__ result)
{
    __shared__ int sh[4];
    int m = 0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x) m = max(m, I[r + 1] - I[r]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    // one atomic per workgroup, and only when it can change the result (thousands of waves hitting one address cost 50 us)
    if (threadIdx.x == 0) {
        const int w = (int)(blockDim.x >> 6);
        for (int i = 1; i < w; i++) m = max(m, sh[i]);
        if (m > 0) atomicMax(result, m);
    }
}

int stream_variant()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_STREAM_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

template <int VARIANT>
void launch_stream_v(const DevCSR &A, const DevCSR *offd, const SpmvArgs &p, hipStream_t st)
{
    const size_t lds = sizeof(double) * (size_t)(DevCSR::kStreamNnz + A.max_row_nnz + 8);
    const int *oI = (p.skip_rows_with_offd && offd) ? offd->I : nullptr;
    if (p.cs)
        hipLaunchKernelGGL((spmv_stream_kernel<true, VARIANT>), dim3((unsigned)A.nblocks), dim3(kBlock), lds, st, A.nrows, A.I, A.J, A.a, A.block_row, oI, p);
    else
        hipLaunchKernelGGL((spmv_stream_kernel<false, VARIANT>), dim3((unsigned)A.nblocks), dim3(kBlock), lds, st, A.nrows, A.I, A.J, A.a, A.block_row, oI, p);
    ELAG_LAUNCH_CHECK();
}

int xc_mode()
{
    static int v = -2;
    if (v == -2) { const char *e = getenv("ELAG_HIP_XC"); v = e ? atoi(e) : -1; }   // 0 = off, 1 = always when applicable, -1 = heuristic
    return v;
}

int xc_xcd_mode()
{
    static int v = -1;
    // default off: measured on cube456 x5 (profiles/r02_spmv_xcd_placement.txt) the chunked placement cuts the launch's HBM
    // reads from 550 to 447 MB (total traffic 0.93 x algorithmic instead of 1.13 x) and costs 1-4 % of time -- the duplicate
    // fetches of x by several XCDs are where the excess traffic comes from, but they are not what bounds the kernel
    if (v < 0) { const char *e = getenv("ELAG_HIP_XC_XCD"); v = e ? atoi(e) : 0; }
    return v;
}

bool xc_lean_enabled()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_XC_LEAN"); v = e ? (atoi(e) != 0) : 1; }
    return v != 0;
}

template <int THREADS>
void launch_xc_t(const DevCSR &A, const int *oI, const SpmvArgs &pin, hipStream_t st, bool probe)
{
    SpmvArgs p = pin;
    p.xcd_chunks = (xc_xcd_mode() && A.xc_nblocks >= 64) ? 1 : 0;
    const int cap = A.xc_window + A.max_row_nnz + 8;
    const size_t lds = sizeof(double) * ((size_t)cap + (size_t)A.xc_max_unique);
    const dim3 grid((unsigned)A.xc_nblocks), block(THREADS);
    if (probe)
        hipLaunchKernelGGL((spmv_probe_xc_kernel<THREADS>), grid, block, lds, st, A.nrows, A.I, A.a, A.lidx, A.xc_block_row, A.uoff, A.ucols, oI, cap, p);
    else if (xc_lean_enabled() && !p.cs && !p.f && !p.yin && !p.s && !p.s2 && !p.out2 && !p.out_perm && !p.out_add && !oI && p.alpha == 1.0)
        hipLaunchKernelGGL((spmv_xc_kernel<false, THREADS, true>), grid, block, lds, st, A.nrows, A.I, A.a, A.lidx, A.xc_block_row, A.uoff, A.ucols, oI, cap, p);
    else if (p.cs)
        hipLaunchKernelGGL((spmv_xc_kernel<true, THREADS>), grid, block, lds, st, A.nrows, A.I, A.a, A.lidx, A.xc_block_row, A.uoff, A.ucols, oI, cap, p);
    else
        hipLaunchKernelGGL((spmv_xc_kernel<false, THREADS>), grid, block, lds, st, A.nrows, A.I, A.a, A.lidx, A.xc_block_row, A.uoff, A.ucols, oI, cap, p);
    ELAG_LAUNCH_CHECK();
}

void launch_xc(const DevCSR &A, const DevCSR *offd, const SpmvArgs &p, hipStream_t st, bool probe = false)
{
    const int *oI = (p.skip_rows_with_offd && offd) ? offd->I : nullptr;
    if (A.xc_window > DevCSR::kStreamNnz) launch_xc_t<512>(A, oI, p, st, probe);   // 4096-entry windows: same entries per lane
    else launch_xc_t<256>(A, oI, p, st, probe);
}

void launch_stream(const DevCSR &A, const DevCSR *offd, const SpmvArgs &p, hipStream_t st)
{
    if (A.xc_state == 0 && xc_mode() != 0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cs);
        if (cs == hipStreamCaptureStatusNone) prepare_spmv(A);
    }
    if (A.xc_state == 1 && stream_variant() != 4) { launch_xc(A, offd, p, st); return; }
    // ELAG_HIP_STREAM_VARIANT=4 selects a DIAGNOSTIC build of the kernel (coalesced x reads, wrong results) that
    // prices the gather: 7.2 TB/s against 5.1-5.4 TB/s for the real kernel (CHANGELOG.md, round 2)
    if (stream_variant() == 4) launch_stream_v<4>(A, offd, p, st);
    else if (p.o2s && !p.cs) launch_stream_v<1>(A, offd, p, st);             // (a build of its own: the option's select in the common epilogue cost the x-cache kernel 7 %)
    else launch_stream_v<0>(A, offd, p, st);
}

}  // namespace

void preload_spmv_kernels()
{
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(stream_plan_kernel));
}

void build_stream_plan(DevCSR &A)
{
    hipStream_t st = ctx().stream;
    A.nblocks = (int)((A.nnz + DevCSR::kStreamNnz - 1) / DevCSR::kStreamNnz);
    if (A.nblocks < 1) A.nblocks = 1;
    A.block_row = dev_alloc<int>(2 * ((size_t)A.nblocks + 1) + 2);
    hipLaunchKernelGGL(stream_plan_kernel, dim3((unsigned)((A.nblocks + 1 + 255) / 256)), dim3(256), 0, st, A.nrows, (const int *)A.I,
                       A.nblocks, DevCSR::kStreamNnz, A.block_row);
    if (A.known_max_row > 0) {
        A.max_row_nnz = A.known_max_row;
        A.known_max_row = 0;
        ELAG_LAUNCH_CHECK();
        return;
    }
    Temp<int> m(1);
    zero_async(m, sizeof(int));
    if (A.nrows) {
        const unsigned g = (unsigned)std::min<long long>(((long long)A.nrows + 255) / 256, 512);
        hipLaunchKernelGGL(max_row_kernel, dim3(g), dim3(256), 0, st, A.nrows, (const int *)A.I, m.p);
    }
    ELAG_LAUNCH_CHECK();
    A.max_row_nnz = fetch_int(m.p);
}

void longest_row(int nrows, const int *I, int *d_out)
{
    hipStream_t st = ctx().stream;
    zero_async(d_out, sizeof(int));
    if (nrows) {
        const unsigned g = (unsigned)std::min<long long>(((long long)nrows + 255) / 256, 512);
        hipLaunchKernelGGL(max_row_kernel, dim3(g), dim3(256), 0, st, nrows, I, d_out);
    }
    ELAG_LAUNCH_CHECK();
}

namespace {

// builds the x-cache tables of A for `window` entries per workgroup; false when it does not apply (or, unforced, does not pay)
bool xc_build(const DevCSR &A, int window, bool force, double min_reuse = 2.0)
{
    const int cap = window + A.max_row_nnz + 8;
    if (cap > 8192) return false;                       // builder capacity; two LDS arrays of the kernel stay <= 64 KB + xs
    hipStream_t st = ctx().stream;
    int *plan = nullptr;
    int nb = A.nblocks;
    if (window == DevCSR::kStreamNnz) {
        plan = A.block_row;
    } else {
        nb = (int)((A.nnz + window - 1) / window);
        plan = dev_alloc<int>(2 * ((size_t)nb + 1) + 2);
        hipLaunchKernelGGL(stream_plan_kernel, dim3((unsigned)((nb + 1 + 255) / 256)), dim3(256), 0, st, A.nrows, (const int *)A.I, nb, window, plan);
        ELAG_LAUNCH_CHECK();
    }
    auto build = [&](int grid, int *ucount, const int *uoff, int *ucols, unsigned short *lidx, int fill, int step) {
        if (cap <= 4096)
            hipLaunchKernelGGL((xc_build_kernel<4096>), dim3((unsigned)grid), dim3(kBlock), 0, st, (const int *)plan, (const int *)A.J, ucount, uoff, ucols, lidx, fill, step);
        else
            hipLaunchKernelGGL((xc_build_kernel<8192>), dim3((unsigned)grid), dim3(kBlock), 0, st, (const int *)plan, (const int *)A.J, ucount, uoff, ucols, lidx, fill, step);
        ELAG_LAUNCH_CHECK();
    };
    auto drop_plan = [&] { if (plan != A.block_row) dev_free(plan); };
    // Does the form pay?  Entries per distinct column of a window, from a sample of up to 512 windows spread evenly over the operator (the
    // full count was a sorting pass over every window of its own -- and all of it wasted on the operators that fail the rule: the
    // caller's-numbering operator and P^T at every Finalize)
    double reuse = 0.0;
    if (!force) {
        const int step = std::max(1, nb / 512), ns = (nb + step - 1) / step;
        Temp<int> sample((size_t)2 * ns);
        build(ns, sample.p, nullptr, nullptr, nullptr, 0, step);
        std::vector<int> hs = host_from_dev((const int *)sample.p, (size_t)2 * ns);
        long long distinct = 0, entries = 0;
        for (int b = 0; b < ns; b++) { distinct += hs[(size_t)2 * b]; entries += hs[(size_t)2 * b + 1]; }
        reuse = distinct ? (double)entries / (double)distinct : 0.0;
        if (reuse < min_reuse) {
            if (getenv("ELAG_HIP_XC_VERBOSE"))
                fprintf(stderr, "[elag_hip] x-cache: n=%d nnz=%lld window=%d: reuse %.2f over %d sampled windows, stream kernel kept\n", A.nrows, (long long)A.nnz, window, reuse, ns);
            drop_plan();
            return false;
        }
    }
    // one sorting pass: positions (final) and every window's distinct columns into a slot of its own, then scan the counts and compact
    int *uoff = dev_alloc<int>((size_t)nb + 1);
    unsigned short *lidx = dev_alloc<unsigned short>((size_t)A.nnz + 8);
    zero_async(lidx + A.nnz, 8 * sizeof(unsigned short));
    int total = 0, mx = 0;
    int *ucols = nullptr;
    {
        Temp<int> ucount((size_t)nb), slots((size_t)nb * (size_t)cap);
        build(nb, ucount.p, nullptr, slots.p, lidx, 2, cap);
        exclusive_scan(ucount, uoff, nb);
        std::vector<int> ho = host_from_dev((const int *)uoff, (size_t)nb + 1);
        total = ho[(size_t)nb];
        for (int c = 0; c < nb; c++) mx = std::max(mx, ho[(size_t)c + 1] - ho[(size_t)c]);
        if (mx <= 65535) {
            ucols = dev_alloc<int>((size_t)std::max(total, 1));
            hipLaunchKernelGGL(xc_compact_kernel, dim3((unsigned)std::min(nb, 4096)), dim3(kBlock), 0, st, nb, cap, (const int *)slots.p, (const int *)uoff, ucols);
            ELAG_LAUNCH_CHECK();                     // (no drain: the scratch is reused in stream order)
        }
    }
    if (getenv("ELAG_HIP_XC_VERBOSE"))
        fprintf(stderr, "[elag_hip] x-cache: n=%d nnz=%lld window=%d windows=%d unique/window avg %.1f max %d reuse %.2f (sampled %.2f)\n", A.nrows, (long long)A.nnz,
                window, nb, (double)total / nb, mx, total ? (double)A.nnz / total : 0.0, reuse);
    if (mx > 65535) {
        dev_free(uoff); dev_free(lidx);
        drop_plan();
        return false;
    }
    A.xc_max_unique = (mx + 7) & ~7;
    A.uoff = uoff;
    A.ucols = ucols;
    A.lidx = lidx;
    A.xc_block_row = plan; A.xc_nblocks = nb; A.xc_window = window;
    A.xc_state = 1;
    return true;
}

int64_t xc_min_nnz()
{
    static int64_t v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_XC_MIN_NNZ"); v = e ? atoll(e) : (int64_t(1) << 21); }
    return v;
}

int xc_window_env()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("ELAG_HIP_XC_WINDOW"); v = e ? atoi(e) : 0; }   // 0 = measure 2048 and 4096
    return v;
}

}  // namespace

void prepare_spmv(const DevCSR &A, bool force)
{
    if (A.xc_state == -2) A.xc_state = 0;                   // (declined by a Finalize that expected few cycles: this caller asks)
    if (A.xc_state != 0 || (xc_mode() == 0 && !force)) return;
    A.xc_state = -1;
    if (A.lanes_per_row != 0 || !A.block_row || A.nnz == 0 || A.nblocks == 0) return;
    // Operators of fewer than 2^21 entries are not measured (ELAG_HIP_XC_MIN_NNZ): they are launch / latency-bound, and a measurement
    // costs three sorted window plans.  From there up the measurement decides: with windows that are not powers of two the form also
    // wins on operators resident in the Infinity Cache (cube456 x5 level 1, 4.8 M entries: 13.5 against 15.0 us; round 1 found the
    // opposite with 2048-entry windows).
    if (!force && xc_mode() < 0 && A.nnz < xc_min_nnz()) return;
    hipStream_t st = ctx().stream;
    if (force || xc_mode() >= 0) {
        const int w = xc_window_env() ? xc_window_env() : 2304;      // forced without a measurement: the window that measures best on MI355X
        if (!xc_build(A, w, force)) A.xc_state = -1;
        return;
    }
    // Heuristic mode.  Round 2 let a measurement decide at every Finalize -- the plain stream kernel against x-cache windows of 2048, 2304
    // and 1664 entries, three sorted window plans and ~40 launches per operator, 29 ms of a 70 ms setup at cube456 x5 -- and every
    // operator it ever measured came out the same way (tools/xc_reuse_probe.py): the 2304-entry window is the best of the three wherever
    // the form wins, and the form wins exactly where a window's entries share their columns well -- entries per distinct column 4.6-6.9
    // on the renumbered level operators and on P (x-cache faster by 7-15 %), 2.9-3.1 on the same operators in the caller's numbering
    // (stream kernel faster by 3-7 %), 1.4 on P^T.  So the rule is the reuse itself, counted by the first pass of the one plan that
    // would be built anyway: x-cache / 2304 from 4 entries per distinct column, the stream kernel below.  ELAG_HIP_XC_MEASURE=1 keeps
    // the measurement (A/B runs).
    static int measure = -1;
    if (measure < 0) { const char *e = getenv("ELAG_HIP_XC_MEASURE"); measure = (e && atoi(e)) ? 1 : 0; }
    if (!measure) {
        const int w = xc_window_env() ? xc_window_env() : 2304;
        static double min_reuse = -1.0;
        if (min_reuse < 0.0) { const char *e = getenv("ELAG_HIP_XC_MIN_REUSE"); min_reuse = e ? atof(e) : 4.0; }
        if (!xc_build(A, w, false, min_reuse)) A.xc_state = -1;
        return;
    }
    // A few launches of each candidate on a scratch vector pair (HIP events on this stream, under kernel names of their own): the plain
    // stream kernel and the x-cache form with three window sizes; the x-cache form is kept only when it is measurably faster.
    Temp<double> tx((size_t)std::max(A.ncols, A.nrows)), ty((size_t)A.nrows);
    zero_async(tx, sizeof(double) * (size_t)std::max(A.ncols, A.nrows));
    SpmvArgs p;
    p.x = tx; p.out = ty;
    hipEvent_t e0, e1;
    ELAG_HIP_CHECK(hipEventCreate(&e0));
    ELAG_HIP_CHECK(hipEventCreate(&e1));
    auto timed = [&](auto &&launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            for (int w = 0; w < 2; w++) launch();
            ELAG_HIP_CHECK(hipEventRecord(e0, st));
            for (int w = 0; w < 8; w++) launch();
            ELAG_LAUNCH_CHECK();
            ELAG_HIP_CHECK(hipEventRecord(e1, st));
            event_wait(e1);
            float ms = 0.f;
            ELAG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 8);
        }
        return best;
    };
    const size_t lds_s = sizeof(double) * (size_t)(DevCSR::kStreamNnz + A.max_row_nnz + 8);
    const float t_stream = timed([&] {
        hipLaunchKernelGGL(spmv_probe_stream_kernel, dim3((unsigned)A.nblocks), dim3(kBlock), lds_s, st, A.nrows, A.I, A.J, A.a, A.block_row,
                           (const int *)nullptr, p);
    });
    float t_best = t_stream;
    int w_best = 0;
    // Window sizes: 2048 (the stream kernel's) and two that are NOT powers of two.  Consecutive workgroups start their value / position
    // streams one window apart, and with 2^k entries per window those starts are 2^k bytes apart too -- the same few HBM channels for
    // every workgroup in flight: measured on cube456 x5, 94-95 us with 2048-entry windows, 101 us with 4096, 86-89 us with 1536 ... 2816
    // that are not powers of two (profiles/r02_spmv_xc_window_scan.txt).
    const int candidates[3] = {DevCSR::kStreamNnz, 2304, 1664};
    float t_w[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < 3; c++) {
        if (xc_window_env() && xc_window_env() != candidates[c]) continue;
        if (!xc_build(A, candidates[c], false)) { A.xc_state = -1; continue; }
        t_w[c] = timed([&] { launch_xc(A, nullptr, p, st, true); });
        if (t_w[c] < 0.99f * t_best) { t_best = t_w[c]; w_best = candidates[c]; }
        A.release_xc();
        A.xc_state = -1;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (getenv("ELAG_HIP_XC_VERBOSE"))
        fprintf(stderr, "[elag_hip] SpMV forms on this operator: stream %.2f us, x-cache/%d %.2f us, x-cache/%d %.2f us, x-cache/%d %.2f us -> %s%d\n",
                t_stream * 1e3, candidates[0], t_w[0] * 1e3, candidates[1], t_w[1] * 1e3, candidates[2], t_w[2] * 1e3, w_best == 0 ? "stream " : "x-cache/",
                w_best);
    if (w_best && !xc_build(A, w_best, false)) A.xc_state = -1;
}

bool prepare_sign_form(const DevCSR &A)
{
    if (A.sign_state != 0) return A.sign_state == 1;
    A.sign_state = -1;
    if (A.nnz == 0 || A.ncols >= (1 << 30)) return false;
    hipStream_t st = ctx().stream;
    int *Js = dev_alloc<int>((size_t)A.nnz);
    Temp<int> bad(1);
    zero_async(bad, sizeof(int));
    const unsigned g = (unsigned)std::min<long long>((A.nnz + kBlock - 1) / kBlock, 4096);
    hipLaunchKernelGGL(sign_pack_kernel, dim3(g), dim3(kBlock), 0, st, (long long)A.nnz, (const int *)A.J, (const double *)A.a, Js, bad.p);
    ELAG_LAUNCH_CHECK();
    const int nbad = fetch_int(bad.p);
    if (nbad) { dev_free(Js); return false; }
    A.Js = Js;
    A.sign_state = 1;
    return true;
}

bool sign_twohop(const DevCSR &D, const DevCSR &DT, const double *wa, const double *pv, const double *xin, double *xout)
{
    // OFF unless ELAG_HIP_TWOHOP=1: measured on the cube456 x3 H(div) hierarchy the one launch is SLOWER than the two it replaces (cycle 0.146 against
    // 0.128 ms, 18 against 20 launches): five dependent round trips per row (row bounds, D's words, the transposed rows' bounds, their words, pv)
    // where each of the two launches has three, and the transposed rows are walked by one lane each.  Kept for A/B runs.
    static int on = -1;
    if (on < 0) { const char *e = getenv("ELAG_HIP_TWOHOP"); on = (e && atoi(e)) ? 1 : 0; }
    if (!on || D.sign_state != 1 || DT.sign_state != 1 || !D.Js || !DT.Js || DT.nrows != D.ncols || DT.ncols != D.nrows) return false;
    if (D.nrows == 0) return true;
    const unsigned grid = (unsigned)(((long long)D.nrows * 4 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((sign_twohop_kernel<4>), dim3(grid), dim3(kBlock), 0, ctx().stream, D.nrows, (const int *)D.I, (const int *)D.Js, (const int *)DT.I, (const int *)DT.Js, wa, pv,
                       xin, xout);
    ELAG_LAUNCH_CHECK();
    return true;
}

void spmv_fused(const DevCSR &A, const DevCSR *offd, const SpmvArgs &args, hipStream_t st)
{
    if (!st) st = ctx().stream;
    ELAG_REQUIRE(!args.out_add || args.out_perm, "internal: out_add needs out_perm (the epilogues index it with the permuted row)");
    ELAG_REQUIRE(!args.o2s || (A.xc_state != 1 && A.lanes_per_row == 0 && A.block_row && args.rows == nullptr && !(offd && offd->nnz > 0 && !args.skip_rows_with_offd) &&
                               A.sign_state != 1 && xc_mode() != 1),
                 "internal: o2s is an option of the plain stream kernel");
    const bool rowlist = args.rows != nullptr;
    const bool offd_terms = offd && offd->nnz > 0 && !args.skip_rows_with_offd;
    if (A.sign_state == 1 && !rowlist && !offd_terms && !args.skip_rows_with_offd) {
        const double avg = A.nrows ? (double)A.nnz / A.nrows : 0.0;
        if (avg <= 6.0) launch_sign<1>(A, args, st);
        else if (avg <= 12.0) launch_sign<2>(A, args, st);
        else if (avg <= 24.0) launch_sign<4>(A, args, st);
        else launch_sign<8>(A, args, st);
        return;
    }
    if (A.lanes_per_row == 0 && !rowlist && !offd_terms && A.block_row) {
        if (A.nrows) launch_stream(A, offd, args, st);
        return;
    }
    int lanes = A.lanes_per_row;
    if (lanes == 0) {  // row subsets / offd terms use the row-group kernel
        const double avg = A.nrows ? (double)(A.nnz + (offd ? offd->nnz : 0)) / A.nrows : 0.0;
        lanes = 1;
        while (lanes < 64 && 2.0 * lanes < avg) lanes <<= 1;
    }
    switch (lanes) {
    case 1: launch_lpr<1>(A, offd, args, st); break;
    case 2: launch_lpr<2>(A, offd, args, st); break;
    case 4: launch_lpr<4>(A, offd, args, st); break;
    case 8: launch_lpr<8>(A, offd, args, st); break;
    case 16: launch_lpr<16>(A, offd, args, st); break;
    case 32: launch_lpr<32>(A, offd, args, st); break;
    default: launch_lpr<64>(A, offd, args, st); break;
    }
}

// ================================================================================================
// Resident sequence: a dependent chain of launch-bound products (the fused level kernels of a small hierarchy and its dense
// tail) as ONE kernel.  Every product alone takes 4-5 us as a graph node whatever its size; here the workgroups of one launch
// (at most two per CU, so that all of them are resident together) walk the list of products, each taking rows grid-stride, with a
// grid-wide barrier in between: release fence, one atomic arrival per workgroup, the last one opens the next generation,
// acquire fence.  The row arithmetic is spmv_fused_row / the dense row kernels', lane for lane, so the results are those of the
// separate launches bit for bit.  Every wait is bounded: a workgroup that does not see the barrier open within ~2^24 polls
// sets the error flag and leaves (so does, then, everyone else).
// ================================================================================================
namespace {

__device__ __forceinline__ bool grid_barrier(unsigned *bar, int *err)
{
    // every wave waits for its own stores; the cache maintenance of the release / acquire (L2 write-back, L1 / L2 invalidate: the
    // caches belong to the CU and the XCD, not to the wave) is done once per workgroup, by the wave that arrives at the barrier
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int polls = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > (1 << 24)) { ok = 0; __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok != 0;
}

inline const double *resident_in_host(const double *p, const double *rhs, const double *sol)
{
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    return v == kResidentRhs ? rhs : (v == kResidentSol ? sol : p);
}

__device__ __forceinline__ const double *resident_in(const double *p, const double *rhs, const double *sol)
{
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    return v == kResidentRhs ? rhs : (v == kResidentSol ? sol : p);
}

template <int LPR>
__device__ __forceinline__ void resident_rows(const ResidentOp &o)
{
    constexpr int GPB = kBlock / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    for (int row = blockIdx.x * GPB + threadIdx.x / LPR; row < o.nrows; row += gridDim.x * GPB) {
        if (o.has_offd) {
            if (o.has_cs) spmv_fused_row<LPR, true, true>(row, lane, o.I, o.J, o.a, o.oI, o.oJ, o.oa, o.p);
            else spmv_fused_row<LPR, true, false>(row, lane, o.I, o.J, o.a, o.oI, o.oJ, o.oa, o.p);
        } else {
            if (o.has_cs) spmv_fused_row<LPR, false, true>(row, lane, o.I, o.J, o.a, o.oI, o.oJ, o.oa, o.p);
            else spmv_fused_row<LPR, false, false>(row, lane, o.I, o.J, o.a, o.oI, o.oJ, o.oa, o.p);
        }
    }
}

__global__ __launch_bounds__(kBlock) void resident_kernel(const ResidentOp *__restrict__ ops, int nops, const double *rhs, double *sol, unsigned *bar,
                                                          int *err)
{
    __shared__ double sh[kBlock / 64];
    for (int i = 0; i < nops; i++) {
        ResidentOp o = ops[i];
        o.p.x = resident_in(o.p.x, rhs, sol); o.p.xg = resident_in(o.p.xg, rhs, sol);
        o.p.f = resident_in(o.p.f, rhs, sol); o.p.yin = resident_in(o.p.yin, rhs, sol);
        o.p.out = const_cast<double *>(resident_in(o.p.out, rhs, sol));
        o.p.out2 = const_cast<double *>(resident_in(o.p.out2, rhs, sol));
        if (o.kind == 0) {
            switch (o.lpr) {
            case 1: resident_rows<1>(o); break;
            case 2: resident_rows<2>(o); break;
            case 4: resident_rows<4>(o); break;
            case 8: resident_rows<8>(o); break;
            case 16: resident_rows<16>(o); break;
            case 32: resident_rows<32>(o); break;
            default: resident_rows<64>(o); break;
            }
        } else if (o.kind == 1) {
            // dense_gemv_row_kernel: a workgroup per row
            const int n = o.ncols;
            const double *x = o.p.x;
            for (int row = blockIdx.x; row < o.nrows; row += gridDim.x) {
                const double *m = o.M + (size_t)row * n;
                double s = 0.0;
                for (int j0 = threadIdx.x; j0 < n; j0 += 4 * kBlock) {
                    const int j1 = j0 + kBlock, j2 = j0 + 2 * kBlock, j3 = j0 + 3 * kBlock;
                    const double a0 = m[j0], b0 = x[j0];
                    const double a1 = j1 < n ? m[j1] : 0.0, b1 = j1 < n ? x[j1] : 0.0;
                    const double a2 = j2 < n ? m[j2] : 0.0, b2 = j2 < n ? x[j2] : 0.0;
                    const double a3 = j3 < n ? m[j3] : 0.0, b3 = j3 < n ? x[j3] : 0.0;
                    s += a0 * b0; s += a1 * b1; s += a2 * b2; s += a3 * b3;
                }
                s = group_reduce<64>(s);
                if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
                __syncthreads();
                if (threadIdx.x == 0) o.p.out[row] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
                __syncthreads();
            }
        } else {
            // dense_gemv_kernel: a wavefront per row
            const int n = o.ncols, lane = threadIdx.x & 63;
            const double *x = o.p.x;
            for (int row = (blockIdx.x * kBlock + threadIdx.x) >> 6; row < o.nrows; row += (gridDim.x * kBlock) >> 6) {
                const double *m = o.M + (size_t)row * n;
                double s = 0.0;
                for (int j = lane; j < n; j += 64) s += m[j] * x[j];
                s = group_reduce<64>(s);
                if (lane == 0) o.p.out[row] = s;
            }
        }
        if (i + 1 < nops && !grid_barrier(bar, err)) return;
    }
}

}  // namespace

bool resident_op_spmv(const DevCSR &A, const DevCSR *offd, const SpmvArgs &args, ResidentOp &op)
{
    // the launch spmv_fused would make must be the row-group kernel
    const bool offd_terms = offd && offd->nnz > 0 && !args.skip_rows_with_offd;
    if (args.rows || args.out_perm || args.out_add || args.skip_rows_with_offd) return false;
    if (A.sign_state == 1 && !offd_terms) return false;
    if (A.lanes_per_row == 0 && !offd_terms && A.block_row) return false;
    int lanes = A.lanes_per_row;
    if (lanes == 0) {
        const double avg = A.nrows ? (double)(A.nnz + (offd ? offd->nnz : 0)) / A.nrows : 0.0;
        lanes = 1;
        while (lanes < 64 && 2.0 * lanes < avg) lanes <<= 1;
    }
    op = ResidentOp();
    op.kind = 0; op.lpr = lanes; op.has_offd = offd_terms ? 1 : 0; op.has_cs = args.cs ? 1 : 0;
    op.nrows = A.nrows; op.ncols = A.ncols;
    op.I = A.I; op.J = A.J; op.a = A.a;
    op.oI = offd ? offd->I : nullptr; op.oJ = offd ? offd->J : nullptr; op.oa = offd ? offd->a : nullptr;
    op.p = args;
    return true;
}

void resident_op_dense(int nrows, int ncols, const double *M, const double *x, double *y, ResidentOp &op)
{
    op = ResidentOp();
    op.kind = ncols >= 256 ? 1 : 2;           // the choice of dense_gemv
    op.nrows = nrows; op.ncols = ncols; op.M = M;
    op.p.x = x; op.p.out = y;
}

void resident_build(const std::vector<ResidentOp> &ops, ResidentSeq &seq)
{
    resident_free(seq);
    if (ops.empty()) return;
    seq.nops = (int)ops.size();
    seq.d_ops = reinterpret_cast<ResidentOp *>(dev_alloc<char>(sizeof(ResidentOp) * ops.size()));
    stage_h2d(seq.d_ops, ops.data(), sizeof(ResidentOp) * ops.size());
    seq.d_bar = dev_alloc<unsigned>(2);
    seq.d_err = dev_alloc<int>(1);
    zero_async(seq.d_bar, 2 * sizeof(unsigned));
    zero_async(seq.d_err, sizeof(int));
    ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    // workgroups: what the widest product can use, at most two per CU (all resident together: 8 waves per CU)
    long long want = 1;
    for (const ResidentOp &o : ops) {
        const long long w = o.kind == 0 ? ((long long)o.nrows * o.lpr + kBlock - 1) / kBlock : (o.kind == 1 ? o.nrows : ((long long)o.nrows * 64 + kBlock - 1) / kBlock);
        want = std::max(want, w);
    }
    const int cus = ctx().num_cus > 0 ? ctx().num_cus : 64;
    long long cap = cus / 2;                       // (measured best on MI355X: 128 of 256 CUs -- more arrivals cost more than they compute)
    if (const char *e = getenv("ELAG_HIP_RESIDENT_WGS")) cap = std::max(1, std::min(atoi(e), 2 * cus));
    seq.grid = (unsigned)std::min<long long>(want, cap);
}

void resident_free(ResidentSeq &seq)
{
    if (seq.d_ops) dev_free(reinterpret_cast<char *&>(seq.d_ops));
    dev_free(seq.d_bar); dev_free(seq.d_err);
    seq = ResidentSeq();
}

void resident_launch(const ResidentSeq &seq, const double *rhs, double *sol, hipStream_t st)
{
    if (!st) st = ctx().stream;
    hipLaunchKernelGGL(resident_kernel, dim3(seq.grid), dim3(kBlock), 0, st, (const ResidentOp *)seq.d_ops, seq.nops, rhs, sol, seq.d_bar, seq.d_err);
    ELAG_LAUNCH_CHECK();
}

int resident_error(const ResidentSeq &seq)
{
    if (!seq.d_err) return 0;
    int e = 0;
    ELAG_HIP_CHECK(hipMemcpy(&e, seq.d_err, sizeof(int), hipMemcpyDeviceToHost));
    return e;
}

// ================================================================================================
// Flat stages: the independent products of one dependency stage of a flattened cycle in ONE launch (see kernels.hpp).
// ================================================================================================
namespace {

// one CSR row by LPR lanes, kFlatU entries per lane and pass: all the column / value loads of a pass are requested before the first
// gather, all the gathers before the first product.  These operators sit in L2 / Infinity Cache and a row is a chain of round trips,
// so the time of a launch is (entries) / (entries in flight per CU): four entries per lane instead of one let a wave cover 4 x 64
// entries per trip (cube456 x3, stage 1 of the flattened cycle: 8.8 -> X us).
constexpr int kFlatU = 4;
template <int LPR, int U = kFlatU, typename JT = int>
__device__ __forceinline__ double flat_row_sum(int s, int e, int lane, const JT *__restrict__ J, const double *__restrict__ a, const double *__restrict__ x)
{
    double sum = 0.0;
    for (int k0 = s + lane; k0 < e; k0 += U * LPR) {
        int j[U];
        double av[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + u * LPR;
            j[u] = 0; av[u] = 0.0;
            if (k < e) { j[u] = (int)J[k]; av[u] = a[k]; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            xv[u] = 0.0;
            if (k0 + u * LPR < e) xv[u] = x[j[u]];
        }
#pragma unroll
        for (int u = 0; u < U; u++) sum += av[u] * xv[u];
    }
    return sum;
}

// the same with the row's first U * LPR entries taken from the operator's flat pack (DevCSR::fpJ / fpa: addressed from the row number alone, so
// their loads leave with the row pointers' instead of behind them -- a stage is a chain of dependent round trips and this removes one for every
// row that fits the pack); entries past the pack come from the CSR arrays as before.  Padding slots hold column 0 / value 0.0: they add +-0.0.
template <int LPR, int U>
__device__ __forceinline__ double flat_row_sum_packed(int row, int lane, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                                      const int *__restrict__ pJ, const double *__restrict__ pa, const double *__restrict__ x)
{
    int j[U];
    double av[U], xv[U];
    const size_t base = ((size_t)row * U) * LPR + lane;
#pragma unroll
    for (int u = 0; u < U; u++) { j[u] = pJ[base + (size_t)u * LPR]; av[u] = pa[base + (size_t)u * LPR]; }
    const int s = I[row], e = I[row + 1];
#pragma unroll
    for (int u = 0; u < U; u++) xv[u] = x[j[u]];
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < U; u++) sum += av[u] * xv[u];
    if (e - s > U * LPR) sum += flat_row_sum<LPR, U>(s + U * LPR, e, lane, J, a, x);
    return sum;
}

template <int LPR, int U, bool J16>
__device__ __forceinline__ void flat_seg_rows(unsigned blk, int nrows, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                              const int *__restrict__ pJ, const double *__restrict__ pa, const double *__restrict__ x, double *__restrict__ out)
{
    constexpr int GPB = kBlock / LPR;
    const int row = (int)blk * GPB + (int)threadIdx.x / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= nrows) return;
    const double part = J16 ? flat_row_sum<LPR, U, unsigned short>(I[row], I[row + 1], lane, reinterpret_cast<const unsigned short *>(J), a, x)
                        : pJ ? flat_row_sum_packed<LPR, U>(row, lane, I, J, a, pJ, pa, x) : flat_row_sum<LPR, U>(I[row], I[row + 1], lane, J, a, x);
    const double sum = group_reduce<LPR>(part);
    if (lane == 0) out[row] = sum;
}

// J16: the CSR segments' J arrays hold 16-bit words (every operator of the stage has at most 65 536 columns): 10 bytes per entry instead of 12
template <int U, bool J16 = false>
__global__ __launch_bounds__(kBlock) void flat_segs_kernel(const FlatSegs S)
{
    __shared__ double sh[kBlock / 64];
    // the segment of this workgroup: selects on values that all arrive with the kernel arguments, no dependent load
    unsigned bb = 0;
    int nrows = S.nrows[0], lpr = S.lpr[0], ncols = S.ncols[0];
    const int *I = S.I[0], *J = S.J[0], *pJ = S.pJ[0];
    const double *a = S.a[0], *x = S.x[0], *pa = S.pa[0];
    double *out = S.out[0];
#pragma unroll
    for (int i = 1; i < kFlatSegs; i++)
        if (i < S.nsegs && blockIdx.x >= S.block_begin[i]) {
            bb = S.block_begin[i]; nrows = S.nrows[i]; lpr = S.lpr[i]; ncols = S.ncols[i];
            I = S.I[i]; J = S.J[i]; a = S.a[i]; x = S.x[i]; out = S.out[i]; pJ = S.pJ[i]; pa = S.pa[i];
        }
    const unsigned blk = blockIdx.x - bb;
    switch (lpr) {
    case 0: {
        // dense row: every lane requests its (up to 4) matrix and vector entries before the first use (dense_gemv_row_kernel)
        const int n = ncols;
        const double *__restrict__ m = a + (size_t)blk * n;
        double s = 0.0;
        for (int j0 = threadIdx.x; j0 < n; j0 += 4 * kBlock) {
            const int j1 = j0 + kBlock, j2 = j0 + 2 * kBlock, j3 = j0 + 3 * kBlock;
            const double a0 = m[j0], b0 = x[j0];
            const double a1 = j1 < n ? m[j1] : 0.0, b1 = j1 < n ? x[j1] : 0.0;
            const double a2 = j2 < n ? m[j2] : 0.0, b2 = j2 < n ? x[j2] : 0.0;
            const double a3 = j3 < n ? m[j3] : 0.0, b3 = j3 < n ? x[j3] : 0.0;
            s += a0 * b0; s += a1 * b1; s += a2 * b2; s += a3 * b3;
        }
        s = group_reduce<64>(s);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[blk] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        break;
    }
    case 1: flat_seg_rows<1, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    case 2: flat_seg_rows<2, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    case 4: flat_seg_rows<4, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    case 8: flat_seg_rows<8, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    case 16: flat_seg_rows<16, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    case 32: flat_seg_rows<32, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    case 64: flat_seg_rows<64, U, J16>(blk, nrows, I, J, a, pJ, pa, x, out); break;
    default: {
        // a workgroup per (long) row
        double s = J16 ? flat_row_sum<kBlock, U, unsigned short>(I[blk], I[blk + 1], (int)threadIdx.x, reinterpret_cast<const unsigned short *>(J), a, x)
                       : flat_row_sum<kBlock, U>(I[blk], I[blk + 1], (int)threadIdx.x, J, a, x);
        s = group_reduce<64>(s);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[blk] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        break;
    }
    }
}

template <int LPR, int NT, int U, bool J16 = false>
__global__ __launch_bounds__(kBlock) void flat_sum_kernel(const FlatSum S)
{
    constexpr int GPB = kBlock / LPR;
    const int row = (int)blockIdx.x * GPB + (int)threadIdx.x / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= S.nrows) return;
    // every term's row bounds are requested together; then term after term, kFlatU entries per lane in flight
    int s[NT], e[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) { s[t] = S.I[t][row]; e[t] = S.I[t][row + 1]; }
    double y0 = 0.0;
    if (lane == 0 && S.yin) y0 = S.yin[row];
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NT; t++)
        sum += J16 ? flat_row_sum<LPR, U, unsigned short>(s[t], e[t], lane, reinterpret_cast<const unsigned short *>(S.J[t]), S.a[t], S.x[t])
                   : flat_row_sum<LPR, U>(s[t], e[t], lane, S.J[t], S.a[t], S.x[t]);
    sum = group_reduce<LPR>(sum);
    if (lane == 0) S.out[row] = y0 + sum;
}

template <int NT, int U, bool J16 = false>
void flat_sum_dispatch(const FlatSum &S, hipStream_t st)
{
    switch (S.lpr) {
    case 1: hipLaunchKernelGGL((flat_sum_kernel<1, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    case 2: hipLaunchKernelGGL((flat_sum_kernel<2, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    case 4: hipLaunchKernelGGL((flat_sum_kernel<4, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    case 8: hipLaunchKernelGGL((flat_sum_kernel<8, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    case 16: hipLaunchKernelGGL((flat_sum_kernel<16, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    case 32: hipLaunchKernelGGL((flat_sum_kernel<32, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    default: hipLaunchKernelGGL((flat_sum_kernel<64, NT, U, J16>), dim3(S.grid), dim3(kBlock), 0, st, S); break;
    }
}

// lanes for rows of the given average length: the smallest power of two >= avg / 2; short of 2^16 lanes in all, the chain of a row is
// the price and rows get up to as many lanes as they have entries (the rule of DevCSR::choose_spmv_shape)
int flat_u()
{
    static int u = 0;
    if (u == 0) { const char *e = getenv("ELAG_HIP_FLAT_U"); u = e ? atoi(e) : kFlatU; if (u != 2 && u != 8) u = 4; }
    return u;
}

int flat_lanes(int nrows, double avg)
{
    static double div = 0.0;
    static long long min_lanes = 0;
    if (div == 0.0) {
        const char *e = getenv("ELAG_HIP_FLAT_LANE_DIV"); div = e && atof(e) > 0 ? atof(e) : (double)flat_u();
        const char *m = getenv("ELAG_HIP_FLAT_MIN_LANES"); min_lanes = m && atoll(m) > 0 ? atoll(m) : 32768;
    }
    int l = 1;
    while (l < 64 && div * l < avg) l <<= 1;
    while (l < 64 && (long long)nrows * l < min_lanes && l < avg) l <<= 1;
    return l;
}

}  // namespace

namespace {
__global__ void flat_pack_kernel(int nrows, int L, int U, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                 int *__restrict__ pJ, double *__restrict__ pa)
{
    const long long tot = (long long)nrows * L * U;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(i % L), u = (int)((i / L) % U), row = (int)(i / ((long long)L * U));
        const int k = I[row] + u * L + lane;
        const bool has = k < I[row + 1];
        pJ[i] = has ? J[k] : 0;
        pa[i] = has ? a[k] : 0.0;
    }
}
__global__ void j16_kernel(long long nnz, const int *__restrict__ J, unsigned short *__restrict__ J16)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (long long)gridDim.x * blockDim.x) J16[i] = (unsigned short)J[i];
}
// the 16-bit copy of an operator's column indices (null where it has more than 65 536 columns, or with ELAG_HIP_FLAT_J16=0)
const unsigned short *flat_j16(const DevCSR &A)
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("ELAG_HIP_FLAT_J16"); on = e ? (atoi(e) != 0) : 1; }
    if (!on || A.ncols > 65536 || A.nnz == 0) return nullptr;
    if (!A.J16) {
        A.J16 = dev_alloc<unsigned short>((size_t)A.nnz + 8);
        hipLaunchKernelGGL(j16_kernel, dim3((unsigned)std::min<long long>((A.nnz + 255) / 256, 4096)), dim3(256), 0, ctx().stream, (long long)A.nnz, (const int *)A.J, A.J16);
        ELAG_LAUNCH_CHECK();
    }
    return A.J16;
}
bool flat_pack_enabled()
{
    static int v = -1;
    // OFF unless ELAG_HIP_FLAT_PACK=1: measured on MI355X (cube456 x3 headline, stages 1 and 2 packed) 20.3 us per cycle against 16.8 -- the
    // padded passes move more bytes than the dependent round trip they remove is worth
    if (v < 0) { const char *e = getenv("ELAG_HIP_FLAT_PACK"); v = e ? (atoi(e) != 0) : 0; }
    return v != 0;
}
}  // namespace

void flat_segs_add_csr(FlatSegs &S, const DevCSR &A, const double *x, double *out)
{
    ELAG_REQUIRE(S.nsegs < kFlatSegs, "internal: too many segments in a flat stage");
    const int i = S.nsegs++;
    const double avg = A.nrows ? (double)A.nnz / A.nrows : 0.0;
    const int l = avg > 64.0 * flat_u() ? kBlock : flat_lanes(A.nrows, avg);
    S.block_begin[i] = S.grid;
    S.nrows[i] = A.nrows; S.ncols[i] = A.ncols; S.lpr[i] = l;
    S.I[i] = A.I; S.J[i] = A.J; S.a[i] = A.a; S.x[i] = x; S.out[i] = out;
    S.pJ[i] = nullptr; S.pa[i] = nullptr;
    S.j16p[i] = flat_j16(A);
    if (flat_pack_enabled() && l >= 1 && l <= 64 && A.nrows > 0 && A.nnz > 0 && A.ncols > 0) {
        // the rows' first passes in lane order (DevCSR::fpJ): once per operator and lane count
        const int U = flat_u();
        if (!(A.fpJ && A.fpL == l && A.fpU == U)) {
            dev_free(A.fpJ); dev_free(A.fpa);
            const size_t tot = (size_t)A.nrows * l * U;
            A.fpJ = dev_alloc<int>(tot);
            A.fpa = dev_alloc<double>(tot);
            A.fpL = l; A.fpU = U;
            hipLaunchKernelGGL(flat_pack_kernel, dim3((unsigned)std::min<size_t>((tot + 255) / 256, 4096)), dim3(256), 0, ctx().stream, A.nrows, l, U, (const int *)A.I,
                               (const int *)A.J, (const double *)A.a, A.fpJ, A.fpa);
            ELAG_LAUNCH_CHECK();
        }
        S.pJ[i] = A.fpJ; S.pa[i] = A.fpa;
    }
    S.grid += l == kBlock ? (unsigned)A.nrows : (unsigned)(((long long)A.nrows * l + kBlock - 1) / kBlock);
}

void flat_segs_add_dense(FlatSegs &S, int nrows, int ncols, const double *M, const double *x, double *out)
{
    ELAG_REQUIRE(S.nsegs < kFlatSegs, "internal: too many segments in a flat stage");
    const int i = S.nsegs++;
    S.block_begin[i] = S.grid;
    S.nrows[i] = nrows; S.ncols[i] = ncols; S.lpr[i] = 0;
    S.a[i] = M; S.x[i] = x; S.out[i] = out;
    S.grid += (unsigned)nrows;
}

void flat_sum_add(FlatSum &S, const DevCSR &A, const double *x)
{
    ELAG_REQUIRE(S.nterms < kFlatTerms, "internal: too many terms in a flat sum");
    ELAG_REQUIRE(S.nterms == 0 || S.nrows == A.nrows, "internal: flat sum over operators of different heights");
    const int t = S.nterms++;
    S.nrows = A.nrows;
    S.I[t] = A.I; S.J[t] = A.J; S.a[t] = A.a; S.x[t] = x;
    S.j16p[t] = flat_j16(A);
    S.grid += (unsigned)A.nnz;          // (entry count until flat_sum_finish)
}

void flat_sum_finish(FlatSum &S, const double *yin, double *out)
{
    const double avg = S.nrows ? (double)S.grid / S.nrows : 0.0;
    S.lpr = flat_lanes(S.nrows, avg);
    S.grid = (unsigned)(((long long)S.nrows * S.lpr + kBlock - 1) / kBlock);
    S.yin = yin; S.out = out;
}

void flat_segs_launch(const FlatSegs &S0, const double *rhs, double *sol, hipStream_t st)
{
    if (!st) st = ctx().stream;
    if (S0.grid == 0) return;
    FlatSegs S = S0;
    for (int i = 0; i < S.nsegs; i++) {
        S.x[i] = resident_in_host(S.x[i], rhs, sol);
        S.out[i] = const_cast<double *>(resident_in_host(S.out[i], rhs, sol));
    }
    // 16-bit columns where every CSR segment of the stage has them (and no flat pack is in use: that form keeps 32-bit columns)
    bool j16 = flat_u() == 4;
    for (int i = 0; i < S.nsegs; i++)
        if (S.lpr[i] != 0 && (!S.j16p[i] || S.pJ[i])) j16 = false;
    if (j16) {
        for (int i = 0; i < S.nsegs; i++)
            if (S.lpr[i] != 0) S.J[i] = reinterpret_cast<const int *>(S.j16p[i]);
        hipLaunchKernelGGL((flat_segs_kernel<4, true>), dim3(S.grid), dim3(kBlock), 0, st, S);
        ELAG_LAUNCH_CHECK();
        return;
    }
    switch (flat_u()) {
    case 2: hipLaunchKernelGGL(flat_segs_kernel<2>, dim3(S.grid), dim3(kBlock), 0, st, S); break;
    case 8: hipLaunchKernelGGL(flat_segs_kernel<8>, dim3(S.grid), dim3(kBlock), 0, st, S); break;
    default: hipLaunchKernelGGL(flat_segs_kernel<4>, dim3(S.grid), dim3(kBlock), 0, st, S); break;
    }
    ELAG_LAUNCH_CHECK();
}

void flat_sum_launch(const FlatSum &S0, const double *rhs, double *sol, hipStream_t st)
{
    if (!st) st = ctx().stream;
    if (S0.grid == 0) return;
    FlatSum S = S0;
    for (int t = 0; t < S.nterms; t++) S.x[t] = resident_in_host(S.x[t], rhs, sol);
    S.yin = resident_in_host(S.yin, rhs, sol);
    S.out = const_cast<double *>(resident_in_host(S.out, rhs, sol));
    const int u = flat_u();
    bool j16 = u == 4;
    for (int t = 0; t < S.nterms; t++)
        if (!S.j16p[t]) j16 = false;
    if (j16) {
        for (int t = 0; t < S.nterms; t++) S.J[t] = reinterpret_cast<const int *>(S.j16p[t]);
        switch (S.nterms) {
        case 1: flat_sum_dispatch<1, 4, true>(S, st); break;
        case 2: flat_sum_dispatch<2, 4, true>(S, st); break;
        case 3: flat_sum_dispatch<3, 4, true>(S, st); break;
        default: flat_sum_dispatch<4, 4, true>(S, st); break;
        }
        ELAG_LAUNCH_CHECK();
        return;
    }
#define ELAG_FLAT_SUM(NT) (u == 2 ? flat_sum_dispatch<NT, 2>(S, st) : u == 8 ? flat_sum_dispatch<NT, 8>(S, st) : flat_sum_dispatch<NT, 4>(S, st))
    switch (S.nterms) {
    case 1: ELAG_FLAT_SUM(1); break;
    case 2: ELAG_FLAT_SUM(2); break;
    case 3: ELAG_FLAT_SUM(3); break;
    default: ELAG_FLAT_SUM(4); break;
    }
#undef ELAG_FLAT_SUM
    ELAG_LAUNCH_CHECK();
}

// ================================================================================================
// vector kernels: grid-stride, 2 doubles per lane per access where alignment allows
// ================================================================================================
namespace {

inline unsigned vec_grid(long long n)
{
    long long g = (n + kBlock - 1) / kBlock;
    const long long cap = 2048;  // ~8 blocks per CU, grid-stride beyond
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

__global__ void axpby_kernel(int n, double a, const double *__restrict__ x, double b, const double *__restrict__ y,
                             double *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double t = 0.0;
        if (x) t = a * x[i];
        if (y) t += b * y[i];
        out[i] = t;
    }
}

__global__ void scale_mul_kernel(int n, double a, const double *__restrict__ x, const double *__restrict__ d,
                                 double *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        out[i] = a * x[i] * (d ? d[i] : 1.0);
}

__global__ void add_scaled_mul_kernel(int n, const double *__restrict__ y, const double *__restrict__ d,
                                      const double *__restrict__ z, double c, const double *__restrict__ r,
                                      double *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double t = z ? z[i] : 0.0;
        if (r) t += c * r[i];
        out[i] = (y ? y[i] : 0.0) + d[i] * t;
    }
}

__global__ void fill_kernel(int n, double v, double *__restrict__ x)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = v;
}

__global__ void gather_kernel(int n, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ buf)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = x[idx[i]];
}

__global__ void gather_scale_kernel(int n, const int *__restrict__ idx, const double *__restrict__ x, const double *__restrict__ d,
                                    double *__restrict__ buf, double *__restrict__ scaled)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = x[idx[i]];
        buf[i] = v;
        scaled[i] = 1.0 * v * d[i];          // same expression as scale_mul_kernel with a = 1
    }
}

__global__ void scatter_kernel(int n, const int *__restrict__ idx, const double *__restrict__ src, double *__restrict__ dst, int add)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = idx[i];
        dst[j] = add ? dst[j] + src[i] : src[i];
    }
}

__device__ __forceinline__ double block_reduce_sum(double v, double *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) r += sh[i];
    return r;  // valid on thread 0
}

__device__ __forceinline__ double block_reduce_max(double v, double *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sh[w] = v;
    __syncthreads();
    double r = v;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) r = fmax(r, sh[i]);
    return r;
}

constexpr int kRedBlocks = 512;

// mode 0: sum x*y ; 1: sum x*x ; 2: max |x| ; 3: max x
__global__ __launch_bounds__(kBlock) void reduce_stage1(long long n, const double *__restrict__ x,
                                                        const double *__restrict__ y, int mode, double *__restrict__ part)
{
    __shared__ double sh[kBlock / 64];
    double acc = (mode == 3) ? -1.0e308 : 0.0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const double v = x[i];
        if (mode == 0) acc += v * y[i];
        else if (mode == 1) acc += v * v;
        else if (mode == 2) acc = fmax(acc, fabs(v));
        else acc = fmax(acc, v);
    }
    const double r = (mode >= 2) ? block_reduce_max(acc, sh) : block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) part[blockIdx.x] = r;
}

__global__ __launch_bounds__(kBlock) void reduce_stage2(int nparts, const double *__restrict__ part, int mode,
                                                        double *__restrict__ result)
{
    __shared__ double sh[kBlock / 64];
    double acc = (mode == 3) ? -1.0e308 : 0.0;
    for (int i = threadIdx.x; i < nparts; i += kBlock) acc = (mode >= 2) ? fmax(acc, part[i]) : acc + part[i];
    const double r = (mode >= 2) ? block_reduce_max(acc, sh) : block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) result[0] = r;
}

double *reduce_scratch()
{
    Context &c = ctx();
    if (!c.reduce_part) c.reduce_part = dev_alloc<double>(kRedBlocks);
    return c.reduce_part;
}

void reduce_launch(long long n, const double *x, const double *y, int mode, double *d_result)
{
    hipStream_t st = ctx().stream;
    long long g = (n + kBlock - 1) / kBlock;
    const int grid = (int)(g < 1 ? 1 : (g > kRedBlocks ? kRedBlocks : g));
    double *part = reduce_scratch();
    hipLaunchKernelGGL(reduce_stage1, dim3(grid), dim3(kBlock), 0, st, n, x, y, mode, part);
    hipLaunchKernelGGL(reduce_stage2, dim3(1), dim3(kBlock), 0, st, grid, part, mode, d_result);
    ELAG_LAUNCH_CHECK();
}

double reduce_to_host(long long n, const double *x, int mode)
{
    if (n == 0) return 0.0;
    Context &c = ctx();
    reduce_launch(n, x, nullptr, mode, c.d_scalar);
    double v = 0.0;
    fetch_doubles(c.d_scalar, 1, &v);
    return v;
}

}  // namespace

void vec_axpby(int n, double a, const double *x, double b, const double *y, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(axpby_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, a, x, b, y, out);
    ELAG_LAUNCH_CHECK();
}
void vec_scale_mul(int n, double a, const double *x, const double *d, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(scale_mul_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, a, x, d, out);
    ELAG_LAUNCH_CHECK();
}
void vec_add_scaled_mul(int n, const double *y, const double *d, const double *z, double c, const double *r, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(add_scaled_mul_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, y, d, z, c, r, out);
    ELAG_LAUNCH_CHECK();
}
void vec_fill(int n, double v, double *x)
{
    if (n == 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, v, x);
    ELAG_LAUNCH_CHECK();
}
void vec_gather(int n, const int *idx, const double *x, double *buf, hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, st ? st : ctx().stream, n, idx, x, buf);
    ELAG_LAUNCH_CHECK();
}
namespace {
__global__ void pack_copy_kernel(int n, int nsend, const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ send_buf, double *__restrict__ copy)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + nsend; i += gridDim.x * blockDim.x) {
        if (i < nsend) send_buf[i] = x[idx[i]];
        else copy[i - nsend] = x[i - nsend];
    }
}
}  // namespace
void vec_pack_copy(int n, int nsend, const int *idx, const double *x, double *send_buf, double *copy)
{
    if (n + nsend == 0) return;
    hipLaunchKernelGGL(pack_copy_kernel, dim3(vec_grid(n + nsend)), dim3(kBlock), 0, ctx().stream, n, nsend, idx, x, send_buf, copy);
    ELAG_LAUNCH_CHECK();
}
void vec_gather_scale(int n, const int *idx, const double *x, const double *d, double *buf, double *scaled)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_scale_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, idx, x, d, buf, scaled);
    ELAG_LAUNCH_CHECK();
}
void vec_scatter(int n, const int *idx, const double *src, double *dst, int add)
{
    if (n == 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, idx, src, dst, add);
    ELAG_LAUNCH_CHECK();
}
void vec_dot(int n, const double *x, const double *y, double *d_result)
{
    if (n == 0) { zero_async(d_result, sizeof(double)); return; }
    reduce_launch(n, x, y, 0, d_result);
}

void vec_hypre_random(int n, int seed, double *out)
{
    // hypre_SeqVectorSetRandomValues: sequential Lehmer stream (a = 16807, m = 2^31 - 1), value 2 r - 1.
    // Generated on the host: the recurrence is inherently serial and runs once per Chebyshev setup.
    std::vector<double> h((size_t)n);
    const long long a = 16807, m = 2147483647;
    long long s = seed;
    for (int i = 0; i < n; i++) { s = (a * s) % m; h[(size_t)i] = 2.0 * ((double)s / (double)m) - 1.0; }
    if (n) {
        stage_h2d(out, h.data(), sizeof(double) * (size_t)n);
        ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    }
}

double reduce_max_abs(int64_t n, const double *a) { return reduce_to_host(n, a, 2); }
double reduce_sum_sq(int64_t n, const double *a) { return reduce_to_host(n, a, 1); }
double reduce_max(int64_t n, const double *a) { return reduce_to_host(n, a, 3); }

// ================================================================================================
// row statistics
// ================================================================================================
namespace {

// one 8-lane group per row
__global__ __launch_bounds__(kBlock) void l1_norms_kernel(int n, const int *__restrict__ I, const int *__restrict__ J,
                                                          const double *__restrict__ a, const int *__restrict__ oI,
                                                          const double *__restrict__ oa, int option,
                                                          double *__restrict__ l1, int *__restrict__ bad)
{
    constexpr int LPR = 8;
    const int row = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= n) return;
    double sd = 0.0, so = 0.0, dneg = 0.0;
    for (int k = I[row] + lane; k < I[row + 1]; k += LPR) {
        const double v = a[k];
        const bool isd = (J[k] == row);
        if (isd && v < 0.0) dneg = 1.0;
        if (option == 1) sd += fabs(v);
        else if (option == 3) sd += v * v;
        else if (isd) sd += fabs(v);  // options 2, 4
    }
    if (oI)
        for (int k = oI[row] + lane; k < oI[row + 1]; k += LPR) {
            const double v = oa[k];
            so += (option == 3) ? v * v : fabs(v);
        }
    sd = group_reduce<LPR>(sd);
    so = group_reduce<LPR>(so);
    dneg = group_reduce<LPR>(dneg);
    if (lane == 0) {
        double r;
        if (option == 4) { r = sd + 0.5 * so; if (r <= 4.0 / 3.0 * sd) r = sd; }
        else r = sd + so;
        if (dneg > 0.0) r = -r;
        if (fabs(r) < 2.220446049250313e-16) atomicAdd(bad, 1);
        l1[row] = r;
    }
}

__global__ __launch_bounds__(kBlock) void extract_diag_kernel(int n, const int *__restrict__ I, const int *__restrict__ J,
                                                              const double *__restrict__ a, double *__restrict__ d)
{
    constexpr int LPR = 8;
    const int row = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= n) return;
    double v = 0.0;
    for (int k = I[row] + lane; k < I[row + 1]; k += LPR)
        if (J[k] == row) v += a[k];
    v = group_reduce<LPR>(v);
    if (lane == 0) d[row] = v;
}

__global__ void rsqrt_kernel(int n, const double *__restrict__ d, double *__restrict__ out, int *__restrict__ bad)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = d[i];
        if (!(v > 0.0)) atomicAdd(bad, 1);
        out[i] = 1.0 / sqrt(v);
    }
}

__global__ void recip_scale_kernel(int n, double w, const double *__restrict__ d, double *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = w / d[i];
}

__global__ __launch_bounds__(kBlock) void fix_zero_rows_kernel(int n, const int *__restrict__ I, const int *__restrict__ J,
                                                               double *__restrict__ a, const int *__restrict__ oI,
                                                               double *__restrict__ oa, int *__restrict__ nfixed)
{
    constexpr int LPR = 8;
    const int row = (blockIdx.x * kBlock + threadIdx.x) / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    if (row >= n) return;
    double l1 = 0.0;
    for (int k = I[row] + lane; k < I[row + 1]; k += LPR) l1 += fabs(a[k]);
    if (oI)
        for (int k = oI[row] + lane; k < oI[row + 1]; k += LPR) l1 += fabs(oa[k]);
    l1 = group_reduce_all<LPR>(l1);
    if (l1 <= 0.0) {
        for (int k = I[row] + lane; k < I[row + 1]; k += LPR) a[k] = (J[k] == row) ? 1.0 : 0.0;
        if (oI)
            for (int k = oI[row] + lane; k < oI[row + 1]; k += LPR) oa[k] = 0.0;
        if (lane == 0) atomicAdd(nfixed, 1);
    }
}

__global__ void abs_kernel(long long n, const double *__restrict__ a, double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = fabs(a[i]);
}

__global__ void sign_kernel(long long n, double *__restrict__ a)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double v = a[i];
        a[i] = (fabs(v) > 1e-9) ? ((v > 0.) ? 1.0 : -1.0) : 0.0;
    }
}

inline unsigned row_grid(int n, int lpr) { return (unsigned)(((long long)n * lpr + kBlock - 1) / kBlock); }

int read_counter(int *d_cnt)
{
    return fetch_int(d_cnt);
}

}  // namespace

void l1_norms(const DevCSR &diag, const DevCSR *offd, int option, double *l1, int *d_bad)
{
    if (diag.nrows == 0) return;
    const bool ho = offd && offd->nnz > 0;
    hipLaunchKernelGGL(l1_norms_kernel, dim3(row_grid(diag.nrows, 8)), dim3(kBlock), 0, ctx().stream, diag.nrows,
                       diag.I, diag.J, diag.a, ho ? offd->I : nullptr, ho ? offd->a : nullptr, option, l1, d_bad);
    ELAG_LAUNCH_CHECK();
}

void extract_diagonal(const DevCSR &diag, double *d)
{
    if (diag.nrows == 0) return;
    hipLaunchKernelGGL(extract_diag_kernel, dim3(row_grid(diag.nrows, 8)), dim3(kBlock), 0, ctx().stream, diag.nrows,
                       diag.I, diag.J, diag.a, d);
    ELAG_LAUNCH_CHECK();
}

void vec_rsqrt(int n, const double *d, double *out, int *d_bad)
{
    if (n == 0) return;
    hipLaunchKernelGGL(rsqrt_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, d, out, d_bad);
    ELAG_LAUNCH_CHECK();
}

void vec_reciprocal_scale(int n, double w, const double *d, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(recip_scale_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, n, w, d, out);
    ELAG_LAUNCH_CHECK();
}

int fix_zero_rows(DevCSR &diag, DevCSR *offd, bool count)
{
    if (diag.nrows == 0) return 0;
    diag.release_sign();
    if (offd) offd->release_sign();
    Temp<int> cnt(1);
    zero_async(cnt, sizeof(int));
    const bool ho = offd && offd->nnz > 0;
    hipLaunchKernelGGL(fix_zero_rows_kernel, dim3(row_grid(diag.nrows, 8)), dim3(kBlock), 0, ctx().stream, diag.nrows,
                       diag.I, diag.J, diag.a, ho ? offd->I : nullptr, ho ? offd->a : nullptr, cnt.p);
    ELAG_LAUNCH_CHECK();
    return count ? read_counter(cnt) : 0;            // (a hierarchy build does not ask: one round trip less per level)
}

void values_abs(int64_t n, const double *a, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(abs_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, (long long)n, a, out);
    ELAG_LAUNCH_CHECK();
}

void values_sign(int64_t n, double *a)
{
    if (n == 0) return;
    hipLaunchKernelGGL(sign_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, ctx().stream, (long long)n, a);
    ELAG_LAUNCH_CHECK();
}

// ================================================================================================
// dense coarse apply: y = Minv x, one wavefront per row (n <= 4096; Minv stays in L2)
// ================================================================================================
namespace {
__global__ __launch_bounds__(kBlock) void dense_gemv_kernel(int nrows, int n, const double *__restrict__ M, const double *__restrict__ x,
                                                            double *__restrict__ y)
{
    const int row = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (row >= nrows) return;
    const double *__restrict__ m = M + (size_t)row * n;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += m[j] * x[j];
    s = group_reduce<64>(s);
    if (lane == 0) y[row] = s;
}

// One workgroup per row: every lane requests its (up to 4 per pass) matrix and vector entries before the first use, so
// a row of <= 1024 entries costs one memory round trip instead of n/64 dependent ones (the coarse solve sits on the
// critical path of every V-cycle: 5.0 -> ~2 us at n = 840).
__global__ __launch_bounds__(kBlock) void dense_gemv_row_kernel(int n, const double *__restrict__ M, const double *__restrict__ x,
                                                                double *__restrict__ y)
{
    __shared__ double sh[kBlock / 64];
    const double *__restrict__ m = M + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int j0 = threadIdx.x; j0 < n; j0 += 4 * kBlock) {
        const int j1 = j0 + kBlock, j2 = j0 + 2 * kBlock, j3 = j0 + 3 * kBlock;
        const double a0 = m[j0], b0 = x[j0];
        const double a1 = j1 < n ? m[j1] : 0.0, b1 = j1 < n ? x[j1] : 0.0;
        const double a2 = j2 < n ? m[j2] : 0.0, b2 = j2 < n ? x[j2] : 0.0;
        const double a3 = j3 < n ? m[j3] : 0.0, b3 = j3 < n ? x[j3] : 0.0;
        s += a0 * b0; s += a1 * b1; s += a2 * b2; s += a3 * b3;
    }
    s = group_reduce<64>(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) y[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
}  // namespace

namespace {
__global__ void dense_transpose_kernel(int n, const double *__restrict__ in, double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int i = by + r, j = bx + threadIdx.x;
        if (i < n && j < n) tile[r][threadIdx.x] = in[(size_t)i * n + j];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int i = bx + r, j = by + threadIdx.x;
        if (i < n && j < n) out[(size_t)i * n + j] = tile[threadIdx.x][r];
    }
}
}  // namespace

void dense_transpose(int n, const double *in, double *out)
{
    if (n == 0) return;
    const unsigned g = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(dense_transpose_kernel, dim3(g, g), dim3(32, 8), 0, ctx().stream, n, in, out);
    ELAG_LAUNCH_CHECK();
}

void dense_gemv(int nrows, int ncols, const double *M, const double *x, double *y)
{
    if (nrows == 0) return;
    if (ncols >= 256) hipLaunchKernelGGL(dense_gemv_row_kernel, dim3((unsigned)nrows), dim3(kBlock), 0, ctx().stream, ncols, M, x, y);
    else hipLaunchKernelGGL(dense_gemv_kernel, dim3(row_grid(nrows, 64)), dim3(kBlock), 0, ctx().stream, nrows, ncols, M, x, y);
    ELAG_LAUNCH_CHECK();
}

}  // namespace k
}  // namespace elag
