// AMGe coarse-space construction for the (H(div), L2) pair: the per-agglomerate dense linear algebra of
// DeRhamSequence::Coarsen (src/amge/DeRhamSequence.cpp:572-692, jformStart_ = 2), batched -- one workgroup per
// agglomerated entity, all of an entity's matrices in LDS (global scratch for entities that do not fit).
//
//   coarsen_traces_kernel      ComputeCoarseTracesWithTargets, loop body (:1818-1926)
//   coarsen_extension_kernel   hFacetExtension, loop body (:2364-2556)
//
// The local systems are tiny (tet agglomerate of 8 children: saddle point of order 17 with 7 right-hand sides), so the
// kernels are latency/LDS bound, not MFMA work: a 16x16x4 f64 MFMA tile would be mostly padding.  Dense helpers below
// are cooperative over the workgroup with workgroup barriers between dependent phases.
//
// Differences from the reference that do not change results beyond rounding: Gaussian elimination with partial pivoting
// instead of LAPACK's Bunch-Kaufman LDL^T (dsytrf) and LU (dgetrf); a one-sided Jacobi (Hestenes) SVD run directly in the
// mass inner product instead of dgesvd on W^{1/2} A (same singular values / left singular subspaces; the sign of a
// singular vector is arbitrary in both).
#include "kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>

namespace elag {
namespace {

constexpr int kMaxCols = 40;   // 1 PV + up to 39 targets (order-2 targets of the vector-valued forms: 3 x 10 = 30)
constexpr int kLdsDoubles = 7936;   // 62 KiB of dynamic LDS per workgroup

struct AsmDev {
    const int64_t *asm_ptr, *blk_off, *loc_off;
    const int *blk_n, *loc_idx;
    const double *blocks;
};

struct Shared {
    double red[48];
    double sv[kMaxCols];
    int perm[kMaxCols];
    int piv[2];
    int cnt;
};

__device__ inline void zero_fill(double *p, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.0;
}

// M (n x n row-major) = sum of the entity's blocks, added block after block (deterministic order).
// Fast path (one wave per workgroup, blocks of at most 64 entries -- every finest level and most coarse ones): the
// descriptors of eight blocks are fetched by eight lanes and broadcast, the eight blocks' values and index pairs are
// requested together, then added in block order by plain LDS read-modify-writes (a single wave executes its LDS operations
// in order, so no barrier is needed and the summation order stays the sequential one).  The general path pays one global
// round trip and one barrier per block.
__device__ void assemble_blocks(double *M, int n, const AsmDev &A, int e)
{
    zero_fill(M, n * n);
    __syncthreads();
    const int64_t t0 = A.asm_ptr[e], t1 = A.asm_ptr[e + 1];
    if (blockDim.x == 64) {
        const int lane = threadIdx.x;
        for (int64_t tb = t0; tb < t1; tb += 8) {
            int my_m = 0;
            int64_t my_bo = 0, my_lo = 0;
            if (lane < 8 && tb + lane < t1) { my_m = A.blk_n[tb + lane]; my_bo = A.blk_off[tb + lane]; my_lo = A.loc_off[tb + lane]; }
            // all blocks of this chunk small?  (uniform decision)
            const int big = __any(my_m * my_m > 64);
            if (!big) {
                double v[8];
                int pos[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int m = __shfl(my_m, u, 64);
                    const int64_t bo = __shfl(my_bo, u, 64);
                    const int64_t lo = __shfl(my_lo, u, 64);
                    pos[u] = -1;
                    v[u] = 0.0;
                    if (lane < m * m) {
                        const int r = lane / m, c = lane - r * m;
                        v[u] = A.blocks[bo + lane];
                        pos[u] = A.loc_idx[lo + r] * n + A.loc_idx[lo + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++)
                    if (pos[u] >= 0) M[pos[u]] += v[u];
            } else {
                for (int u = 0; u < 8 && tb + u < t1; u++) {
                    const int m = __shfl(my_m, u, 64);
                    const double *b = A.blocks + __shfl(my_bo, u, 64);
                    const int *li = A.loc_idx + __shfl(my_lo, u, 64);
                    for (int i = lane; i < m * m; i += 64) {
                        const int r = i / m, c = i - r * m;
                        M[li[r] * n + li[c]] += b[i];
                    }
                }
            }
        }
        __syncthreads();
        return;
    }
    for (int64_t t = t0; t < t1; t++) {
        const int m = A.blk_n[t];
        const double *b = A.blocks + A.blk_off[t];
        const int *li = A.loc_idx + A.loc_off[t];
        for (int i = threadIdx.x; i < m * m; i += blockDim.x) {
            const int r = i / m, c = i - r * m;
            M[li[r] * n + li[c]] += b[i];
        }
        __syncthreads();
    }
}

__device__ void block_sum3(double &a, double &b, double &c, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        b += __shfl_down(b, o, 64);
        c += __shfl_down(c, o, 64);
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[3 * w] = a; red[3 * w + 1] = b; red[3 * w + 2] = c; }
    __syncthreads();
    a = b = c = 0.0;
    for (int i = 0; i < nw; i++) { a += red[3 * i]; b += red[3 * i + 1]; c += red[3 * i + 2]; }
}

// A: n x m row-major; the first n columns are the system, columns n..m-1 the right-hand sides, replaced by the
// solutions.  Partial pivoting (largest magnitude, lowest index on ties).  Returns 1 when a pivot is zero / not finite.
__device__ int gauss_solve(double *A, int n, int m, int *piv)
{
    for (int k = 0; k < n; k++) {
        if (threadIdx.x < 64) {
            double best = -1.0;
            int bi = k;
            for (int i = k + threadIdx.x; i < n; i += 64) {
                const double v = fabs(A[i * m + k]);
                if (v > best || !(v == v)) { best = v; bi = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_down(best, o, 64);
                const int oi = __shfl_down(bi, o, 64);
                if (!(best == best)) continue;
                if (!(ov == ov) || ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (threadIdx.x == 0) { piv[0] = bi; piv[1] = (best > 0.0 && best == best && best < 1e300) ? 0 : 1; }
        }
        __syncthreads();
        const int p = piv[0];
        if (piv[1]) return 1;
        if (p != k)
            for (int j = k + threadIdx.x; j < m; j += blockDim.x) {
                const double t = A[k * m + j];
                A[k * m + j] = A[p * m + j];
                A[p * m + j] = t;
            }
        __syncthreads();
        const double inv = 1.0 / A[k * m + k];
        const int w = m - k - 1, h = n - k - 1;
        for (int t = threadIdx.x; t < w * h; t += blockDim.x) {
            const int i = k + 1 + t / w, j = k + 1 + t % w;
            A[i * m + j] -= (A[i * m + k] * inv) * A[k * m + j];
        }
        __syncthreads();
    }
    const int nr = m - n;
    for (int k = n - 1; k >= 0; k--) {
        const double inv = 1.0 / A[k * m + k];
        for (int c = threadIdx.x; c < nr; c += blockDim.x) A[k * m + n + c] *= inv;
        __syncthreads();
        for (int t = threadIdx.x; t < k * nr; t += blockDim.x) {
            const int i = t / nr, c = t - i * nr;
            A[i * m + n + c] -= A[i * m + k] * A[k * m + n + c];
        }
        __syncthreads();
    }
    return 0;
}

// One-sided Jacobi: rotates the k columns (length n, col-major) of A until they are orthogonal in the inner product whose
// Gram data is carried by Y = W A (pass Y == A for the Euclidean product; Y is rotated along).  Afterwards sh.sv[j] is
// the norm of column j and sh.perm the column order by descending norm.  floor2: squared norm below which a column is noise
// (a small fraction of the caller's cut-off).
__device__ void jacobi_orthogonalise(double *A, double *Y, int n, int k, Shared &sh, double floor2)
{
    const bool same = (A == Y);
    for (int sweep = 0; sweep < 40; sweep++) {
        int rotated = 0;
        for (int p = 0; p < k - 1; p++)
            for (int q = p + 1; q < k; q++) {
                double a = 0.0, b = 0.0, g = 0.0;
                for (int i = threadIdx.x; i < n; i += blockDim.x) {
                    a += A[p * n + i] * Y[p * n + i];
                    b += A[q * n + i] * Y[q * n + i];
                    g += A[p * n + i] * Y[q * n + i];
                }
                block_sum3(a, b, g, sh.red);
                // (two columns whose norms are both far below the cut-off are rounding noise that will be dropped: rotating
                // them against each other would only keep the sweeps going)
                if (a > 0.0 && b > 0.0 && (a > floor2 || b > floor2) && fabs(g) > 1e-15 * sqrt(a * b)) {
                    rotated = 1;
                    const double zeta = (b - a) / (2.0 * g);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    for (int i = threadIdx.x; i < n; i += blockDim.x) {
                        const double ap = A[p * n + i], aq = A[q * n + i];
                        A[p * n + i] = c * ap - s * aq;
                        A[q * n + i] = s * ap + c * aq;
                        if (!same) {
                            const double yp = Y[p * n + i], yq = Y[q * n + i];
                            Y[p * n + i] = c * yp - s * yq;
                            Y[q * n + i] = s * yp + c * yq;
                        }
                    }
                }
                __syncthreads();
            }
        if (!rotated) break;
    }
    for (int j = 0; j < k; j++) {
        double a = 0.0, b = 0.0, g = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) a += A[j * n + i] * Y[j * n + i];
        block_sum3(a, b, g, sh.red);
        if (threadIdx.x == 0) sh.sv[j] = a > 0.0 ? sqrt(a) : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int j = 0; j < k; j++) sh.perm[j] = j;
        for (int i = 1; i < k; i++) {   // insertion sort, descending, stable
            const int pj = sh.perm[i];
            int j = i - 1;
            while (j >= 0 && sh.sv[sh.perm[j]] < sh.sv[pj]) { sh.perm[j + 1] = sh.perm[j]; j--; }
            sh.perm[j + 1] = pj;
        }
    }
    __syncthreads();
}

// Y (n x k col-major) = M (n x n row-major, leading dimension ld) * X (n x k col-major)
__device__ void mat_times_cols(const double *M, int ld, const double *X, double *Y, int n, int k)
{
    for (int t = threadIdx.x; t < n * k; t += blockDim.x) {
        const int c = t / n, i = t - c * n;
        double s = 0.0;
        for (int j = 0; j < n; j++) s += M[i * ld + j] * X[c * n + j];
        Y[c * n + i] = s;
    }
}

// F (kc x n row-major, ld_out) = (P^T M P)^-1 (M P)^T ; P, MP: n x kc col-major; aug: kc x (kc+n) scratch.
// cm (may be null): symmetrised P^T M P, row-major with leading dimension ld_cm.
__device__ int dof_functional(const double *P, const double *MP, int n, int kc, double *aug, double *F, double *cm, int ld_cm, int *piv)
{
    const int m = kc + n;
    for (int t = threadIdx.x; t < kc * kc; t += blockDim.x) {
        const int a = t / kc, b = t - a * kc;
        double s = 0.0;
        for (int i = 0; i < n; i++) s += P[a * n + i] * MP[b * n + i];
        aug[a * m + b] = s;
    }
    for (int t = threadIdx.x; t < kc * n; t += blockDim.x) {
        const int a = t / n, i = t - a * n;
        aug[a * m + kc + i] = MP[a * n + i];
    }
    __syncthreads();
    if (cm)
        for (int t = threadIdx.x; t < kc * kc; t += blockDim.x) {
            const int a = t / kc, b = t - a * kc;
            cm[a * ld_cm + b] = 0.5 * (aug[a * m + b] + aug[b * m + a]);
        }
    __syncthreads();
    const int bad = gauss_solve(aug, kc, m, piv);
    if (bad) return 1;
    for (int t = threadIdx.x; t < kc * n; t += blockDim.x) {
        const int a = t / n, i = t - a * n;
        F[a * n + i] = aug[a * m + kc + i];
    }
    return 0;
}

struct TracesArgs {
    const int *ent_list;          // entities of this launch (nullptr: all of them, blockIdx.x is the entity)
    int n_ent, nT;
    double svd_tol;
    const int *ndof;
    const int64_t *dof_off;       // prefix sums of ndof
    AsmDev mass;
    const double *pv, *T;
    int *ncoarse;
    double *p_loc, *cmass, *functional;
    const int64_t *ws_off;        // >= 0: offset into gws (entity does not fit in LDS)
    double *gws;
    int *singular;
};

__global__ void coarsen_traces_kernel(TracesArgs a)
{
    extern __shared__ double dyn[];
    __shared__ Shared sh;
    const int e = a.ent_list ? a.ent_list[blockIdx.x] : (int)blockIdx.x;
    const int n = a.ndof[e], k0 = a.nT, kmax = 1 + k0;
    const int64_t doff = a.dof_off[e];
    double *ws = a.ws_off[e] >= 0 ? a.gws + a.ws_off[e] : dyn;
    double *M = ws;
    double *Tm = M + (size_t)n * n;          // n x k0
    double *Y = Tm + (size_t)n * k0;         // n x k0
    double *pv = Y + (size_t)n * k0;         // n
    double *Mpv = pv + n;                    // n
    double *P = Mpv + n;                     // n x kmax
    double *MP = P + (size_t)n * kmax;       // n x kmax
    double *aug = MP + (size_t)n * kmax;     // kmax x (kmax + n)

    assemble_blocks(M, n, a.mass, e);
    for (int i = threadIdx.x; i < n; i += blockDim.x) pv[i] = a.pv[doff + i];
    for (int i = threadIdx.x; i < n * k0; i += blockDim.x) Tm[i] = a.T[doff * k0 + i];
    __syncthreads();
    mat_times_cols(M, n, pv, Mpv, n, 1);
    __syncthreads();
    double pvpv = 0.0, d1 = 0.0, d2 = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) pvpv += pv[i] * Mpv[i];
    block_sum3(pvpv, d1, d2, sh.red);
    // Deflate (InnerProduct.cpp:157-169): t <- t - pv (pv, t)_M / (pv, pv)_M
    for (int c = 0; c < k0; c++) {
        double s = 0.0;
        d1 = d2 = 0.0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) s += Mpv[i] * Tm[c * n + i];
        block_sum3(s, d1, d2, sh.red);
        const double f = s * (-1.0 / pvpv);
        for (int i = threadIdx.x; i < n; i += blockDim.x) Tm[c * n + i] += f * pv[i];
    }
    __syncthreads();
    mat_times_cols(M, n, Tm, Y, n, k0);
    __syncthreads();
    { const double cut = pvpv * a.svd_tol * 1e-3; jacobi_orthogonalise(Tm, Y, n, k0, sh, cut * cut); }
    // keep the directions whose singular value is not below (pv,pv)_M * tol (:1875-1882)
    if (threadIdx.x == 0) {
        const double tol = pvpv * a.svd_tol;
        int k = 0;
        const int kcap = k0 < n ? k0 : n;
        while (k < kcap && !(sh.sv[sh.perm[k]] < tol)) k++;
        sh.cnt = k;
    }
    __syncthreads();
    const int k = sh.cnt, kc = 1 + k;
    const double sq = sqrt(pvpv);
    for (int i = threadIdx.x; i < n; i += blockDim.x) { P[i] = pv[i]; MP[i] = Mpv[i]; }
    for (int t = threadIdx.x; t < n * k; t += blockDim.x) {
        const int c = t / n, i = t - c * n, src = sh.perm[c];
        const double f = sq / sh.sv[src];
        P[(c + 1) * n + i] = Tm[src * n + i] * f;
        MP[(c + 1) * n + i] = Y[src * n + i] * f;
    }
    __syncthreads();
    double *op = a.p_loc + doff * kmax;
    for (int t = threadIdx.x; t < n * kmax; t += blockDim.x) op[t] = (t < n * kc) ? P[t] : 0.0;
    double *ocm = a.cmass + (size_t)e * kmax * kmax;
    for (int t = threadIdx.x; t < kmax * kmax; t += blockDim.x) ocm[t] = 0.0;
    double *of = a.functional + doff * kmax;
    for (int t = threadIdx.x; t < n * kmax; t += blockDim.x) of[t] = 0.0;
    __syncthreads();
    const int bad = dof_functional(P, MP, n, kc, aug, of, ocm, kmax, sh.piv);
    if (threadIdx.x == 0) {
        a.ncoarse[e] = kc;
        if (bad) atomicAdd(a.singular, 1);
    }
}

struct ExtArgs {
    const int *ent_list;          // entities of this launch (nullptr: all of them)
    int n_ent, nT, mode;          // mode 0: FacetSaddlePoint (PV multiplier row), 1: RidgePeakSaddlePoint ([M B^T; B -C])
    int null_dofs;                // 0: no null-space dofs are built (the "peak" stage of the 0-forms)
    int debug_stop;               // diagnostic builds (-DELAG_AMGE_PHASE_STOP): leave the kernel after phase k (ELAG_HIP_AMGE_STOP)
    double svd_tol, smallest_entry;
    const int *nu, *ni, *np, *npi, *nq, *next, *nrt;
    AsmDev mass_u, mass_p, mass_q;
    const int64_t *d_ptr;
    const int *d_row, *d_col;
    const double *d_val;
    const int64_t *d2_ptr;
    const int *d2_row, *d2_col;
    const double *d2_val;
    const double *pv_p, *R_b, *cP, *T, *PDc;
    // packed input / output offsets per entity
    const int64_t *off_p, *off_Rb, *off_cP, *off_T, *off_PDc, *off_ext, *off_dc, *off_rt, *off_nul, *off_fun, *off_cm;
    int *n_null;
    double *ext, *dc, *rt, *nul, *functional, *cmass;
    const int64_t *ws_off;
    double *gws;
    int *singular;
};

__global__ void coarsen_extension_kernel(ExtArgs a)
{
    extern __shared__ double dyn[];
    __shared__ Shared sh;
    const int e = a.ent_list ? a.ent_list[blockIdx.x] : (int)blockIdx.x;
    const int nu = a.nu[e], ni = a.ni[e], np = a.np[e], npi = a.npi[e], next = a.next[e], nrt = a.nrt[e];
    const int nq = a.mode ? a.nq[e] : 0;
    const int nb = nu - ni;
    const int nTe = (a.null_dofs && ni > nrt && a.nT > 0) ? a.nT : 0;          // :2451, :2930
    const int n = ni + npi + (a.mode ? 0 : 1), nrhs = next + nrt + nTe, m = n + nrhs;
    const int ncolmax = next + nrt + a.nT, kfmax = nrt + a.nT;
    const bool solve = a.mode == 0 || ni > 0;                                   // :2778 guards the ridge / peak solves only
    double *ws = a.ws_off[e] >= 0 ? a.gws + a.ws_off[e] : dyn;
    double *Maa = ws;                               // nu x nu
    double *W = Maa + (size_t)nu * nu;              // np x np
    double *Dl = W + (size_t)np * np;               // np x nu
    double *B = Dl + (size_t)np * nu;               // npi x nu
    double *tloc = B + (size_t)npi * nu;            // np
    double *K = tloc + np;                          // n x (n + next + nrt + nT)
    double *ti = K + (size_t)n * (n + next + nrt + a.nT);   // ni x nT
    double *basis = ti + (size_t)ni * a.nT;         // nu x ncolmax
    double *MB = basis + (size_t)nu * ncolmax;      // nu x ncolmax
    double *aug = MB + (size_t)nu * ncolmax;        // kfmax x (kfmax + ni)
    double *Pi = aug + (size_t)kfmax * (kfmax + ni); // 2 x (ni x kfmax): interior bubbles and M_ii times them
    double *W2 = Pi + 2 * (size_t)ni * kfmax;       // nq x nq        (mode 1)
    double *D2l = W2 + (size_t)nq * nq;             // nq x np
    double *WD2 = D2l + (size_t)nq * np;            // nq x np

    const double *pv_p = a.pv_p + a.off_p[e];
    const double *R_b = a.R_b + a.off_Rb[e];
    const double *cP = a.cP + a.off_cP[e];
    const double *T = a.T + a.off_T[e];
    const double *PDc = a.mode ? a.PDc + a.off_PDc[e] : nullptr;

    assemble_blocks(Maa, nu, a.mass_u, e);
    assemble_blocks(W, np, a.mass_p, e);
    zero_fill(Dl, np * nu);
    if (a.mode) {
        assemble_blocks(W2, nq, a.mass_q, e);
        zero_fill(D2l, nq * np);
    }
    __syncthreads();
#ifdef ELAG_AMGE_PHASE_STOP
    if (a.debug_stop == 1) return;
#endif
    for (int64_t t = a.d_ptr[e] + threadIdx.x; t < a.d_ptr[e + 1]; t += blockDim.x) Dl[a.d_row[t] * nu + a.d_col[t]] = a.d_val[t];
    if (a.mode)
        for (int64_t t = a.d2_ptr[e] + threadIdx.x; t < a.d2_ptr[e + 1]; t += blockDim.x) D2l[a.d2_row[t] * np + a.d2_col[t]] = a.d2_val[t];
    __syncthreads();
    // B_d = W_d D_d (:2258, :2690), rows of the interior p dofs; tloc = W pv (SaddlePointSolver.cpp:104-108)
    for (int t = threadIdx.x; t < npi * nu; t += blockDim.x) {
        const int p = t / nu, j = t - p * nu;
        double s = 0.0;
        for (int q = 0; q < np; q++) s += W[p * np + q] * Dl[q * nu + j];
        B[t] = s;
    }
    if (a.mode == 0)
        for (int p = threadIdx.x; p < np; p += blockDim.x) {
            double s = 0.0;
            for (int q = 0; q < np; q++) s += W[p * np + q] * pv_p[q];
            tloc[p] = s;
        }
    else
        for (int t = threadIdx.x; t < nq * np; t += blockDim.x) {             // W2 D2 (GetMinusC, :2583-2606)
            const int q = t / np, p = t - q * np;
            double s = 0.0;
            for (int r = 0; r < nq; r++) s += W2[q * nq + r] * D2l[r * np + p];
            WD2[t] = s;
        }
    zero_fill(K, n * m);
    __syncthreads();
    for (int t = threadIdx.x; t < ni * ni; t += blockDim.x) { const int i = t / ni, j = t - i * ni; K[i * m + j] = Maa[i * nu + j]; }
    for (int t = threadIdx.x; t < npi * ni; t += blockDim.x) {
        const int p = t / ni, j = t - p * ni;
        const double v = B[p * nu + j];
        K[(ni + p) * m + j] = v;
        K[j * m + ni + p] = v;
    }
    if (a.mode == 0) {
        for (int p = threadIdx.x; p < np; p += blockDim.x) { K[(n - 1) * m + ni + p] = tloc[p]; K[(ni + p) * m + n - 1] = tloc[p]; }
    } else {
        for (int t = threadIdx.x; t < npi * npi; t += blockDim.x) {           // -C = -(D2^T W2 D2), interior block
            const int p = t / npi, p2 = t - p * npi;
            double s = 0.0;
            for (int q = 0; q < nq; q++) s += D2l[q * np + p] * WD2[q * np + p2];
            K[(ni + p) * m + ni + p2] = -s;
        }
    }
    // right-hand sides: trace extensions (:2386-2390, :2812-2823), RangeT bubbles (:2428-2431), target extensions (:2469-2476)
    for (int t = threadIdx.x; t < (ni + npi) * next; t += blockDim.x) {
        const int c = t / (ni + npi), r = t - c * (ni + npi);
        const double *row = r < ni ? Maa + (size_t)r * nu + ni : B + (size_t)(r - ni) * nu + ni;
        double s = 0.0;
        for (int b = 0; b < nb; b++) s += row[b] * R_b[c * nb + b];
        s = -s;
        if (a.mode && r >= ni)
            for (int q = 0; q < np; q++) s += W[(r - ni) * np + q] * PDc[c * np + q];
        K[r * m + n + c] = s;
    }
    for (int t = threadIdx.x; t < npi * nrt; t += blockDim.x) {
        const int c = t / npi, p = t - c * npi;
        double s = 0.0;
        for (int q = 0; q < npi; q++) s += W[p * np + q] * cP[c * npi + q];
        K[(ni + p) * m + n + next + c] = s;
    }
    for (int t = threadIdx.x; t < (ni + npi) * nTe; t += blockDim.x) {
        const int c = t / (ni + npi), r = t - c * (ni + npi);
        double s = 0.0;
        if (r < ni) {
            for (int b = 0; b < nb; b++) s -= Maa[r * nu + ni + b] * T[c * nu + ni + b];
        } else {
            for (int j = 0; j < ni; j++) s += B[(r - ni) * nu + j] * T[c * nu + j];
        }
        K[r * m + n + next + nrt + c] = s;
    }
    __syncthreads();
#ifdef ELAG_AMGE_PHASE_STOP
    if (a.debug_stop == 2) return;
#endif
    int bad = solve ? gauss_solve(K, n, m, sh.piv) : 0;
#ifdef ELAG_AMGE_PHASE_STOP
    if (a.debug_stop == 3) return;
#endif
    if (bad) {
        if (threadIdx.x == 0) { atomicAdd(a.singular, 1); a.n_null[e] = 0; }
        return;
    }
    // (3) extensions and the coarse derivative row (:2394-2409)
    double *oext = a.ext + a.off_ext[e];
    for (int t = threadIdx.x; t < ni * next; t += blockDim.x) { const int c = t / ni, i = t - c * ni; oext[t] = K[i * m + n + c]; }
    for (int c = threadIdx.x; c < next; c += blockDim.x) {
        double v = 0.0;
        if (a.mode == 0) {
            const double l = K[(n - 1) * m + n + c];
            v = fabs(l) > a.smallest_entry ? -l : 0.0;
        }
        a.dc[a.off_dc[e] + c] = v;
    }
    // (4) RangeT bubbles
    double *ort = a.rt + a.off_rt[e];
    for (int t = threadIdx.x; t < ni * nrt; t += blockDim.x) { const int c = t / ni, i = t - c * ni; ort[t] = K[i * m + n + next + c]; }
    // (5) null-space dofs: targets minus their extension, SVD (:2478-2520)
    for (int t = threadIdx.x; t < ni * nTe; t += blockDim.x) { const int c = t / ni, i = t - c * ni; ti[t] = T[c * nu + i] - K[i * m + n + next + nrt + c]; }
    __syncthreads();
    int knull = 0;
    if (nTe > 0) {
        { const double cut = a.svd_tol * 1e-3; jacobi_orthogonalise(ti, ti, ni, nTe, sh, cut * cut); }
        if (threadIdx.x == 0) {
            int k = 0;
            const int kcap = nTe < ni ? nTe : ni;
            while (k < kcap && !(sh.sv[sh.perm[k]] < a.svd_tol)) k++;
            sh.cnt = k;
        }
        __syncthreads();
        knull = sh.cnt;
    }
#ifdef ELAG_AMGE_PHASE_STOP
    if (a.debug_stop == 4) return;
#endif
    double *onul = a.nul + a.off_nul[e];
    for (int t = threadIdx.x; t < ni * a.nT; t += blockDim.x) {
        const int c = t / ni, i = t - c * ni;
        onul[t] = c < knull ? ti[sh.perm[c] * ni + i] / sh.sv[sh.perm[c]] : 0.0;
    }
    if (threadIdx.x == 0) a.n_null[e] = knull;
    // local basis: [extended traces | RangeT | null] on all u dofs of the agglomerate (:2540-2555)
    const int ncol = next + nrt + knull;
    for (int t = threadIdx.x; t < nu * ncol; t += blockDim.x) {
        const int c = t / nu, i = t - c * nu;
        double v = 0.0;
        if (c < next) v = i < ni ? K[i * m + n + c] : R_b[c * nb + (i - ni)];
        else if (c < next + nrt) v = i < ni ? K[i * m + n + c] : 0.0;
        else if (i < ni) { const int src = sh.perm[c - next - nrt]; v = ti[src * ni + i] / sh.sv[src]; }
        basis[t] = v;
    }
    __syncthreads();
    mat_times_cols(Maa, nu, basis, MB, nu, ncol);
    __syncthreads();
    double *ocm = a.cmass + a.off_cm[e];
    for (int t = threadIdx.x; t < ncolmax * ncolmax; t += blockDim.x) ocm[t] = 0.0;
    __syncthreads();
    for (int t = threadIdx.x; t < ncol * ncol; t += blockDim.x) {
        const int r = t / ncol, c = t - r * ncol;
        if (c < r) continue;
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < nu; i++) { s1 += basis[r * nu + i] * MB[c * nu + i]; s2 += basis[c * nu + i] * MB[r * nu + i]; }
        const double v = 0.5 * (s1 + s2);
        ocm[r * ncolmax + c] = v;
        ocm[c * ncolmax + r] = v;
    }
#ifdef ELAG_AMGE_PHASE_STOP
    if (a.debug_stop == 5) return;
#endif
    // cochain projector functional of the interior dofs: [RangeT | null] in the M_ii inner product (:2530-2538)
    double *ofun = a.functional + a.off_fun[e];
    for (int t = threadIdx.x; t < kfmax * ni; t += blockDim.x) ofun[t] = 0.0;
    const int kf = nrt + knull;
    if (kf > 0 && ni > 0) {
        double *PMi = Pi + (size_t)ni * kf;
        for (int t = threadIdx.x; t < ni * kf; t += blockDim.x) { const int c = t / ni, i = t - c * ni; Pi[t] = basis[(next + c) * nu + i]; }
        __syncthreads();
        mat_times_cols(Maa, nu, Pi, PMi, ni, kf);     // leading ni x ni block of M_aa
        __syncthreads();
        bad = dof_functional(Pi, PMi, ni, kf, aug, ofun, nullptr, 0, sh.piv);
        if (bad && threadIdx.x == 0) atomicAdd(a.singular, 1);
    }
}

// ---- ranks of small matrices (Betti numbers of agglomerated entities) ------------------------------------------------------
// AgglomeratedTopologyCheck::computeBettiNumbersAgglomeratedEntities (src/topology/AgglomeratedTopologyCheck.cpp:241-318) takes
// DenseMatrix::Rank(1e-9) of the local boundary operators (entries 0, +-1): here by Gaussian elimination with row pivoting per
// column; a column whose remaining entries are all below the tolerance adds nothing to the rank.
struct RankArgs {
    const int *ent_list;
    int n_mat;
    double tol;
    const int *nr, *nc;
    const int64_t *t_ptr;
    const int *t_row, *t_col;
    const double *t_val;
    int *rank;
    const int64_t *ws_off;
    double *gws;
};

__global__ void batched_rank_kernel(RankArgs a)
{
    extern __shared__ double dyn[];
    __shared__ int piv[2];
    const int e = a.ent_list ? a.ent_list[blockIdx.x] : (int)blockIdx.x;
    const int nr = a.nr[e], nc = a.nc[e];
    double *A = a.ws_off[e] >= 0 ? a.gws + a.ws_off[e] : dyn;
    zero_fill(A, nr * nc);
    __syncthreads();
    for (int64_t t = a.t_ptr[e] + threadIdx.x; t < a.t_ptr[e + 1]; t += blockDim.x) A[a.t_row[t] * nc + a.t_col[t]] = a.t_val[t];
    __syncthreads();
    int r = 0;                                            // rows 0 .. r-1 hold the pivots found so far
    for (int c = 0; c < nc && r < nr; c++) {
        if (threadIdx.x < 64) {
            double best = -1.0;
            int bi = r;
            for (int i = r + threadIdx.x; i < nr; i += 64) {
                const double v = fabs(A[i * nc + c]);
                if (v > best) { best = v; bi = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_down(best, o, 64);
                const int oi = __shfl_down(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (threadIdx.x == 0) { piv[0] = bi; piv[1] = best > a.tol ? 1 : 0; }
        }
        __syncthreads();
        const int p = piv[0], ok = piv[1];
        if (ok) {
            if (p != r)
                for (int j = c + threadIdx.x; j < nc; j += blockDim.x) {
                    const double t = A[r * nc + j];
                    A[r * nc + j] = A[p * nc + j];
                    A[p * nc + j] = t;
                }
            __syncthreads();
            const double inv = 1.0 / A[r * nc + c];
            const int w = nc - c - 1, h = nr - r - 1;
            for (int t = threadIdx.x; t < w * h; t += blockDim.x) {
                const int i = r + 1 + t / w, j = c + 1 + t % w;
                A[i * nc + j] -= (A[i * nc + c] * inv) * A[r * nc + j];
            }
            r++;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.rank[e] = r;
}

// ---- host side ----------------------------------------------------------------------------------------------------------

// Device scratch of one ABI call.  The calls of a coarsening stage need some dozens of buffers each; hipMalloc / hipFree per
// buffer cost more than the kernels (about 10 ms a call against 1 ms of kernel at cube456 x3).  The arena is one grow-only
// allocation per host thread, sized by the previous call's demand: buffers are bump-allocated from it, what does not fit is
// allocated separately for this call, and the next call starts with an arena large enough for everything.  All buffers die
// together when the call returns (every entry point synchronises its stream first).
struct Arena {
    char *base = nullptr;
    size_t cap = 0, off = 0, want = 0;
    std::vector<void *> spill;
    void begin()
    {
        end();
        if (want > cap) {
            if (base) (void)hipFree(base);
            base = nullptr;
            cap = want + want / 4;
            if (std::getenv("ELAG_HIP_ARENA_TRACE")) std::fprintf(stderr, "[arena] grow to %zu MiB\n", cap >> 20);
            base = static_cast<char *>(device_malloc(cap));
        }
        off = 0;
        want = 0;
    }
    void *take(size_t bytes)
    {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        want += bytes;
        if (off + bytes <= cap) {
            void *p = base + off;
            off += bytes;
            return p;
        }
        void *p = nullptr;
        if (std::getenv("ELAG_HIP_ARENA_TRACE")) std::fprintf(stderr, "[arena] spill %zu KiB (cap %zu MiB, used %zu MiB)\n", bytes >> 10, cap >> 20, off >> 20);
        p = device_malloc(bytes);
        spill.push_back(p);
        return p;
    }
    void end()
    {
        for (void *p : spill) (void)hipFree(p);
        spill.clear();
    }
    void release()                                           // elag_hip_finalize: give this thread's arena back
    {
        end();
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = off = want = 0;
    }
    ~Arena() { end(); }                                      // a thread that never finalises leaves the arena to process teardown
};
thread_local Arena g_arena;
}  // namespace
void amge_release_arena() { g_arena.release(); }
namespace {
struct ArenaScope {
    ArenaScope() { g_arena.begin(); }
    ~ArenaScope() { g_arena.end(); }
};
template <class T>
T *arena_alloc(size_t n) { return static_cast<T *>(g_arena.take(n * sizeof(T))); }
template <class T>
T *arena_from_host(const T *h, size_t n)
{
    T *p = arena_alloc<T>(n);
    if (n) stage_h2d(p, h, n * sizeof(T));
    return p;
}

struct AsmUpload {
    int64_t *asm_ptr = nullptr, *blk_off = nullptr, *loc_off = nullptr;
    int *blk_n = nullptr, *loc_idx = nullptr;
    double *blocks = nullptr;
    AsmDev dev{};
    void upload(const elag_hip_dense_assembly &h, int n_ent)
    {
        asm_ptr = arena_from_host(h.asm_ptr, (size_t)n_ent + 1);
        blk_off = arena_from_host(h.blk_off, (size_t)h.n_items);
        loc_off = arena_from_host(h.loc_off, (size_t)h.n_items);
        blk_n = arena_from_host(h.blk_n, (size_t)h.n_items);
        loc_idx = arena_from_host(h.loc_idx, (size_t)h.n_loc);
        blocks = arena_from_host(h.blocks, (size_t)h.n_block_values);
        dev = AsmDev{asm_ptr, blk_off, loc_off, blk_n, loc_idx, blocks};
    }
};

void check_assembly(const elag_hip_dense_assembly &h, int n_ent, const int *order, const char *what)
{
    ELAG_REQUIRE(h.asm_ptr && h.asm_ptr[0] == 0 && h.asm_ptr[n_ent] == h.n_items, "%s: assembly list does not cover its items", what);
    for (int e = 0; e < n_ent; e++)
        for (int64_t t = h.asm_ptr[e]; t < h.asm_ptr[e + 1]; t++) {
            ELAG_REQUIRE(h.blk_n[t] >= 0 && h.blk_off[t] >= 0 && h.blk_off[t] + (int64_t)h.blk_n[t] * h.blk_n[t] <= h.n_block_values,
                         "%s: block %lld outside the packed values", what, (long long)t);
            ELAG_REQUIRE(h.loc_off[t] >= 0 && h.loc_off[t] + h.blk_n[t] <= h.n_loc, "%s: index list %lld outside loc_idx", what, (long long)t);
            for (int i = 0; i < h.blk_n[t]; i++)
                ELAG_REQUIRE(h.loc_idx[h.loc_off[t] + i] >= 0 && h.loc_idx[h.loc_off[t] + i] < order[e],
                             "%s: local index out of range in entity %d", what, e);
        }
}

template <class T>
struct DevBuf {                                              // a buffer of the current call's arena
    T *p = nullptr;
    explicit DevBuf(size_t n) : p(arena_alloc<T>(n)) {}
    DevBuf(const T *h, size_t n) : p(arena_from_host(h, n)) {}
    operator T *() const { return p; }
};

// Scratch placement: entities that fit into the dynamic LDS budget run there, the others get a slice of global scratch.
// Where an entity's workspace lives (LDS or a slot of the global scratch) and how many threads work on it.  Both depend on
// the entity alone, never on what else is in the launch: an entity computed by two ranks (an agglomerated facet on their
// interface, derham_dist.py) or in two differently composed batches goes through the same reductions in the same order and
// comes out bit-identical.  Entities of the two workgroup sizes are launched separately.
struct Placement {
    static constexpr int64_t kWideNeed = 3000;               // more doubles than this: 256 threads instead of one wave
    std::vector<int64_t> ws_off;
    int64_t gws_total = 0;
    std::vector<int> group[2];                               // entities per workgroup size (64, 256)
    int lds_doubles[2] = {0, 0};
    void finish(const std::vector<int64_t> &need)
    {
        ws_off.resize(need.size());
        for (size_t e = 0; e < need.size(); e++) {
            const int g = need[e] > kWideNeed ? 1 : 0;
            group[g].push_back((int)e);
            if (need[e] <= kLdsDoubles) { ws_off[e] = -1; lds_doubles[g] = std::max<int>(lds_doubles[g], (int)need[e]); }
            else { ws_off[e] = gws_total; gws_total += need[e]; }
        }
    }
    // launch(list or nullptr, number of workgroups, threads, dynamic LDS bytes) once per non-empty group
    template <class F> void for_each_group(F &&launch) const
    {
        static const int threads[2] = {64, 256};
        if (group[0].empty() || group[1].empty()) {
            const int g = group[0].empty() ? 1 : 0;
            if (!group[g].empty()) launch((const int *)nullptr, (int)group[g].size(), threads[g], (size_t)lds_doubles[g] * sizeof(double));
            return;
        }
        for (int g = 0; g < 2; g++) {
            lists[g].reset(new DevBuf<int>(group[g].data(), group[g].size()));
            launch((const int *)lists[g]->p, (int)group[g].size(), threads[g], (size_t)lds_doubles[g] * sizeof(double));
        }
    }
    mutable std::unique_ptr<DevBuf<int>> lists[2];
};

float run_timed(hipStream_t st, const std::function<void()> &launch)
{
    hipEvent_t e0, e1;
    ELAG_HIP_CHECK(hipEventCreate(&e0));
    ELAG_HIP_CHECK(hipEventCreate(&e1));
    ELAG_HIP_CHECK(hipEventRecord(e0, st));
    launch();
    ELAG_HIP_CHECK(hipEventRecord(e1, st));
    event_wait(e1);
    float ms = 0.f;
    ELAG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms;
}

std::vector<int64_t> prefix(const std::vector<int64_t> &cnt)
{
    std::vector<int64_t> p(cnt.size() + 1, 0);
    for (size_t i = 0; i < cnt.size(); i++) p[i + 1] = p[i] + cnt[i];
    return p;
}

}  // namespace
}  // namespace elag

using namespace elag;

extern "C" {

ELAG_HIP_API int elag_hip_coarsen_traces(const elag_hip_coarsen_traces_in *in, const elag_hip_coarsen_traces_out *out, double *kernel_ms)
{
    return guarded([&] {
        require_init();
        ArenaScope arena_scope;
        ELAG_REQUIRE(in && out, "null argument");
        const int ne = in->n_ent, nT = in->nT, kmax = 1 + nT;
        ELAG_REQUIRE(ne >= 0 && nT >= 0 && kmax <= kMaxCols, "at most %d target vectors are supported (got %d)", kMaxCols - 1, nT);
        if (kernel_ms) *kernel_ms = 0.0;
        if (ne == 0) return;
        check_assembly(in->mass, ne, in->ndof, "coarsen_traces mass");
        hipStream_t st = ctx().stream;
        std::vector<int64_t> cnt(ne), need(ne);
        for (int e = 0; e < ne; e++) {
            ELAG_REQUIRE(in->ndof[e] > 0, "entity %d has no dofs", e);
            cnt[e] = in->ndof[e];
            const int64_t n = in->ndof[e];
            need[e] = n * n + 2 * n * nT + 2 * n + 2 * n * kmax + (int64_t)kmax * (kmax + n);
        }
        const std::vector<int64_t> doff = prefix(cnt);
        const int64_t nd = doff[ne];
        Placement pl;
        pl.finish(need);
        AsmUpload mass;
        mass.upload(in->mass, ne);
        DevBuf<int> d_ndof(in->ndof, ne), d_nc((size_t)ne), d_sing(1);
        DevBuf<int64_t> d_doff(doff.data(), (size_t)ne + 1), d_wsoff(pl.ws_off.data(), (size_t)ne);
        DevBuf<double> d_pv(in->pv, (size_t)nd), d_T(in->T, (size_t)nd * nT), d_p((size_t)nd * kmax), d_cm((size_t)ne * kmax * kmax),
            d_f((size_t)nd * kmax), d_gws((size_t)pl.gws_total);
        zero_async(d_sing.p, sizeof(int));
        TracesArgs a{};
        a.n_ent = ne; a.nT = nT; a.svd_tol = in->svd_tol; a.ndof = d_ndof; a.dof_off = d_doff; a.mass = mass.dev;
        a.pv = d_pv; a.T = d_T; a.ncoarse = d_nc; a.p_loc = d_p; a.cmass = d_cm; a.functional = d_f;
        a.ws_off = d_wsoff; a.gws = d_gws; a.singular = d_sing;
        const float ms = run_timed(st, [&] {
            pl.for_each_group([&](const int *list, int n, int threads, size_t lds) {
                a.ent_list = list;
                hipLaunchKernelGGL(coarsen_traces_kernel, dim3(n), dim3(threads), lds, st, a);
            });
        });
        ELAG_LAUNCH_CHECK();
        if (kernel_ms) *kernel_ms = ms;
        int bad = 0;
        stage_d2h(&bad, d_sing.p, sizeof(int));
        stage_d2h(out->ncoarse, d_nc.p, sizeof(int) * ne);
        stage_d2h(out->p_loc, d_p.p, sizeof(double) * nd * kmax);
        stage_d2h(out->cmass, d_cm.p, sizeof(double) * (size_t)ne * kmax * kmax);
        stage_d2h(out->functional, d_f.p, sizeof(double) * nd * kmax);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        if (bad) fail(ELAG_HIP_ERR_NUMERIC, "coarsen_traces: %d entities with a singular coarse mass matrix", bad);
    });
}

ELAG_HIP_API int elag_hip_coarsen_extension(const elag_hip_coarsen_extension_in *in, const elag_hip_coarsen_extension_out *out, double *kernel_ms)
{
    return guarded([&] {
        require_init();
        ArenaScope arena_scope;
        ELAG_REQUIRE(in && out, "null argument");
        const int ne = in->n_ent, nT = in->nT;
        ELAG_REQUIRE(ne >= 0 && nT >= 0 && nT < kMaxCols, "at most %d target vectors are supported (got %d)", kMaxCols - 1, nT);
        if (kernel_ms) *kernel_ms = 0.0;
        if (ne == 0) return;
        const int mode = in->mode;
        ELAG_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (facet) or 1 (ridge / peak)");
        check_assembly(in->mass_u, ne, in->nu, "coarsen_extension mass_u");
        check_assembly(in->mass_p, ne, in->np, "coarsen_extension mass_p");
        if (mode) check_assembly(in->mass_q, ne, in->nq, "coarsen_extension mass_q");
        hipStream_t st = ctx().stream;
        std::vector<int64_t> c_p(ne), c_Rb(ne), c_cP(ne), c_T(ne), c_PDc(ne), c_ext(ne), c_dc(ne), c_rt(ne), c_nul(ne), c_fun(ne), c_cm(ne), need(ne);
        for (int e = 0; e < ne; e++) {
            const int64_t nu = in->nu[e], ni = in->ni[e], np = in->np[e], nx = in->next[e], nrt = in->nrt[e];
            const int64_t npi = mode ? in->npi[e] : np, nq = mode ? in->nq[e] : 0;
            ELAG_REQUIRE(nu > 0 && ni >= 0 && ni <= nu && np >= 0 && npi >= 0 && npi <= np && nq >= 0 && nx >= 0 && nrt >= 0 && (mode || np > 0),
                         "bad dimensions of entity %d", e);
            if (mode)
                for (int64_t t = in->d2_ptr[e]; t < in->d2_ptr[e + 1]; t++)
                    ELAG_REQUIRE(in->d2_row[t] >= 0 && in->d2_row[t] < nq && in->d2_col[t] >= 0 && in->d2_col[t] < np, "second derivative triplet outside entity %d", e);
            for (int64_t t = in->d_ptr[e]; t < in->d_ptr[e + 1]; t++)
                ELAG_REQUIRE(in->d_row[t] >= 0 && in->d_row[t] < np && in->d_col[t] >= 0 && in->d_col[t] < nu, "derivative triplet outside entity %d", e);
            c_p[e] = np; c_Rb[e] = (nu - ni) * nx; c_cP[e] = npi * nrt; c_T[e] = nu * nT; c_PDc[e] = mode ? np * nx : 0;
            c_ext[e] = ni * nx; c_dc[e] = nx; c_rt[e] = ni * nrt; c_nul[e] = ni * nT; c_fun[e] = (nrt + nT) * ni;
            const int64_t ncm = nx + nrt + nT;
            c_cm[e] = ncm * ncm;
            const int64_t n = ni + npi + (mode ? 0 : 1), m = n + nx + nrt + nT, kf = nrt + nT;
            need[e] = nu * nu + np * np + np * nu + npi * nu + np + n * m + ni * nT + 2 * nu * ncm + kf * (kf + ni) + 2 * ni * kf + nq * nq + 2 * nq * np;
        }
        const auto o_p = prefix(c_p), o_Rb = prefix(c_Rb), o_cP = prefix(c_cP), o_T = prefix(c_T), o_PDc = prefix(c_PDc), o_ext = prefix(c_ext), o_dc = prefix(c_dc),
                   o_rt = prefix(c_rt), o_nul = prefix(c_nul), o_fun = prefix(c_fun), o_cm = prefix(c_cm);
        Placement pl;
        pl.finish(need);
        AsmUpload mu, mp, mq;
        mu.upload(in->mass_u, ne);
        mp.upload(in->mass_p, ne);
        if (mode) mq.upload(in->mass_q, ne);
        const int64_t nd = in->d_ptr[ne], nd2 = mode ? in->d2_ptr[ne] : 0;
        std::vector<int> np_copy(in->np, in->np + ne);
        DevBuf<int> d_npi(mode ? in->npi : np_copy.data(), ne), d_nq(mode ? in->nq : np_copy.data(), ne), d_d2row(mode ? in->d2_row : nullptr, (size_t)nd2),
            d_d2col(mode ? in->d2_col : nullptr, (size_t)nd2);
        std::vector<int64_t> zero_ptr((size_t)ne + 1, 0);
        DevBuf<int64_t> d_d2ptr(mode ? in->d2_ptr : zero_ptr.data(), (size_t)ne + 1), f_PDc(o_PDc.data(), ne);
        DevBuf<double> d_d2val(mode ? in->d2_val : nullptr, (size_t)nd2), d_PDc(mode ? in->PDc : nullptr, (size_t)o_PDc[ne]);
        DevBuf<int> d_nu(in->nu, ne), d_ni(in->ni, ne), d_np(in->np, ne), d_nx(in->next, ne), d_nrt(in->nrt, ne), d_drow(in->d_row, (size_t)nd),
            d_dcol(in->d_col, (size_t)nd), d_nnull((size_t)ne), d_sing(1);
        DevBuf<int64_t> d_dptr(in->d_ptr, (size_t)ne + 1), d_wsoff(pl.ws_off.data(), (size_t)ne);
        DevBuf<int64_t> f_p(o_p.data(), ne), f_Rb(o_Rb.data(), ne), f_cP(o_cP.data(), ne), f_T(o_T.data(), ne), f_ext(o_ext.data(), ne),
            f_dc(o_dc.data(), ne), f_rt(o_rt.data(), ne), f_nul(o_nul.data(), ne), f_fun(o_fun.data(), ne), f_cm(o_cm.data(), ne);
        DevBuf<double> d_dval(in->d_val, (size_t)nd), d_pvp(in->pv_p, (size_t)o_p[ne]), d_Rb(in->R_b, (size_t)o_Rb[ne]), d_cP(in->cP, (size_t)o_cP[ne]),
            d_T(in->T, (size_t)o_T[ne]);
        DevBuf<double> d_ext((size_t)o_ext[ne]), d_dc((size_t)o_dc[ne]), d_rt((size_t)o_rt[ne]), d_nul((size_t)o_nul[ne]), d_fun((size_t)o_fun[ne]),
            d_cm((size_t)o_cm[ne]), d_gws((size_t)pl.gws_total);
        zero_async(d_sing.p, sizeof(int));
        ExtArgs a{};
        a.n_ent = ne; a.nT = nT; a.mode = mode; a.null_dofs = in->null_dofs; a.debug_stop = getenv("ELAG_HIP_AMGE_STOP") ? atoi(getenv("ELAG_HIP_AMGE_STOP")) : 0; a.svd_tol = in->svd_tol; a.smallest_entry = in->smallest_entry;
        a.nu = d_nu; a.ni = d_ni; a.np = d_np; a.npi = d_npi; a.nq = d_nq; a.next = d_nx; a.nrt = d_nrt; a.mass_u = mu.dev; a.mass_p = mp.dev; a.mass_q = mq.dev;
        a.d2_ptr = d_d2ptr; a.d2_row = d_d2row; a.d2_col = d_d2col; a.d2_val = d_d2val; a.PDc = d_PDc; a.off_PDc = f_PDc;
        a.d_ptr = d_dptr; a.d_row = d_drow; a.d_col = d_dcol; a.d_val = d_dval;
        a.pv_p = d_pvp; a.R_b = d_Rb; a.cP = d_cP; a.T = d_T;
        a.off_p = f_p; a.off_Rb = f_Rb; a.off_cP = f_cP; a.off_T = f_T; a.off_ext = f_ext; a.off_dc = f_dc; a.off_rt = f_rt; a.off_nul = f_nul;
        a.off_fun = f_fun; a.off_cm = f_cm;
        a.n_null = d_nnull; a.ext = d_ext; a.dc = d_dc; a.rt = d_rt; a.nul = d_nul; a.functional = d_fun; a.cmass = d_cm;
        a.ws_off = d_wsoff; a.gws = d_gws; a.singular = d_sing;
        const float ms = run_timed(st, [&] {
            pl.for_each_group([&](const int *list, int n, int threads, size_t lds) {
                a.ent_list = list;
                hipLaunchKernelGGL(coarsen_extension_kernel, dim3(n), dim3(threads), lds, st, a);
            });
        });
        ELAG_LAUNCH_CHECK();
        if (kernel_ms) *kernel_ms = ms;
        int bad = 0;
        stage_d2h(&bad, d_sing.p, sizeof(int));
        stage_d2h(out->n_null, d_nnull.p, sizeof(int) * ne);
        auto down = [&](double *h, const double *d, int64_t n) { if (n) stage_d2h(h, d, sizeof(double) * n); };
        down(out->ext, d_ext, o_ext[ne]); down(out->dc, d_dc, o_dc[ne]); down(out->rt, d_rt, o_rt[ne]); down(out->nul, d_nul, o_nul[ne]);
        down(out->functional, d_fun, o_fun[ne]); down(out->cmass, d_cm, o_cm[ne]);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        if (bad)
            fail(ELAG_HIP_ERR_NUMERIC, "coarsen_extension: %d singular local systems (bad topology, e.g. a torus-shaped or disconnected agglomerate?)", bad);
    });
}

ELAG_HIP_API int elag_hip_batched_rank(int n_mat, const int *nrows, const int *ncols, const int64_t *t_ptr, const int *t_row, const int *t_col,
                                       const double *t_val, double tol, int *rank)
{
    return guarded([&] {
        require_init();
        ArenaScope arena_scope;
        ELAG_REQUIRE(n_mat >= 0 && (n_mat == 0 || (nrows && ncols && t_ptr && rank)), "null argument");
        if (n_mat == 0) return;
        hipStream_t st = ctx().stream;
        std::vector<int64_t> need((size_t)n_mat);
        for (int e = 0; e < n_mat; e++) {
            ELAG_REQUIRE(nrows[e] >= 0 && ncols[e] >= 0, "negative matrix size");
            for (int64_t t = t_ptr[e]; t < t_ptr[e + 1]; t++)
                ELAG_REQUIRE(t_row[t] >= 0 && t_row[t] < nrows[e] && t_col[t] >= 0 && t_col[t] < ncols[e], "triplet outside matrix %d", e);
            need[(size_t)e] = (int64_t)nrows[e] * ncols[e];
        }
        Placement pl;
        pl.finish(need);
        const int64_t nt = t_ptr[n_mat];
        DevBuf<int> d_nr(nrows, n_mat), d_nc(ncols, n_mat), d_row(t_row, (size_t)nt), d_col(t_col, (size_t)nt), d_rank((size_t)n_mat);
        DevBuf<int64_t> d_ptr(t_ptr, (size_t)n_mat + 1), d_ws(pl.ws_off.data(), (size_t)n_mat);
        DevBuf<double> d_val(t_val, (size_t)nt), d_gws((size_t)pl.gws_total);
        RankArgs a{nullptr, n_mat, tol, d_nr, d_nc, d_ptr, d_row, d_col, d_val, d_rank, d_ws, d_gws};
        pl.for_each_group([&](const int *list, int n, int threads, size_t lds) {
            a.ent_list = list;
            hipLaunchKernelGGL(batched_rank_kernel, dim3(n), dim3(threads), lds, st, a);
        });
        ELAG_LAUNCH_CHECK();
        stage_d2h(rank, d_rank.p, sizeof(int) * n_mat);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
    });
}

}  // extern "C"
