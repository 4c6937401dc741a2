// Krylov shells around the V-cycle, resident on the device.
//
// Reference: parelag::KrylovSolver (src/linalg/solver_ops/ParELAG_KrylovSolver.cpp:23-96) is a thin wrapper that picks
// mfem::CGSolver / GMRESSolver / FGMRESSolver by "Solver name", sets iterative_mode = false on solver and
// preconditioner and forwards "Relative tolerance" / "Absolute tolerance" / "Maximum iterations" / "Restart size";
// parelag::StationarySolver::Mult (solver_ops/ParELAG_StationarySolver.cpp:41-130) is the preconditioned Richardson loop.
// The iterations themselves live in MFEM (linalg/solvers.cpp, third party, not in the reference tree); they are restated
// here from the published MFEM 4.x algorithms, statement by statement (same order of operations, same tests, same
// final_iter / final_norm / converged bookkeeping).
//
// GPU design: nothing in the loop goes back to the host.  Every scalar of the recurrences (nom, den, alpha, beta, the
// Hessenberg matrix, the Givens rotations) lives in a small device state block; one-thread "scalar" kernels advance the
// recurrences and take the reference's branches, vector kernels read their coefficients from that block and are
// predicated on its `done` flag.  The host enqueues iterations blindly (CG: one captured hipGraph per iteration) and
// only polls the flag every `check_every` iterations; once `done` is set the remaining enqueued kernels are no-ops, so
// the iterate and the iteration count are those of the reference's sequential loop -- operation by operation for CG, BiCGSTAB, MINRES,
// the stationary iteration and GMRES / FGMRES with MFEM's modified Gram-Schmidt loop, one basis vector at a time.  On one rank a GMRES step's
// whole chain, its norm and the new basis vector are ONE launch (gmres_chain_kernel: w stays in registers, the inner products' partial sums
// go from workgroup to workgroup as tagged granules); ELAG_HIP_GMRES_CHAIN=0, a distributed operator or more than 1.5 M unknowns: a launch per
// link, the same bits.  (A two-vector chain that evaluated MGS's coefficients in another order -- rounds 4 and 5, opt-in in round 5 -- is gone:
// slower than the chain in one launch and not MFEM's rounding.)  The step's Givens rotations ride in that launch too; the ONE update x += V y
// of a solve that converges inside a cycle is launched by the host when it has read `done` (gmres_final_update_kernel).  CG on a
// launch-bound system: direction update, product and (d, A d) in one launch (cg_dir_amul_dot_kernel), the update of x and r in the launch
// that computes alpha: six launches a pass with a three-launch cycle.
#include <cmath>

#include <chrono>
#include <cstring>
#include "comm.hpp"
#include "smoother.hpp"
#include "staging.hpp"
#include "row_group.hpp"

using namespace elag;

namespace {

constexpr int kB = 256;
inline unsigned g1(long long n) { return (unsigned)std::max<long long>(1, (n + kB - 1) / kB); }

enum { S_NOM, S_DEN, S_BETANOM, S_R0, S_ALPHA, S_BETA, S_TMP, S_FINAL_NORM, S_GBETA, S_THRESH, S_NORMK1, S_RATIO,
       // BiCGSTAB / MINRES recurrences
       S_RHO1, S_RHO2, S_OMEGA, S_RESID, S_GOAL, S_TS, S_ETA, S_GAMMA0, S_GAMMA1, S_SIGMA0, S_SIGMA1, S_DELTA, S_MRHO1, S_MRHO2, S_MRHO3,
       S_C0, S_C1, S_C2, S_COUNT = 40 };
enum { F_DONE, F_CONVERGED, F_FINAL_ITER, F_ITER, F_PENDING, F_ITER2, F_STEP, F_JBASE, F_FINAL_COUNT, F_COUNT = 10 };   // F_STEP / F_JBASE: GMRES -- the step counter and the iteration number of a cycle's first step, kept on the device so that a step's launches take no per-step argument (they replay as a graph)

struct KState {
    int fl[F_COUNT];          // (first: the flags and the scalars the host reads -- S_NOM ... S_FINAL_NORM -- are 26 consecutive words, one fetch_ints)
    double sc[S_COUNT];
    int chain[4];             // gmres_chain_kernel: [0] launches completed (the tag of the next launch's partial sums is this + 1), [1] workgroups out, [2] somebody gave up
};
constexpr int kPollWords = F_COUNT + 2 * (S_FINAL_NORM + 1);
static_assert(F_COUNT % 2 == 0 && kPollWords <= 30, "the polled words: the flags, then the leading scalars");

// ---- CG scalar kernels (mfem::CGSolver::Mult) --------------------------------------------------------------------------
__global__ void cg_init1_kernel(KState *st, double rel_tol, double abs_tol, int max_iter)
{
    const double nom = st->sc[S_TMP];
    st->sc[S_NOM] = nom;
    st->sc[S_BETANOM] = nom;
    st->fl[F_DONE] = 0; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = max_iter; st->fl[F_ITER] = 1;
    if (nom < 0.0) {   // "PCG: The preconditioner is not positive definite."
        st->fl[F_DONE] = 1; st->fl[F_FINAL_ITER] = 0; st->sc[S_FINAL_NORM] = nom;
        return;
    }
    const double r0 = fmax(nom * rel_tol * rel_tol, abs_tol * abs_tol);
    st->sc[S_R0] = r0;
    if (nom <= r0) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = 0; st->sc[S_FINAL_NORM] = sqrt(nom); }
}

__global__ void cg_init2_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    const double den = st->sc[S_TMP];
    st->sc[S_DEN] = den;
    if (den == 0.0) { st->fl[F_DONE] = 1; st->fl[F_FINAL_ITER] = 0; st->sc[S_FINAL_NORM] = sqrt(st->sc[S_NOM]); return; }
    st->sc[S_ALPHA] = st->sc[S_NOM] / den;
}

__global__ void cg_step1_kernel(KState *st, int max_iter)
{
    if (st->fl[F_DONE]) return;
    const double betanom = st->sc[S_TMP];
    st->sc[S_BETANOM] = betanom;
    st->sc[S_FINAL_NORM] = sqrt(betanom);
    int i = st->fl[F_ITER];
    if (betanom < 0.0) { st->fl[F_DONE] = 1; st->fl[F_FINAL_ITER] = i; return; }
    if (betanom <= st->sc[S_R0]) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = i; return; }
    if (++i > max_iter) { st->fl[F_DONE] = 1; st->fl[F_ITER] = i; return; }   // final_iter stays max_iter
    st->fl[F_ITER] = i;
    st->sc[S_BETA] = betanom / st->sc[S_NOM];
}

__global__ void cg_step2_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    const double den = st->sc[S_TMP];
    st->sc[S_DEN] = den;
    if (den == 0.0) { st->fl[F_DONE] = 1; st->fl[F_FINAL_ITER] = st->fl[F_ITER]; return; }
    st->sc[S_NOM] = st->sc[S_BETANOM];
    st->sc[S_ALPHA] = st->sc[S_NOM] / den;
}

// x += alpha d ; r -= alpha z
__global__ void cg_update_kernel(int n, const KState *__restrict__ st, double *__restrict__ x, const double *__restrict__ d,
                                 double *__restrict__ r, const double *__restrict__ z)
{
    if (st->fl[F_DONE]) return;
    const double alpha = st->sc[S_ALPHA];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { x[i] = x[i] + alpha * d[i]; r[i] = r[i] + (-alpha) * z[i]; }
}

// d = src + beta d
__global__ void cg_dir_kernel(int n, const KState *__restrict__ st, const double *__restrict__ src, double *__restrict__ d)
{
    if (st->fl[F_DONE]) return;
    const double beta = st->sc[S_BETA];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = src[i] + beta * d[i];
}

// ---- GMRES / FGMRES (mfem::GMRESSolver::Mult, FGMRESSolver::Mult) -------------------------------------------------------
// hs layout: H column-major (m+1) x m, then s, cs, sn, y (m+1 each)
__device__ inline double &Hm(double *hs, int m, int k, int i) { return hs[(size_t)i * (m + 1) + k]; }

__global__ void gmres_init_kernel(KState *st, double rel_tol, double abs_tol, int max_iter)
{
    const double beta = sqrt(st->sc[S_TMP]);
    st->sc[S_GBETA] = beta;
    st->fl[F_DONE] = 0; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = max_iter; st->fl[F_PENDING] = -1; st->fl[F_STEP] = 0; st->fl[F_JBASE] = 1;
    st->fl[F_FINAL_COUNT] = 0;
    const double thr = fmax(rel_tol * beta, abs_tol);
    st->sc[S_THRESH] = thr;
    st->sc[S_FINAL_NORM] = beta;
    st->sc[S_BETANOM] = beta; st->sc[S_R0] = thr;      // (among the words the host polls: the residual estimate of the last step and what it has to reach -- the host sizes its batches from them)
    if (beta <= thr) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = 0; }
}

__global__ void gmres_start_cycle_kernel(KState *st, double *hs, int m, int jbase)
{
    if (st->fl[F_DONE]) return;
    st->fl[F_JBASE] = jbase;
    double *s = hs + (size_t)m * (m + 1);
    for (int k = 0; k <= m; k++) s[k] = 0.0;
    s[0] = st->sc[S_GBETA];
}

// dst = src / sc[idx]
__global__ void scale_inv_kernel(int n, const KState *__restrict__ st, int idx, const double *__restrict__ src, double *__restrict__ dst)
{
    if (st->fl[F_DONE]) return;
    const double f = 1.0 / st->sc[idx];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f * src[i];
}

// H(k,i) = tmp ; w -= tmp v_k
__global__ void gmres_orth_kernel(int n, const KState *__restrict__ st, double *__restrict__ hs, int m, int k, int i, double *__restrict__ w,
                                  const double *__restrict__ v)
{
    if (st->fl[F_DONE]) return;
    const double h = st->sc[S_TMP];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) Hm(hs, m, k, i) = h;
    if (t < n) w[t] = w[t] + (-h) * v[t];
}

// ---- fused modified Gram-Schmidt (single rank) ---------------------------------------------------------------------------
// MGS is a chain h_k = <w, v_k>; w -= h_k v_k.  One kernel per k does both halves of two neighbouring links: it finishes
// the dot whose per-block partial sums the previous kernel left in part_in (every block adds them up in the same fixed
// order), applies w -= h_k v_k, and while the updated w is in registers accumulates the partial sums of the NEXT dot
// (<w, v_{k+1}>, or <w, w> after the last vector).  i+2 launches and 4 vector passes per link instead of 3(i+1) and 5.
constexpr int kPart = 512;   // partial sums per dot (= blocks of the fused kernels)

// The sum of a wavefront's 64 values, the same on every lane: an inclusive scan along each row of 16 lanes (row_shr 1, 2, 4, 8), the rows'
// totals carried over by row_bcast 15 and 31 -- six DPP steps of ~20 cycles where six ds_bpermute exchanges took ~0.5 us (measured in
// gmres_chain_kernel, whose every link waits for two such sums).  A fixed tree: ((a15 + a14) + (a13 + a12)) + ... per row, (row3 + row2) +
// (row1 + row0) at lane 63.  Every inner product of this file is summed through it, so the kernels that must agree bit for bit do.
template <int CTRL, int ROWS>
__device__ inline double dpp_take(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v)
{
    v += dpp_take<0x111, 0xf>(v);     // row_shr:1
    v += dpp_take<0x112, 0xf>(v);     // row_shr:2
    v += dpp_take<0x114, 0xf>(v);     // row_shr:4
    v += dpp_take<0x118, 0xf>(v);     // row_shr:8   (lane 15 of every row: the row's sum)
    v += dpp_take<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_take<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3   (lane 63: the wavefront's sum)
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

__device__ inline double block_sum_256(double v, double *sh)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;   // same value on every thread
}

// the kPart partial sums of an inner product, added up: thread q takes the pair (2q, 2q + 1) -- the two virtual blocks of one workgroup of
// gmres_chain_kernel, which hands their sum over as one value --, then the block's tree
__device__ inline double pair_of(const double *__restrict__ part) { return part[2 * threadIdx.x] + part[2 * threadIdx.x + 1]; }
static_assert(kPart == 2 * kB, "pair_of(): one pair of partial sums per thread");
__device__ inline double sum_partials(const double *__restrict__ part, double *sh) { return block_sum_256(pair_of(part), sh); }

// part_out[b] = partial sums of <w, v> (no update): first link of a step
__global__ void __launch_bounds__(kB) gmres_dot0_kernel(int n, const KState *__restrict__ st, const double *__restrict__ w,
                                                        const double *__restrict__ v, double *__restrict__ part_out)
{
    __shared__ double sh[4];
    const int done = st->fl[F_DONE];                                  // (the flag and the thread's first element travel together)
    const int t0 = blockIdx.x * kB + threadIdx.x;
    double w0 = 0.0, v0 = 0.0;
    if (t0 < n) { w0 = w[t0]; v0 = v[t0]; }
    if (done) return;
    double acc = 0.0;
    if (t0 < n) acc += w0 * v0;
    for (int t = t0 + kPart * kB; t < n; t += kPart * kB) acc += w[t] * v[t];
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) part_out[blockIdx.x] = acc;
}

// h = sum(part_in); H(k,i) = h; w -= h v_k; part_out = partial sums of <w, vnext> (vnext == nullptr: <w, w>)
__global__ void __launch_bounds__(kB) gmres_orth_fused_kernel(int n, const KState *__restrict__ st, double *__restrict__ hs, int m, int k, int i,
                                                              double *__restrict__ w, const double *__restrict__ vk,
                                                              const double *__restrict__ vnext, const double *__restrict__ part_in,
                                                              double *__restrict__ part_out)
{
    __shared__ double sh[4];
    // A Gram-Schmidt step is a chain of these kernels, each a chain of dependent round trips (flag, partial sums, operands): the operands of
    // the thread's first kPre elements -- all of them up to ~790 k unknowns -- and the partial sums are requested together with the flag, before
    // anything is waited for (same arithmetic, same order).
    constexpr int kPre = 6;
    const int t0 = blockIdx.x * kB + threadIdx.x, stride = kPart * kB;
    double pw[kPre], pv[kPre], pn[kPre];
#pragma unroll
    for (int u = 0; u < kPre; u++) {
        const int t = t0 + u * stride;
        pw[u] = 0.0; pv[u] = 0.0; pn[u] = 0.0;
        if (t < n) { pw[u] = w[t]; pv[u] = vk[t]; if (vnext) pn[u] = vnext[t]; }
    }
    const double pp = pair_of(part_in);
    if (st->fl[F_DONE]) return;
    const double h = block_sum_256(pp, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) Hm(hs, m, k, i) = h;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < kPre; u++) {
        const int t = t0 + u * stride;
        if (t < n) {
            const double wn = pw[u] + (-h) * pv[u];
            w[t] = wn;
            acc += wn * (vnext ? pn[u] : wn);
        }
    }
    for (int t = t0 + kPre * stride; t < n; t += stride) {
        const double wn = w[t] + (-h) * vk[t];
        w[t] = wn;
        acc += wn * (vnext ? vnext[t] : wn);
    }
    acc = block_sum_256(acc, sh);
    if (threadIdx.x == 0) part_out[blockIdx.x] = acc;
}

// H(i+1,i) = sqrt(sum(part_in))
__global__ void __launch_bounds__(kB) gmres_norm_fused_kernel(KState *st, double *hs, int m, int i, const double *__restrict__ part_in)
{
    __shared__ double sh[4];
    if (st->fl[F_DONE]) return;
    const double nw = sqrt(sum_partials(part_in, sh));
    if (threadIdx.x == 0) { Hm(hs, m, i + 1, i) = nw; st->sc[S_BETA] = nw; }
}

__device__ inline void gen_rot(double dx, double dy, double &cs, double &sn)
{
    if (dy == 0.0) { cs = 1.0; sn = 0.0; }
    else if (fabs(dy) > fabs(dx)) { const double t = dx / dy; sn = 1.0 / sqrt(1.0 + t * t); cs = t * sn; }
    else { const double t = dy / dx; cs = 1.0 / sqrt(1.0 + t * t); sn = t * cs; }
}
__device__ inline void app_rot(double &dx, double &dy, double cs, double sn)
{
    const double t = cs * dx + sn * dy;
    dy = -sn * dx + cs * dy;
    dx = t;
}
__device__ inline void backsolve(double *hs, int m, int k)   // Update(): y = s; back substitution with H(0..k, 0..k)
{
    double *s = hs + (size_t)m * (m + 1), *y = s + 3 * (m + 1);
    for (int q = 0; q <= m; q++) y[q] = s[q];
    for (int i = k; i >= 0; i--) {
        y[i] /= Hm(hs, m, i, i);
        for (int j = i - 1; j >= 0; j--) y[j] -= Hm(hs, m, j, i) * y[i];
    }
}

// ---- the whole chain of a step in ONE launch ------------------------------------------------------------------------------------------
// The chain above is i + 3 launches per step and each link re-reads w and two basis vectors (6 us a link at 700 k unknowns, 528 links in a
// 32-iteration GMRES(50): a third of the Darcy solve).  Here kChainG workgroups keep their elements of w in registers for the whole step (a
// workgroup = two virtual blocks of kB threads, the blocks of the kernels above: same elements, same order of summation); a link costs one
// read of v_k and one hand-off: the workgroup publishes the sum of its two blocks as two tagged 8-byte granules {launch number, half of the
// double} with agent-scope stores, a wavefront of each workgroup that holds no vector data polls all kChainG values with agent-scope loads
// (cdna_hip_programming.md, Guideline 16, form R2: the data is the flag; MI355X_MICROARCH.md prices the hand-off at 0.8-1.1 us against
// 1.5-1.9 us for a kernel boundary plus the next kernel's first round trips), adds them up in the order of sum_partials() and hands the
// coefficient to the others through LDS.  While it waits, the data wavefronts' loads of the next basis vector are in flight (loads return in
// order per wavefront: the polls must not queue behind them, hence the wavefront of its own).  The norm of the step and v_{i+1} = w / |w|
// come out of the same launch.  Bit for bit the per-link chain: the arithmetic is MFEM's modified Gram-Schmidt loop, one vector at a time.
//   Every workgroup must be resident (one per CU: they are, unless another process fills the device): every wait is bounded
// (ELAG_HIP_GMRES_CHAIN_MS, 250 ms); a launch that gives up stores code 2 in the context's error word, the next ABI call fails and the solvers
// of the process go back to a launch per link.
typedef unsigned long long u64k;
typedef unsigned int u32x4k __attribute__((ext_vector_type(4)));
constexpr int kChainG = kPart / 2;          // workgroups of gmres_chain_kernel = values handed over per inner product
struct ChainArgs {
    int n, m, i;
    KState *st;
    double *hs, *w;
    const double *V;          // the basis, column q at V + q n
    double *vout;             // v_{i+1}
    u64k *gran;               // [m + 2][kChainG][2]
    int *host_err;
    long long ticks;
    int stall;                // test hook: workgroup stall - 1 publishes its first partial sums ten time limits late (ELAG_HIP_GMRES_CHAIN_STALL)
    int givens;               // the step's rotations (gmres_givens_kernel's work) by workgroup 0's polling wavefront, as the coefficients arrive
};

// PF: the next basis vector is requested a link ahead (a third register array); without it the load follows the coefficient.  Two forms are
// built: 6 elements per thread with the look-ahead (up to 786 432 unknowns), 12 without (up to 1 572 864); 24 elements in half-size
// workgroups, two per CU, spilled 230 registers and are not built -- larger systems keep the launch per link.
template <int PRE, bool PF>
__global__ void __launch_bounds__(2 * kB + 64) gmres_chain_kernel(const ChainArgs A)
{
    constexpr int kData = 2 * kB, kWaves = kData / 64;
    __shared__ double sh_a[2][kWaves];
    __shared__ double sh_h[2];
    __shared__ int sh_dead;
    if (A.st->fl[F_DONE]) return;
    const bool poller = threadIdx.x >= kData;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned tag = (unsigned)A.st->chain[0] + 1u;
    if (threadIdx.x == 0) sh_dead = 0;
    const int vb = blockIdx.x * 2 + (threadIdx.x >> 8);               // the virtual block of a data thread
    const int t0 = vb * kB + (threadIdx.x & (kB - 1)), stride = kPart * kB;
    const int n = A.n, i = A.i;
    double pw[PRE], pv[PRE], pn[PF ? PRE : 1];
    auto load_vec = [&](const double *__restrict__ v, double *dst) {
#pragma unroll
        for (int u = 0; u < PRE; u++) { const int t = t0 + u * stride; dst[u] = t < n ? v[t] : 0.0; }
    };
    if (!poller) { load_vec(A.w, pw); load_vec(A.V, pv); }
    // The step's rotations ride along (A.givens): workgroup 0's polling wavefront is handed every coefficient of column i anyway, so its first
    // lane applies rotation k - 1 when H(k, i) arrives -- while the data wavefronts are busy with the link -- and generates the step's own
    // rotation from the norm: gmres_givens_kernel's operations in its order, one launch of a step less.  The rotations so far wait in LDS.
    extern __shared__ double sh_rot[];                                // c at [0, m), s at [m, 2 m)
    __shared__ double sh_g[4];                                        // the entry of column i being rotated, s_i, s_{i+1}, the threshold (LDS, not registers: the 12-element form has none to spare)
    __shared__ int sh_jbase;
    const bool giv = A.givens != 0 && poller && blockIdx.x == 0;
    if (giv) {
        const double *rs = A.hs + (size_t)A.m * (A.m + 1), *rc = rs + (A.m + 1), *rn = rc + (A.m + 1);
        for (int k = lane; k < i; k += 64) { sh_rot[k] = rc[k]; sh_rot[A.m + k] = rn[k]; }
        if (lane == 0) { sh_g[1] = rs[i]; sh_g[2] = rs[i + 1]; sh_g[3] = A.st->sc[S_THRESH]; sh_jbase = A.st->fl[F_JBASE]; }
    }
    __syncthreads();
    // data wavefronts: each one's sum to LDS, then the barrier; the workgroup's first thread adds them up as block_sum_256() and pair_of() do and
    // publishes the one value (a data wavefront's store: loads and stores of a wavefront share a counter, the polling wavefront's polls would
    // wait for the write-through of its own store)
    auto publish = [&](double acc, int slot) {
        if (!poller) {
            acc = wave_sum(acc);
            if (lane == 0) sh_a[slot & 1][wv] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (slot == 0 && A.stall == (int)blockIdx.x + 1) {
                const long long t_in = (long long)wall_clock64();
                while ((long long)wall_clock64() - t_in < 10 * A.ticks) __builtin_amdgcn_s_sleep(8);
            }
            const double *q = sh_a[slot & 1];
            const double r = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
            const u64k bits = (u64k)__double_as_longlong(r);
            u64k *g = A.gran + ((size_t)slot * kChainG + blockIdx.x) * 2;
            __hip_atomic_store(g, ((u64k)tag << 32) | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g + 1, ((u64k)tag << 32) | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // the polling wavefront: the kChainG sums of `slot`, four per lane, each granule pair one 16-byte load; all four requested before any is
    // looked at (tested one by one as they come, the compiler puts a wait between the requests: a round trip each); added up as
    // sum_partials() does: lane l's j-th value is thread 64 j + l's pair.  The sum to everybody through LDS.  (Nothing is assumed about the
    // 16 bytes as a whole: each aligned 8-byte half is one granule with its own tag, and a pair counts only when both halves carry this
    // launch's tag -- a load that catches one new and one old half is a round that is repeated.)
    auto collect = [&](int slot, bool root) -> double {
        if (poller) {
            const u64k *g = A.gran + ((size_t)slot * kChainG + lane) * 2;
            double p[4];
            unsigned spins = 0;
            long long tw = 0;
            bool dead = false;
            while (true) {
                u32x4k w[4];
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                             "global_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                             "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
                             "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\t"
                             "s_waitcnt vmcnt(0)"
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(g) : "memory");
                bool have = true;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    have = have && w[j].y == tag && w[j].w == tag;
                    p[j] = __hiloint2double((int)w[j].z, (int)w[j].x);
                }
                if (__all(have)) break;
                if ((++spins & 63u) != 0) continue;
                const long long now = (long long)wall_clock64();
                if (spins == 64u) tw = now;
                const bool late = now - tw > A.ticks;
                const bool stop = __hip_atomic_load(A.st->chain + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (__any(late || stop)) {
                    if (__any(late) && lane == 0) {
                        __hip_atomic_store(A.st->chain + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(A.host_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    dead = true;
                    break;
                }
            }
            double r = 0.0;
            if (!dead) {
                const double s0 = wave_sum(p[0]), s1 = wave_sum(p[1]), s2 = wave_sum(p[2]), s3 = wave_sum(p[3]);
                r = (s0 + s1) + (s2 + s3);
                if (root) r = sqrt(r);
            }
            if (lane == 0) { sh_h[slot & 1] = r; if (dead) sh_dead = 1; }
        }
        __syncthreads();
        return sh_h[slot & 1];
    };
    double acc = 0.0;
    if (!poller) {
#pragma unroll
        for (int u = 0; u < PRE; u++) if (t0 + u * stride < n) acc += pw[u] * pv[u];
    }
    publish(acc, 0);
    if (PF && !poller && i >= 1) load_vec(A.V + (size_t)n, pn);
    bool dead = false;
    for (int k = 0; k <= i; k++) {
        const double h = collect(k, false);
        if (sh_dead) { dead = true; break; }
        if (poller && blockIdx.x == 0 && lane == 0) {
            if (!giv) Hm(A.hs, A.m, k, i) = h;
            else if (k == 0) sh_g[0] = h;
            else {                                                    // app_rot(H(k-1,i), H(k,i), cs[k-1], sn[k-1])
                double cur = sh_g[0], nxt = h;
                app_rot(cur, nxt, sh_rot[k - 1], sh_rot[A.m + k - 1]);
                Hm(A.hs, A.m, k - 1, i) = cur;
                sh_g[0] = nxt;
            }
        }
        acc = 0.0;
        if (!poller) {
            const bool last = k == i;
            if (PF) {
#pragma unroll
                for (int u = 0; u < PRE; u++) {
                    if (t0 + u * stride < n) {
                        const double wn = pw[u] + (-h) * pv[u];
                        pw[u] = wn;
                        acc += wn * (last ? wn : pn[u]);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < PRE; u++) pw[u] = pw[u] + (-h) * pv[u];           // (elements beyond n: 0 + (-h) 0, never stored, never summed)
                if (!last) load_vec(A.V + (size_t)(k + 1) * n, pv);
#pragma unroll
                for (int u = 0; u < PRE; u++) if (t0 + u * stride < n) acc += pw[u] * (last ? pw[u] : pv[u]);
            }
        }
        publish(acc, k + 1);
        if (PF && !poller) {
#pragma unroll
            for (int u = 0; u < PRE; u++) pv[u] = pn[u];
            if (k + 2 <= i) load_vec(A.V + (size_t)(k + 2) * n, pn);
        }
    }
    if (!dead) {
        const double nw = collect(i + 1, true);
        dead = sh_dead != 0;
        if (!dead) {
            if (poller && blockIdx.x == 0 && lane == 0) {
                A.st->sc[S_BETA] = nw;
                if (!giv) Hm(A.hs, A.m, i + 1, i) = nw;
                else {
                    double *rs = A.hs + (size_t)A.m * (A.m + 1), *rc = rs + (A.m + 1), *rn = rc + (A.m + 1);
                    double cur = sh_g[0], last = nw, ci, sni, si = sh_g[1], si1 = sh_g[2];
                    gen_rot(cur, last, ci, sni);
                    app_rot(cur, last, ci, sni);
                    Hm(A.hs, A.m, i, i) = cur; Hm(A.hs, A.m, i + 1, i) = last;
                    app_rot(si, si1, ci, sni);
                    rc[i] = ci; rn[i] = sni; rs[i] = si; rs[i + 1] = si1;
                    const double resid = fabs(si1);
                    A.st->sc[S_BETANOM] = resid;
                    if (resid <= sh_g[3]) {
                        backsolve(A.hs, A.m, i);
                        A.st->fl[F_FINAL_COUNT] = i + 1;
                        A.st->sc[S_FINAL_NORM] = resid; A.st->fl[F_FINAL_ITER] = sh_jbase + i; A.st->fl[F_CONVERGED] = 1; A.st->fl[F_DONE] = 1;
                    }
                }
            }
            if (!poller) {
                const double f = 1.0 / nw;
#pragma unroll
                for (int u = 0; u < PRE; u++) {
                    const int t = t0 + u * stride;
                    if (t < n) { A.w[t] = pw[u]; A.vout[t] = f * pw[u]; }
                }
            }
        }
    }
    // the last workgroup out closes the launch: the next one's tag, the counters back to zero
    if (poller && lane == 0) {
        const int out = __hip_atomic_fetch_add(A.st->chain + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (out == (int)gridDim.x - 1) {
            __hip_atomic_store(A.st->chain + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.st->chain + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.st->chain, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- CG with the reductions finished inside their consumers (single rank) ------------------------------------------------
// betanom = sum(part) is re-summed by every block of the direction update (same fixed order everywhere); block 0 takes
// cg_step1's branches and writes the state.  The iteration counter is double-buffered (read F_ITER, write F_ITER2;
// cg_step2 copies it back) so that no block reads a value another block of the same launch is writing; a block that
// observes the freshly written `done` flag returns, which is exactly what the others decide from `stop`.
__global__ void __launch_bounds__(kB) cg_fused_dir_kernel(int n, KState *__restrict__ st, int max_iter, const double *__restrict__ part,
                                                          const double *__restrict__ src, double *__restrict__ d)
{
    __shared__ double sh[4];
    if (st->fl[F_DONE]) return;
    const double betanom = sum_partials(part, sh);
    const int i = st->fl[F_ITER];
    const double r0 = st->sc[S_R0], nom = st->sc[S_NOM];
    const bool stop = (betanom < 0.0) || (betanom <= r0) || (i + 1 > max_iter);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->sc[S_BETANOM] = betanom;
        st->sc[S_FINAL_NORM] = sqrt(betanom);
        st->fl[F_ITER2] = i;
        if (betanom < 0.0) { st->fl[F_FINAL_ITER] = i; st->fl[F_DONE] = 1; }
        else if (betanom <= r0) { st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = i; st->fl[F_DONE] = 1; }
        else if (i + 1 > max_iter) { st->fl[F_ITER2] = i + 1; st->fl[F_DONE] = 1; }   // final_iter stays max_iter
        else { st->fl[F_ITER2] = i + 1; st->sc[S_BETA] = betanom / nom; }
    }
    if (stop) return;
    const double beta = betanom / nom;
    for (int t = blockIdx.x * kB + threadIdx.x; t < n; t += kPart * kB) d[t] = src[t] + beta * d[t];
}

// The direction update, the product A d and the partial sums of (d, A d) in ONE launch, for the systems whose product is launch-bound (a
// row-group product: a group of LPR lanes per row, at most kPart kB rows): three launches of a pass -- cg_fused_dir_kernel, the product,
// gmres_dot0_kernel -- were ~13 of the 40 us a pass takes at the 42 309 unknowns of cube456 x3.  Every workgroup finishes (r, z) itself as
// cg_fused_dir_kernel does; the new direction is not gathered from memory but formed where it is needed, d'_c = z_c + beta d_c with the
// expression of the update (the old direction stays untouched in its buffer: the directions alternate between two), by the lanes that
// spmv_fused_row would give the entry to, summed in its order; lane 0 of a row stores (A d')_row and d'_row.  Workgroup b owns rows
// [256 b, 256 b + 256) -- the elements of block b of gmres_dot0_kernel -- in passes of blockDim / LPR rows, leaves d'_row (A d')_row in LDS
// and adds the 256 up through the tree of block_sum_256(): the partial sums, hence alpha, are those of the three launches bit for bit.
//   ROWS = 64 (4 to 16 lanes per row): a workgroup owns the 64 rows of ONE wavefront of such a block and stores that wavefront's sum -- a
// quarter of a partial sum, which the launch that computes alpha puts together as block_sum_256() would have -- so that the product spreads
// over the chip as spmv_fused's does (662 workgroups at 42 309 rows; with 256 rows each 166 of the 256 CUs did all of it: 8.4 us a launch).
template <int LPR, int ROWS>
__global__ void __launch_bounds__(1024) cg_dir_amul_dot_kernel(int n, KState *__restrict__ st, int max_iter, const double *__restrict__ part_in,
                                                               const double *__restrict__ src, const double *__restrict__ d_old,
                                                               double *__restrict__ d_new, const int *__restrict__ I, const int *__restrict__ J,
                                                               const double *__restrict__ a, double *__restrict__ ad, double *__restrict__ part_out)
{
    __shared__ double sh[4];
    __shared__ double sprod[kB];
    // (the flag, the scalars, the partial sums and the first rows' bounds are requested together, before anything is waited for: one round trip, not four)
    const int done = st->fl[F_DONE], it = st->fl[F_ITER];
    const double r0 = st->sc[S_R0], nom = st->sc[S_NOM];
    double pp = 0.0;
    if (threadIdx.x < kB) pp = pair_of(part_in);
    const int row0 = (int)blockIdx.x * ROWS;
    const int lane = threadIdx.x & (LPR - 1), grp = (int)threadIdx.x / LPR, rpp = (int)blockDim.x / LPR;
    constexpr int U = k::rowgroup::kRowU;
    // A workgroup of 1024 threads covers its 256 rows in LPR / 4 passes; two passes travel together -- row bounds, then the entries, then the
    // gathers of both are requested before anything is waited for (a pass is a chain of four round trips: one after the other was 8.5 us at
    // 42 309 rows where the product alone takes 4.8) -- and the first two are requested before (r, z) is added up: only the products need beta.
    // A lane's sum is formed in spmv_fused_row's order either way: its entries in storage order.
    constexpr int NP = (ROWS == kB && LPR >= 8) ? 2 : 1;
    int row[NP], s[NP], nd[NP];
    double own_s[NP], own_d[NP], sum[NP];
    int jd[NP][U];
    double av[NP][U], xs[NP][U], xd[NP][U];
    auto request_bounds = [&](int p0) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int lr = p0 + p * rpp + grp;
            row[p] = (lr < ROWS && row0 + lr < n) ? row0 + lr : -1;
            s[p] = 0; nd[p] = 0; own_s[p] = 0.0; own_d[p] = 0.0; sum[p] = 0.0;
            if (row[p] >= 0) {
                s[p] = I[row[p]]; nd[p] = I[row[p] + 1] - s[p];
                if (lane == 0) { own_s[p] = src[row[p]]; own_d[p] = d_old[row[p]]; }
            }
        }
    };
    auto request_entries = [&]() {
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = lane + u * LPR;
                jd[p][u] = 0; av[p][u] = 0.0;
                if (q < nd[p]) { jd[p][u] = J[s[p] + q]; av[p][u] = a[s[p] + q]; }
            }
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int q = lane + u * LPR;
                xs[p][u] = 0.0; xd[p][u] = 0.0;
                if (q < nd[p]) { xs[p][u] = src[jd[p][u]]; xd[p][u] = d_old[jd[p][u]]; }
            }
    };
    request_bounds(0);
    if (done) return;
    request_entries();
    // (r, z) as sum_partials() adds it up, by the first kB threads
    if (threadIdx.x < kB) { pp = wave_sum(pp); if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = pp; }
    __syncthreads();
    const double betanom = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    const bool stop = (betanom < 0.0) || (betanom <= r0) || (it + 1 > max_iter);
    if (blockIdx.x == 0 && threadIdx.x == 0) {                        // cg_fused_dir_kernel's branches
        st->sc[S_BETANOM] = betanom;
        st->sc[S_FINAL_NORM] = sqrt(betanom);
        st->fl[F_ITER2] = it;
        if (betanom < 0.0) { st->fl[F_FINAL_ITER] = it; st->fl[F_DONE] = 1; }
        else if (betanom <= r0) { st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = it; st->fl[F_DONE] = 1; }
        else if (it + 1 > max_iter) { st->fl[F_ITER2] = it + 1; st->fl[F_DONE] = 1; }
        else { st->fl[F_ITER2] = it + 1; st->sc[S_BETA] = betanom / nom; }
    }
    if (stop) return;
    const double beta = betanom / nom;
    for (int p0 = 0; p0 < ROWS; p0 += NP * rpp) {
        if (p0 > 0) { request_bounds(p0); request_entries(); }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            if (lane < nd[p]) {                                       // (spmv_fused_row's loop does not run for a lane without an entry)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = lane + u * LPR;
                    const double xv = q < nd[p] ? xs[p][u] + beta * xd[p][u] : 0.0;   // the entry of the new direction, as the update would have stored it
                    sum[p] += av[p][u] * xv;
                }
            }
            for (int q0 = lane + U * LPR; q0 < nd[p]; q0 += U * LPR) {    // rows beyond U entries per lane: the later passes of spmv_fused_row's loop
                int j2[U];
                double a2[U], s2[U], d2[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = q0 + u * LPR;
                    j2[u] = 0; a2[u] = 0.0;
                    if (q < nd[p]) { j2[u] = J[s[p] + q]; a2[u] = a[s[p] + q]; }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = q0 + u * LPR;
                    s2[u] = 0.0; d2[u] = 0.0;
                    if (q < nd[p]) { s2[u] = src[j2[u]]; d2[u] = d_old[j2[u]]; }
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int q = q0 + u * LPR;
                    const double xv = q < nd[p] ? s2[u] + beta * d2[u] : 0.0;
                    sum[p] += a2[u] * xv;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const double tot = k::rowgroup::group_reduce<LPR>(sum[p]);
            if (lane == 0 && row[p] >= 0) {
                double t = 1.0 * tot;                                  // the product's epilogue with its defaults: alpha = 1, no other operand
                t *= 1.0; t += 0.0 * 0.0; t *= 1.0; t += 0.0 * 0.0;
                const double dn = own_s[p] + beta * own_d[p];
                ad[row[p]] = t;
                d_new[row[p]] = dn;
                sprod[row[p] - row0] = dn * t;
            }
        }
    }
    __syncthreads();
    double acc = 0.0;
    if (ROWS == 64) {                                                  // one wavefront's sum: part_out holds quarters (cg_fused_step2_update_kernel adds them up)
        if (threadIdx.x < 64) {
            if (row0 + (int)threadIdx.x < n) acc += sprod[threadIdx.x];
            acc = wave_sum(acc);
            if (threadIdx.x == 0) part_out[blockIdx.x] = acc;
        }
        return;
    }
    if (threadIdx.x < kB) {
        if (row0 + (int)threadIdx.x < n) acc += sprod[threadIdx.x];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) part_out[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// cg_step2 with the dot finished in-kernel (one block)
__global__ void __launch_bounds__(kB) cg_fused_step2_kernel(KState *st, const double *__restrict__ part)
{
    __shared__ double sh[4];
    if (st->fl[F_DONE]) return;
    const double den = sum_partials(part, sh);
    if (threadIdx.x == 0) {
        st->fl[F_ITER] = st->fl[F_ITER2];
        st->sc[S_DEN] = den;
        if (den == 0.0) { st->fl[F_FINAL_ITER] = st->fl[F_ITER]; st->fl[F_DONE] = 1; return; }
        st->sc[S_NOM] = st->sc[S_BETANOM];
        st->sc[S_ALPHA] = st->sc[S_NOM] / den;
    }
}

// cg_step2 and the update of x and r that mfem::CGSolver makes at the top of the NEXT pass, in one launch: every block adds the partial sums of
// (d, A d) up itself (the same fixed order everywhere), block 0 takes cg_step2's branch and writes the state, and the update follows at once
// -- only den == 0 ends the loop between the two statements, and then nothing is applied.  One launch of a pass less (4.4 us of the 40 a pass
// takes at the 42 309 unknowns of cube456 x3); alpha = nom / den is the same division, the iterates the same bits.
// QUARTERS: part holds the four wavefront sums of every block (cg_dir_amul_dot_kernel<LPR, 64>): a partial sum is (q0 + q1) + (q2 + q3), block_sum_256()'s last line
template <bool QUARTERS>
__global__ void __launch_bounds__(kB) cg_fused_step2_update_kernel(int n, KState *__restrict__ st, const double *__restrict__ part, double *__restrict__ x,
                                                                   const double *__restrict__ d, double *__restrict__ r, const double *__restrict__ z)
{
    __shared__ double sh[4];
    // (the flag, the scalars, the partial sums and the thread's first elements are requested together: one round trip)
    const int done = st->fl[F_DONE];
    const double betanom = st->sc[S_BETANOM];
    const int iter2 = st->fl[F_ITER2];
    double pp;
    if (QUARTERS) {
        typedef double d2k __attribute__((ext_vector_type(2)));
        const d2k *q = reinterpret_cast<const d2k *>(part + 8 * threadIdx.x);   // (64 aligned bytes a thread: four 16-byte loads)
        const d2k q01 = q[0], q23 = q[1], q45 = q[2], q67 = q[3];
        pp = ((q01.x + q01.y) + (q23.x + q23.y)) + ((q45.x + q45.y) + (q67.x + q67.y));
    } else pp = pair_of(part);
    const int t0 = blockIdx.x * kB + threadIdx.x;
    double x0 = 0.0, d0 = 0.0, rr0 = 0.0, z0 = 0.0;
    if (t0 < n) { x0 = x[t0]; d0 = d[t0]; rr0 = r[t0]; z0 = z[t0]; }
    if (done) return;
    const double den = block_sum_256(pp, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->fl[F_ITER] = iter2;
        st->sc[S_DEN] = den;
        if (den == 0.0) { st->fl[F_FINAL_ITER] = iter2; st->fl[F_DONE] = 1; }
        else { st->sc[S_NOM] = betanom; st->sc[S_ALPHA] = betanom / den; }
    }
    if (den == 0.0) return;
    const double alpha = betanom / den;
    if (t0 < n) { x[t0] = x0 + alpha * d0; r[t0] = rr0 + (-alpha) * z0; }
    for (int t = t0 + kPart * kB; t < n; t += kPart * kB) { x[t] = x[t] + alpha * d[t]; r[t] = r[t] + (-alpha) * z[t]; }
}

__global__ void gmres_norm_kernel(KState *st, double *hs, int m, int i)
{
    if (st->fl[F_DONE]) return;
    const double nw = sqrt(st->sc[S_TMP]);
    Hm(hs, m, i + 1, i) = nw;
    st->sc[S_BETA] = nw;   // divisor of the next basis vector
}

// One wavefront: column i of H and the rotations so far are fetched side by side into LDS, thread 0 applies them from there (the same
// operations in the same order), the column goes back side by side.  (One thread working in global memory paid a memory round trip per
// rotation: 9.2 us a step at i ~ 16, measured, against the launch floor.)  Dynamic LDS: 4 m + 6 doubles.
__global__ void __launch_bounds__(64) gmres_givens_kernel(KState *st, double *hs, int m, int i)
{
    extern __shared__ double gv[];
    double *hin = gv, *hout = hin + (m + 2), *c = hout + (m + 2), *sg = c + (m + 1);
    int step_id = 0;
    const int done = st->fl[F_DONE];
    if (threadIdx.x == 0) { step_id = st->fl[F_STEP] + 1; st->fl[F_STEP] = step_id; }   // (counted whether or not the solve is over: a step's update kernel runs for ITS step only)
    if (done) return;
    double *s = hs + (size_t)m * (m + 1), *cs = s + (m + 1), *sn = cs + (m + 1);
    for (int k = threadIdx.x; k <= i + 1; k += 64) hin[k] = Hm(hs, m, k, i);
    for (int k = threadIdx.x; k < i; k += 64) { c[k] = cs[k]; sg[k] = sn[k]; }
    double si = 0.0, si1 = 0.0, thresh = 0.0;
    int jbase = 0;
    if (threadIdx.x == 0) { si = s[i]; si1 = s[i + 1]; thresh = st->sc[S_THRESH]; jbase = st->fl[F_JBASE]; }
    __syncthreads();
    bool converged = false;
    double resid = 0.0;
    if (threadIdx.x == 0) {
        double cur = hin[0];
        for (int k = 0; k < i; k++) {                       // app_rot(H(k,i), H(k+1,i), cs[k], sn[k])
            double nxt = hin[k + 1];
            app_rot(cur, nxt, c[k], sg[k]);
            hout[k] = cur;
            cur = nxt;
        }
        double last = hin[i + 1], ci, sni;
        gen_rot(cur, last, ci, sni);
        app_rot(cur, last, ci, sni);
        hout[i] = cur; hout[i + 1] = last;
        app_rot(si, si1, ci, sni);
        cs[i] = ci; sn[i] = sni; s[i] = si; s[i + 1] = si1;
        resid = fabs(si1);
        converged = resid <= thresh;
        st->sc[S_BETANOM] = resid;
    }
    __syncthreads();
    for (int k = threadIdx.x; k <= i + 1; k += 64) Hm(hs, m, k, i) = hout[k];
    __syncthreads();
    if (converged) {
        backsolve(hs, m, i);
        st->fl[F_PENDING] = step_id;   // the update kernel of this step applies x += sum_{q<=i} y_q v_q ...
        st->fl[F_FINAL_COUNT] = i + 1; // ... or the one the host launches when it reads `done` (gmres_final_update_kernel)
        st->sc[S_FINAL_NORM] = resid; st->fl[F_FINAL_ITER] = jbase + i; st->fl[F_CONVERGED] = 1; st->fl[F_DONE] = 1;
    }
}

__global__ void gmres_restart_backsolve_kernel(KState *st, double *hs, int m, int k)
{
    const int step_id = ++st->fl[F_STEP];
    if (st->fl[F_DONE]) return;
    backsolve(hs, m, k);
    st->fl[F_PENDING] = step_id;
}

// x += sum_{q < count} y_q V_q    (runs only for the step that set F_PENDING; `done` may already be set by that step)
__global__ void gmres_update_kernel(int n, const KState *__restrict__ st, const double *__restrict__ hs, int m, int count,
                                    const double *__restrict__ V, double *__restrict__ x)
{
    if (st->fl[F_PENDING] != st->fl[F_STEP]) return;
    const double *y = hs + (size_t)m * (m + 1) + 3 * (m + 1);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {      // (a small grid: all but one of a solve's launches return at once)
        double acc = x[t];
        for (int q = 0; q < count; q++) acc = acc + y[q] * V[(size_t)q * n + t];
        x[t] = acc;
    }
}

// The update of the step that converged, launched ONCE by the host when it has read `done` (a launch of every step that returned at once in
// all but one of them was 4.8 us of the ~260 a step of the Darcy x3 solve takes): x += sum_{q < F_FINAL_COUNT} y_q V_q, nothing where the solve
// ended otherwise (the restart's update has been applied by the restart).
__global__ void gmres_final_update_kernel(int n, const KState *__restrict__ st, const double *__restrict__ hs, int m, const double *__restrict__ V,
                                          double *__restrict__ x)
{
    const int count = st->fl[F_FINAL_COUNT];
    if (count <= 0) return;
    const double *y = hs + (size_t)m * (m + 1) + 3 * (m + 1);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        double acc = x[t];
        for (int q = 0; q < count; q++) acc = acc + y[q] * V[(size_t)q * n + t];
        x[t] = acc;
    }
}

__global__ void gmres_restart_norm_kernel(KState *st, int j)
{
    if (st->fl[F_DONE]) return;
    const double beta = sqrt(st->sc[S_TMP]);
    st->sc[S_GBETA] = beta;
    st->sc[S_FINAL_NORM] = beta;
    st->sc[S_BETANOM] = beta;
    if (beta <= st->sc[S_THRESH]) { st->fl[F_FINAL_ITER] = j; st->fl[F_CONVERGED] = 1; st->fl[F_DONE] = 1; }
}

// ---- generic predicated vector kernels for the two solvers below ---------------------------------------------------------
// coefficient: slot < 0 -> 1.0, otherwise sign * st->sc[slot]
__device__ inline double kcoef(const KState *st, int slot, double sign) { return slot < 0 ? sign : sign * st->sc[slot]; }
// out = ca * x + cb * y   (out may alias x or y); pending_id > 0: also runs for the step that set F_PENDING to it
__global__ void lin2_kernel(int n, const KState *__restrict__ st, int pending_id, int sa, double sga, const double *x, int sb, double sgb, const double *y,
                            double *out)
{
    if (st->fl[F_DONE] && !(pending_id > 0 && st->fl[F_PENDING] == pending_id)) return;
    const double ca = kcoef(st, sa, sga), cb = kcoef(st, sb, sgb);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (x ? ca * x[i] : 0.0) + cb * y[i];
}
// out = x + ca * y  (exactly mfem's add(x, a, y, out) / Vector::Add: no product on the first operand)
__global__ void axpy_kernel(int n, const KState *__restrict__ st, int pending_id, const double *x, int sa, double sga, const double *y, double *out)
{
    if (st->fl[F_DONE] && !(pending_id > 0 && st->fl[F_PENDING] == pending_id)) return;
    const double ca = kcoef(st, sa, sga);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] + ca * y[i];
}
// p = r + beta (p - omega v)   (BiCGSTAB: add(p, -omega, v, p); add(r, beta, p, p))
__global__ void bicg_p_kernel(int n, const KState *__restrict__ st, const double *__restrict__ r, const double *__restrict__ v, double *__restrict__ p)
{
    if (st->fl[F_DONE]) return;
    const double beta = st->sc[S_BETA], momega = -st->sc[S_OMEGA];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const double t = p[i] + momega * v[i]; p[i] = r[i] + beta * t; }
}
// dst = src * (1 / sc[slot])     (mfem Vector::operator/=)
__global__ void scale_recip_kernel(int n, const KState *__restrict__ st, int slot, double *__restrict__ v)
{
    if (st->fl[F_DONE]) return;
    const double m = 1.0 / st->sc[slot];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * m;
}
__global__ void copy_pred_kernel(int n, const KState *__restrict__ st, const double *__restrict__ src, double *__restrict__ dst)
{
    if (st->fl[F_DONE]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- BiCGSTAB scalar kernels (mfem::BiCGSTABSolver::Mult) ------------------------------------------------------------------
__global__ void bicg_init_kernel(KState *st, double rel_tol, double abs_tol, int max_iter)
{
    const double resid = sqrt(st->sc[S_TMP]);
    st->sc[S_RESID] = resid; st->sc[S_FINAL_NORM] = resid;
    st->sc[S_GOAL] = fmax(resid * rel_tol, abs_tol);
    st->sc[S_RHO2] = 1.0; st->sc[S_ALPHA] = 1.0; st->sc[S_OMEGA] = 1.0;
    st->fl[F_DONE] = 0; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = max_iter; st->fl[F_PENDING] = 0;
    if (resid <= st->sc[S_GOAL]) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = 0; }
}
__global__ void bicg_rho_kernel(KState *st, int i)
{
    if (st->fl[F_DONE]) return;
    const double rho1 = st->sc[S_TMP];
    st->sc[S_RHO1] = rho1;
    if (rho1 == 0.0) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = i; return; }
    if (i > 1) st->sc[S_BETA] = (rho1 / st->sc[S_RHO2]) * (st->sc[S_ALPHA] / st->sc[S_OMEGA]);
}
__global__ void bicg_alpha_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    st->sc[S_ALPHA] = st->sc[S_RHO1] / st->sc[S_TMP];
}
// resid = |s|; converged in the middle of the step: x += alpha phat still has to run (F_PENDING = id)
__global__ void bicg_mid_kernel(KState *st, int i, int id)
{
    if (st->fl[F_DONE]) return;
    const double resid = sqrt(st->sc[S_TMP]);
    st->sc[S_RESID] = resid; st->sc[S_FINAL_NORM] = resid;
    if (resid < st->sc[S_GOAL]) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = i; st->fl[F_PENDING] = id; }
}
__global__ void bicg_ts_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    st->sc[S_TS] = st->sc[S_TMP];
}
__global__ void bicg_omega_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    st->sc[S_OMEGA] = st->sc[S_TS] / st->sc[S_TMP];
}
__global__ void bicg_end_kernel(KState *st, int i)
{
    if (st->fl[F_DONE]) return;
    st->sc[S_RHO2] = st->sc[S_RHO1];
    const double resid = sqrt(st->sc[S_TMP]);
    st->sc[S_RESID] = resid; st->sc[S_FINAL_NORM] = resid;
    if (resid < st->sc[S_GOAL]) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = i; return; }
    if (st->sc[S_OMEGA] == 0.0) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = i; }
}

// ---- MINRES scalar kernels (mfem::MINRESSolver::Mult) ----------------------------------------------------------------------
__global__ void minres_init_kernel(KState *st, double rel_tol, double abs_tol, int max_iter)
{
    const double eta = sqrt(st->sc[S_TMP]);
    st->sc[S_ETA] = eta; st->sc[S_BETA] = eta;
    st->sc[S_GAMMA0] = 1.0; st->sc[S_GAMMA1] = 1.0; st->sc[S_SIGMA0] = 0.0; st->sc[S_SIGMA1] = 0.0;
    st->sc[S_GOAL] = fmax(rel_tol * eta, abs_tol);
    st->sc[S_FINAL_NORM] = fabs(eta);
    st->fl[F_DONE] = 0; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = max_iter; st->fl[F_PENDING] = 0;
    if (eta <= st->sc[S_GOAL]) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = 0; }
}
__global__ void minres_alpha_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    st->sc[S_ALPHA] = st->sc[S_TMP];
}
// delta, rho3, rho2 from the OLD beta; then beta = sqrt(TMP), rho1, the w-update coefficients, gamma, x-update coefficient
__global__ void minres_beta_kernel(KState *st)
{
    if (st->fl[F_DONE]) return;
    const double alpha = st->sc[S_ALPHA], beta_old = st->sc[S_BETA];
    const double gamma0 = st->sc[S_GAMMA0], gamma1 = st->sc[S_GAMMA1], sigma0 = st->sc[S_SIGMA0], sigma1 = st->sc[S_SIGMA1];
    const double delta = gamma1 * alpha - gamma0 * sigma1 * beta_old;
    const double rho3 = sigma0 * beta_old;
    const double rho2 = sigma1 * alpha + gamma0 * gamma1 * beta_old;
    const double beta = sqrt(st->sc[S_TMP]);
    const double rho1 = hypot(delta, beta);
    st->sc[S_BETA] = beta; st->sc[S_DELTA] = delta; st->sc[S_MRHO1] = rho1; st->sc[S_MRHO2] = rho2; st->sc[S_MRHO3] = rho3;
    st->sc[S_C0] = 1.0 / rho1; st->sc[S_C1] = -rho2 / rho1; st->sc[S_C2] = -rho3 / rho1;
    st->sc[S_GAMMA0] = gamma1;
    st->sc[S_GAMMA1] = delta / rho1;
    st->sc[S_TS] = st->sc[S_GAMMA1] * st->sc[S_ETA];          // coefficient of x.Add(gamma1*eta, w0)
}
__global__ void minres_eta_kernel(KState *st, int it)
{
    if (st->fl[F_DONE]) return;
    st->sc[S_SIGMA0] = st->sc[S_SIGMA1];
    st->sc[S_SIGMA1] = st->sc[S_BETA] / st->sc[S_MRHO1];
    const double eta = -st->sc[S_SIGMA1] * st->sc[S_ETA];
    st->sc[S_ETA] = eta;
    st->sc[S_FINAL_NORM] = fabs(eta);
    if (fabs(eta) <= st->sc[S_GOAL]) { st->fl[F_DONE] = 1; st->fl[F_CONVERGED] = 1; st->fl[F_FINAL_ITER] = it; }
}

// ---- stationary (StationarySolver::Mult) --------------------------------------------------------------------------------
__global__ void stat_init_kernel(KState *st, int max_iter)
{
    st->sc[S_NORMK1] = sqrt(st->sc[S_TMP]);
    st->sc[S_RATIO] = 1.0;
    st->sc[S_FINAL_NORM] = st->sc[S_NORMK1];
    st->fl[F_DONE] = 0; st->fl[F_CONVERGED] = 0; st->fl[F_FINAL_ITER] = 0; st->fl[F_ITER] = 0;
    (void)max_iter;
}

__global__ void stat_step_kernel(KState *st, double rel_tol, double abs_tol)
{
    if (st->fl[F_DONE]) return;
    const double nk = sqrt(st->sc[S_TMP]), nk1 = st->sc[S_NORMK1];
    st->fl[F_ITER] += 1;
    st->fl[F_FINAL_ITER] = st->fl[F_ITER];
    st->sc[S_RATIO] *= nk / nk1;
    st->sc[S_FINAL_NORM] = nk;
    if (nk < abs_tol || st->sc[S_RATIO] < rel_tol) { st->fl[F_CONVERGED] = 1; st->fl[F_DONE] = 1; return; }
    if (nk / nk1 == 1.0) { st->fl[F_DONE] = 1; return; }   // "Computed a zero correction! Stopping..."
    st->sc[S_NORMK1] = nk;
}

// sol += c ; (pred)
__global__ void add_pred_kernel(int n, const KState *__restrict__ st, double *__restrict__ x, const double *__restrict__ c)
{
    if (st->fl[F_DONE]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] + c[i];
}
__global__ void sub_pred_kernel(int n, const KState *__restrict__ st, double *__restrict__ r, const double *__restrict__ t)
{
    if (st->fl[F_DONE]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r[i] = r[i] - t[i];
}

}  // namespace

// (a source's kernels are loaded when its first kernel is used: a process's first solve paid ~9 ms for this file's inside what a caller times
//  as the solve -- 13.7 against 4.2 ms for the reference's default example; elag_hip_init touches it instead, like the other large sources)
namespace elag {
void preload_krylov_kernels()
{
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(cg_init1_kernel));
}
}  // namespace elag

struct elag_hip_krylov_s {
    ParCSR *A = nullptr;
    // Preconditioned by a hierarchy that renumbered its levels (and built on this very operator), the whole iteration runs in
    // the hierarchy's numbering: Aop is the hierarchy's level-0 operator, the right-hand side and the initial guess are
    // permuted once on entry and the solution once on exit -- instead of a gather and a scatter around every V-cycle and an
    // A p product on the less local numbering.  Dot products only change their order of summation.
    ParCSR *Aop = nullptr;
    const int *d_perm = nullptr;     // new -> old, device (owned by the hierarchy)
    double *bperm = nullptr;
    int solver = ELAG_HIP_KRYLOV_CG;
    int max_iter = 10, restart = 50, check_every = 8;
    double rel_tol = 0.0, abs_tol = 0.0;
    bool use_graph = true;
    // preconditioner
    int prec_kind = ELAG_HIP_PREC_NONE;
    elag_hip_smoother_s *psm = nullptr;
    elag_hip_hierarchy_s *phier = nullptr;
    elag_hip_coarse_s *pcoarse = nullptr;
    // device state
    KState *st = nullptr;
    // host mirrors of st->fl and of the leading scalars of st->sc (filled from the fetched words: plain memory -- a solver's construction used
    // to pin two blocks of its own and hipMalloc its state block, ~0.4 ms each time and ~9 ms the first time in a process)
    int h_flags[F_COUNT] = {};
    double h_sc[S_COUNT] = {};
    double *x = nullptr, *r = nullptr, *z = nullptr, *d = nullptr, *w = nullptr, *scr = nullptr;
    double *V = nullptr, *Z = nullptr, *hs = nullptr, *part = nullptr;
    u64k *gm_gran = nullptr;                      // gmres_chain_kernel's tagged sums, [restart + 2][kChainG][2]
    int gm_chain = std::getenv("ELAG_HIP_GMRES_CHAIN") ? atoi(std::getenv("ELAG_HIP_GMRES_CHAIN")) : 1;   // (read when the solver is made: a test compares the chains)
    // the update of the converged step: one launch when the host has read `done` (default) or a launch in every step that returns at once in all but one
    bool gm_update_each = std::getenv("ELAG_HIP_GMRES_UPDATE_EACH_STEP") && atoi(std::getenv("ELAG_HIP_GMRES_UPDATE_EACH_STEP"));
    // the step's rotations inside gmres_chain_kernel (0: gmres_givens_kernel, a launch of its own; needed by the update in every step, which counts the steps there)
    bool gm_chain_givens = !(std::getenv("ELAG_HIP_GMRES_CHAIN_GIVENS") && !atoi(std::getenv("ELAG_HIP_GMRES_CHAIN_GIVENS")));
    bool cg_tail = !(std::getenv("ELAG_HIP_CG_TAIL_UPDATE") && !atoi(std::getenv("ELAG_HIP_CG_TAIL_UPDATE")));   // CG: the update of x, r in the launch that computes alpha (0: a launch of its own)
    double *e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // BiCGSTAB / MINRES work vectors
    hipGraphExec_t cg_iter = nullptr;
    // CG on a launch-bound system: direction update, product and (d, A d) in one launch (cg_dir_amul_dot_kernel); the directions alternate
    // between d and d2, the product goes to ad; a replayed pass is one graph per parity.  ELAG_HIP_CG_LEAN=0: the three launches
    bool cg_lean_env = !(std::getenv("ELAG_HIP_CG_LEAN") && !atoi(std::getenv("ELAG_HIP_CG_LEAN")));
    bool cg_lean = false;
    double *d2 = nullptr, *ad = nullptr, *qpart = nullptr;   // qpart: the wavefront sums of (d, A d), four per block of gmres_dot0_kernel
    int cg_par = 0, cg_lpr = 1;                   // cg_lpr: the lanes per row the product had when the solver was made (the kernel reads I, J, a itself: any count is a correct product)
    hipGraphExec_t cg_iter_par[2] = {nullptr, nullptr};
    std::vector<hipGraphExec_t> gm_step;          // GMRES: one replayable step per position of the restart cycle
    bool gm_graph_ok = true, gm_flex = false, gm_graphs_chain = false;   // gm_graphs_chain: some captured step holds a gmres_chain_kernel
    std::string launch_mode = "plain";          // what elag_hip_krylov_launch_mode reports (set by the last Mult)
    std::string gm_graph_note;                    // why the step graphs are off for this solver (empty: they are on, or were never asked for)
    // results of the last Mult
    int converged = 0, final_iter = 0;
    double final_norm = 0.0;

    int n() const { return A->nrows(); }
    bool distributed() const { return is_distributed(*A); }
    bool has_prec() const { return prec_kind != ELAG_HIP_PREC_NONE; }

    ~elag_hip_krylov_s()
    {
        if (cg_iter) (void)hipGraphExecDestroy(cg_iter);
        for (auto g : cg_iter_par) if (g) (void)hipGraphExecDestroy(g);
        dev_free(d2); dev_free(ad); dev_free(qpart);
        for (auto g : gm_step) if (g) (void)hipGraphExecDestroy(g);
        dev_free(x); dev_free(r); dev_free(z); dev_free(d); dev_free(w); dev_free(scr); dev_free(V); dev_free(Z); dev_free(hs); dev_free(part);
        dev_free(bperm); dev_free(gm_gran);
        for (auto &p : e) dev_free(p);
        dev_free(st);
    }

    void setup()
    {
        const size_t nn = (size_t)n();
        st = reinterpret_cast<KState *>(dev_alloc<double>((sizeof(KState) + sizeof(double) - 1) / sizeof(double)));
        zero_async(st, sizeof(KState));
        Aop = A;
        if (prec_kind == ELAG_HIP_PREC_HIERARCHY && !distributed() && std::getenv("ELAG_HIP_KRYLOV_NATURAL") == nullptr) {
            ParCSR *Ai = nullptr;
            if (hierarchy_internal_numbering(phier, A, &Ai, &d_perm)) { Aop = Ai; bperm = dev_alloc<double>(nn); }
            else d_perm = nullptr;
        }
        k::prepare_spmv(Aop->diag);
        x = dev_alloc<double>(nn); r = dev_alloc<double>(nn); z = dev_alloc<double>(nn); scr = dev_alloc<double>(nn);
        if (solver == ELAG_HIP_KRYLOV_CG) {
            d = dev_alloc<double>(nn); part = dev_alloc<double>(2 * (size_t)kPart);
            // (the product as spmv_fused would run it: the row-group kernel on a rank-local matrix, not the stream / x-cache / sign forms)
            const DevCSR &D = Aop->diag;
            cg_lean = cg_lean_env && cg_tail && !distributed() && !Aop->needs_exchange() && nn >= 1 && nn <= (size_t)kPart * kB && D.lanes_per_row >= 1 &&
                      D.lanes_per_row <= 16 && D.sign_state != 1 && D.xc_state != 1;
            if (cg_lean) {
                d2 = dev_alloc<double>(nn); ad = dev_alloc<double>(nn); cg_lpr = D.lanes_per_row;
                zero_async(part, sizeof(double) * 2 * (size_t)kPart);   // (the blocks beyond the last row: written by nobody, read as zeros)
                qpart = dev_alloc<double>(4 * (size_t)kPart);
                zero_async(qpart, sizeof(double) * 4 * (size_t)kPart);
            }
        }
        if (solver == ELAG_HIP_KRYLOV_STATIONARY) { d = dev_alloc<double>(nn); }
        if (solver == ELAG_HIP_KRYLOV_BICGSTAB || solver == ELAG_HIP_KRYLOV_MINRES)
            for (auto &p : e) p = dev_alloc<double>(nn);
        if (solver == ELAG_HIP_KRYLOV_GMRES || solver == ELAG_HIP_KRYLOV_FGMRES) {
            ELAG_REQUIRE(restart >= 1, "\"Restart size\" must be >= 1");
            ELAG_REQUIRE(restart <= 2000, "\"Restart size\" %d: at most 2000 here (a column of the Hessenberg matrix and the rotations are worked in LDS)", restart);
            const size_t m = (size_t)restart;
            w = dev_alloc<double>(nn);
            V = dev_alloc<double>((m + 1) * nn);
            if (solver == ELAG_HIP_KRYLOV_FGMRES) Z = dev_alloc<double>(m * nn);
            hs = dev_alloc<double>(m * (m + 1) + 4 * (m + 1));
            part = dev_alloc<double>(2 * (size_t)kPart);
            if (!distributed() && gm_chain) {
                gm_gran = dev_alloc<u64k>((m + 2) * (size_t)kChainG * 2);
                zero_async(gm_gran, sizeof(u64k) * (m + 2) * (size_t)kChainG * 2);
            }
        }
    }

    // z_out = B r_in from the zero guess (the reference sets iterative_mode = false on the preconditioner: KrylovSolver.cpp:79-83)
    void prec(const double *rin, double *zout)
    {
        if (prec_kind == ELAG_HIP_PREC_SMOOTHER) {
            const int nf = psm->flips(true);
            double *c = (nf & 1) ? scr : zout, *a = (nf & 1) ? zout : scr;
            psm->apply(rin, c, a, true, false);
            if (c != zout) fail(ELAG_HIP_ERR_STATE, "internal: smoother result in scratch");
        } else if (prec_kind == ELAG_HIP_PREC_HIERARCHY) {
            if (d_perm) hierarchy_apply_internal(phier, rin, zout);
            else hierarchy_apply_raw(phier, rin, zout, true);
        } else if (prec_kind == ELAG_HIP_PREC_COARSE) {
            pcoarse->apply(rin, zout);
        } else {
            dev_copy(zout, rin, sizeof(double) * (size_t)n(), ctx().stream);
        }
    }

    void amul(const double *in, double *out)
    {
        k::SpmvArgs a;
        a.x = in; a.alpha = 1.0; a.out = out;
        par_spmv(*Aop, a);
    }
    void residual(const double *b, const double *xin, double *out)   // out = b - A x
    {
        k::SpmvArgs a;
        a.x = xin; a.alpha = -1.0; a.f = b; a.gamma = 1.0; a.out = out;
        par_spmv(*Aop, a);
    }

    // (u, v) summed over the ranks, left in st->sc[S_TMP]
    void dot(const double *u, const double *v)
    {
        Context &c = ctx();
        k::vec_dot(n(), u, v, &st->sc[S_TMP]);
        if (distributed() && comm_active()) {
            ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
            ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
            comm_allreduce_device(&st->sc[S_TMP], 1, 0, c.comm_stream);
            ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
            ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
        }
    }

    // ELAG_HIP_KRYLOV_TIMING=1: host time of a solve split into enqueueing and waiting in poll() (printed by solve())
    double t_poll = 0.0;
    int n_poll = 0;
    // the flags and the scalars the host looks at (h_sc beyond S_FINAL_NORM is not refreshed), in one round trip
    void fetch_state()
    {
        int w[kPollWords];
        fetch_ints(st->fl, kPollWords, w);
        std::memcpy(h_flags, w, sizeof(int) * F_COUNT);
        std::memcpy(h_sc, w + F_COUNT, sizeof(double) * (S_FINAL_NORM + 1));
    }
    bool poll()
    {
        const auto t0 = std::chrono::steady_clock::now();
        fetch_state();
        t_poll += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        n_poll++;
        return h_flags[F_DONE] != 0;
    }

    void fetch_results()
    {
        fetch_state();
        converged = h_flags[F_CONVERGED]; final_iter = h_flags[F_FINAL_ITER]; final_norm = h_sc[S_FINAL_NORM];
    }

    bool capturable() const
    {
        if (!use_graph || distributed() || A->needs_exchange()) return false;
        // an iteration is the preconditioner's cycle plus half a dozen small kernels and a graph launch costs the same fixed few
        // microseconds: where the hierarchy measured its cycle faster as plain launches, the iteration is issued that way too
        if (prec_kind == ELAG_HIP_PREC_HIERARCHY) return hierarchy_is_local(phier) && !hierarchy_prefers_plain_launches(phier);
        // a whitelist: a solver replays inside a graph only if it says that its apply() is nothing but launches on the compute
        // stream -- one that reads a count back or allocates would spin for the poll budget under capture and then invalidate it
        if (prec_kind == ELAG_HIP_PREC_SMOOTHER) return psm->graph_capturable();
        if (prec_kind == ELAG_HIP_PREC_COARSE) return !pcoarse->replicated && !pcoarse->custom_apply;
        return prec_kind == ELAG_HIP_PREC_NONE;
    }

    // single rank: the update of x and r rides on the launch that computes its alpha (cg_fused_step2_update_kernel); run_cg applies the first pass's itself
    bool cg_tail_update() const { return !distributed() && cg_tail; }
    void cg_iteration()
    {
        hipStream_t s = ctx().stream;
        const int nn = n();
        if (!cg_tail_update()) hipLaunchKernelGGL(cg_update_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, x, (const double *)d, r, (const double *)z);
        if (!distributed()) {
            // single rank: each dot is one partial-sum kernel, finished inside the kernel that consumes it
            const double *src = has_prec() ? z : r;
            if (has_prec()) prec(r, z);
            hipLaunchKernelGGL(gmres_dot0_kernel, dim3(kPart), dim3(kB), 0, s, nn, (const KState *)st, (const double *)r, src, part);
            if (cg_lean) {
                const double *dold = cg_par ? d2 : d;
                double *dnew = cg_par ? d : d2;
                const DevCSR &D = Aop->diag;
                // 4 to 16 lanes per row: a workgroup per wavefront of a block of gmres_dot0_kernel (64 rows), quarters of the partial sums; 1 or 2: a workgroup per block
                const bool quarters = cg_lpr >= 4;
                const int rows = quarters ? 64 : kB;
                const dim3 g((unsigned)((nn + rows - 1) / rows)), b((unsigned)std::min(1024, rows * cg_lpr));
                double *pout = quarters ? qpart : part + kPart;
#define ELAG_CG_LEAN(L, R) hipLaunchKernelGGL((cg_dir_amul_dot_kernel<L, R>), g, b, 0, s, nn, st, max_iter, (const double *)part, src, dold, dnew, (const int *)D.I, \
                                              (const int *)D.J, (const double *)D.a, ad, pout)
                switch (cg_lpr) {
                case 1: ELAG_CG_LEAN(1, kB); break;
                case 2: ELAG_CG_LEAN(2, kB); break;
                case 4: ELAG_CG_LEAN(4, 64); break;
                case 8: ELAG_CG_LEAN(8, 64); break;
                default: ELAG_CG_LEAN(16, 64); break;
                }
#undef ELAG_CG_LEAN
                if (quarters) hipLaunchKernelGGL(cg_fused_step2_update_kernel<true>, dim3(kPart), dim3(kB), 0, s, nn, st, (const double *)qpart, x, (const double *)dnew, r, (const double *)ad);
                else hipLaunchKernelGGL(cg_fused_step2_update_kernel<false>, dim3(kPart), dim3(kB), 0, s, nn, st, (const double *)(part + kPart), x, (const double *)dnew, r, (const double *)ad);
                ELAG_LAUNCH_CHECK();
                cg_par ^= 1;
                return;
            }
            hipLaunchKernelGGL(cg_fused_dir_kernel, dim3(kPart), dim3(kB), 0, s, nn, st, max_iter, (const double *)part, src, d);
            amul(d, z);
            hipLaunchKernelGGL(gmres_dot0_kernel, dim3(kPart), dim3(kB), 0, s, nn, (const KState *)st, (const double *)d, (const double *)z, part + kPart);
            if (cg_tail_update())
                hipLaunchKernelGGL(cg_fused_step2_update_kernel<false>, dim3(kPart), dim3(kB), 0, s, nn, st, (const double *)(part + kPart), x, (const double *)d, r, (const double *)z);
            else hipLaunchKernelGGL(cg_fused_step2_kernel, dim3(1), dim3(kB), 0, s, st, (const double *)(part + kPart));
            ELAG_LAUNCH_CHECK();
            return;
        }
        if (has_prec()) { prec(r, z); dot(r, z); } else dot(r, r);
        hipLaunchKernelGGL(cg_step1_kernel, dim3(1), dim3(1), 0, s, st, max_iter);
        hipLaunchKernelGGL(cg_dir_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, (const double *)(has_prec() ? z : r), d);
        amul(d, z);
        dot(d, z);
        hipLaunchKernelGGL(cg_step2_kernel, dim3(1), dim3(1), 0, s, st);
        ELAG_LAUNCH_CHECK();
    }

    void run_cg(const double *b, bool iterative)
    {
        hipStream_t s = ctx().stream;
        const int nn = n();
        if (iterative) residual(b, x, r);
        else {
            dev_copy(r, b, sizeof(double) * (size_t)nn, s);
            k::vec_fill(nn, 0.0, x);
        }
        if (has_prec()) { prec(r, z); dev_copy(d, z, sizeof(double) * (size_t)nn, s); }
        else dev_copy(d, r, sizeof(double) * (size_t)nn, s);
        dot(d, r);
        hipLaunchKernelGGL(cg_init1_kernel, dim3(1), dim3(1), 0, s, st, rel_tol, abs_tol, max_iter);
        amul(d, z);
        dot(z, d);
        hipLaunchKernelGGL(cg_init2_kernel, dim3(1), dim3(1), 0, s, st);
        ELAG_LAUNCH_CHECK();
        if (poll()) return;
        const bool graph = capturable();
        launch_mode = graph ? "graph" : "plain";
        if (cg_tail_update())   // mfem's first statement of the first pass; every later pass's is applied by the launch that computes its alpha
            hipLaunchKernelGGL(cg_update_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, x, (const double *)d, r, (const double *)z);
        cg_par = 0;                                   // (the first direction is in d)
        auto capture_pass = [&](hipGraphExec_t *exec) {
            hipGraph_t g = nullptr;
            ELAG_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            try { cg_iteration(); } catch (...) { (void)hipStreamEndCapture(s, &g); if (g) (void)hipGraphDestroy(g); throw; }
            ELAG_HIP_CHECK(hipStreamEndCapture(s, &g));
            ELAG_HIP_CHECK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
        };
        if (graph && !cg_lean && !cg_iter) capture_pass(&cg_iter);
        if (graph && cg_lean && !cg_iter_par[0]) {    // one graph per parity of the direction buffers (a capture enqueues nothing: the parity is put back)
            capture_pass(&cg_iter_par[0]);
            capture_pass(&cg_iter_par[1]);
            cg_par = 0;
        }
        const int total = std::max(1, max_iter);   // mfem's loop body runs once even with max_iter == 0
        // Iterations enqueued past convergence are no-ops for the iterate but still run the (unpredicated) preconditioner,
        // so the batch between two reads of the state is sized from the observed convergence rate: the host reads
        // (B r, r) with the flag and enqueues about as many iterations as the rate predicts are still needed.
        // (the first rate comes from the initial (B r, r), which the poll above has fetched: without it the second batch repeated the first one's
        //  size -- 4 + 4 passes for the 6 the reference's default example needs with its L1 Gauss-Seidel cycle, two 0.56 ms cycles for nothing)
        double prev = h_sc[S_BETANOM];
        int batch = std::min(check_every, std::max(1, std::min(total, 4)));
        for (int it = 0; it < total;) {
            batch = std::min(batch, total - it);
            for (int q = 0; q < batch; q++) {
                if (graph && cg_lean) { ELAG_HIP_CHECK(hipGraphLaunch(cg_iter_par[cg_par], s)); cg_par ^= 1; }
                else if (graph) ELAG_HIP_CHECK(hipGraphLaunch(cg_iter, s));
                else cg_iteration();
            }
            it += batch;
            if (poll()) break;
            const double cur = h_sc[S_BETANOM], r0v = h_sc[S_R0];
            int next = check_every;
            if (prev > 0.0 && cur > 0.0 && cur < prev && r0v > 0.0 && cur > r0v) {
                const double rate = std::log(prev / cur) / batch;          // decrease of log (Br,r) per iteration
                next = (int)std::ceil(std::log(cur / r0v) / rate);
            } else if (prev == 0.0) {
                next = batch;                                               // no rate yet: same size once more
            }
            prev = cur;
            batch = std::max(1, std::min(check_every, next));
        }
    }

    // which gmres_chain_kernel a step uses: 0 none (a launch per link), 1 / 2 = 6 / 12 elements per thread.  None: the switch is off,
    // the operator is distributed or belongs to a rank thread of a process that runs several (their chains would wait for CUs the others'
    // spinning workgroups hold), a chain of this process has given up before, or the vectors do not fit the registers
    int chain_form(int nn) const
    {
        if (!gm_gran || ctx().comm_kind == 2 || !ctx().h_dev_err || (device_error_codes_seen() & (1 << 2))) return 0;
        const long long per = (long long)kPart * kB;
        for (int form = std::min(std::max(gm_chain, 1), 2); form <= 2; form++)                              // (ELAG_HIP_GMRES_CHAIN = 2: the wider form on a small system too)
            if (nn <= per * (6 << (form - 1))) return form;
        return 0;
    }

    void run_gmres(const double *b, bool iterative, bool flexible)
    {
        hipStream_t s = ctx().stream;
        const int nn = n(), m = restart;
        auto Vq = [&](int q) { return V + (size_t)q * nn; };
        auto Zq = [&](int q) { return Z + (size_t)q * nn; };
        // initial (preconditioned) residual
        if (flexible) {
            if (iterative) residual(b, x, r);
            else { k::vec_fill(nn, 0.0, x); dev_copy(r, b, sizeof(double) * (size_t)nn, s); }
        } else if (has_prec()) {
            if (iterative) { residual(b, x, w); prec(w, r); }
            else { k::vec_fill(nn, 0.0, x); prec(b, r); }
        } else {
            if (iterative) residual(b, x, r);
            else { k::vec_fill(nn, 0.0, x); dev_copy(r, b, sizeof(double) * (size_t)nn, s); }
        }
        dot(r, r);
        hipLaunchKernelGGL(gmres_init_kernel, dim3(1), dim3(1), 0, s, st, rel_tol, abs_tol, max_iter);
        ELAG_LAUNCH_CHECK();
        if (poll()) return;
        // Steps enqueued past convergence change nothing but still run the (unpredicated) preconditioner -- three V-cycles of the Block-LDU graph --,
        // so the host looks at the state as often as the residual estimate's rate says it should: after a batch, the steps still needed are
        // log(estimate / threshold) / (its decrease per step), at most check_every (the SPE10-like system needs 36 steps: polled every 8, four of 40 were for nothing)
        int since_poll = 0, poll_after = std::min(check_every, 4);
        double est_prev = h_sc[S_BETANOM];
        bool finished = false;
        // One inner step (product, preconditioner, modified Gram-Schmidt against i + 1 vectors, rotation) is some 30 (a launch per link: 40 + i)
        // launches that depend on nothing the host knows: replayed as ONE graph per position i of the cycle (built on first use, kept for
        // the solver's life).  Enqueueing them one by one is half of a solve's wall time on a fast host (cube456 x3 Darcy: 6-8 of 12 ms)
        // and all of it on a slow one (22-29 ms seen on some boxes of the pool).  ELAG_HIP_GMRES_GRAPH=0: plain launches.
        auto step = [&](int i) {
            double *target = flexible ? r : w;   // the vector that is orthogonalised
            if (flexible) {
                if (has_prec()) prec(Vq(i), Zq(i));
                else dev_copy(Zq(i), Vq(i), sizeof(double) * (size_t)nn, s);
                amul(Zq(i), r);
            } else if (has_prec()) {
                amul(Vq(i), r);
                prec(r, w);
            } else {
                amul(Vq(i), w);
            }
            bool basis_vector_written = false, rotated = false;
            if (!distributed()) {
                // fused modified Gram-Schmidt: see gmres_orth_fused_kernel
                if (chain_form(nn) != 0) {
                    // the whole chain, the norm and v_{i+1} in one launch (gmres_chain_kernel)
                    ChainArgs a;
                    a.n = nn; a.m = m; a.i = i; a.st = st; a.hs = hs; a.w = target; a.V = V; a.vout = Vq(i + 1); a.gran = gm_gran;
                    a.host_err = ctx().h_dev_err;
                    a.givens = (gm_chain_givens && !gm_update_each) ? 1 : 0;
                    rotated = a.givens != 0;
                    double ms = 250.0;
                    if (const char *e = std::getenv("ELAG_HIP_GMRES_CHAIN_MS")) ms = atof(e);
                    a.ticks = std::max(1ll, (long long)(ms * 1e5));
                    a.stall = std::getenv("ELAG_HIP_GMRES_CHAIN_STALL") ? atoi(std::getenv("ELAG_HIP_GMRES_CHAIN_STALL")) : 0;
                    switch (chain_form(nn)) {
                    case 1: hipLaunchKernelGGL((gmres_chain_kernel<6, true>), dim3(kChainG), dim3(2 * kB + 64), sizeof(double) * 2 * (size_t)m, s, a); break;
                    default: hipLaunchKernelGGL((gmres_chain_kernel<12, false>), dim3(kChainG), dim3(2 * kB + 64), sizeof(double) * 2 * (size_t)m, s, a); break;
                    }
                    basis_vector_written = true;
                } else {
                double *pin = part, *pout = part + kPart;
                hipLaunchKernelGGL(gmres_dot0_kernel, dim3(kPart), dim3(kB), 0, s, nn, (const KState *)st, (const double *)target, (const double *)Vq(0), pin);
                for (int k = 0; k <= i; k++) {
                    hipLaunchKernelGGL(gmres_orth_fused_kernel, dim3(kPart), dim3(kB), 0, s, nn, (const KState *)st, hs, m, k, i, target,
                                       (const double *)Vq(k), (const double *)(k < i ? Vq(k + 1) : nullptr), (const double *)pin, pout);
                    std::swap(pin, pout);
                }
                hipLaunchKernelGGL(gmres_norm_fused_kernel, dim3(1), dim3(kB), 0, s, st, hs, m, i, (const double *)pin);
                }
            } else {
                for (int k = 0; k <= i; k++) {
                    dot(target, Vq(k));
                    hipLaunchKernelGGL(gmres_orth_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, hs, m, k, i, target, (const double *)Vq(k));
                }
                dot(target, target);
                hipLaunchKernelGGL(gmres_norm_kernel, dim3(1), dim3(1), 0, s, st, hs, m, i);
            }
            if (!basis_vector_written)
                hipLaunchKernelGGL(scale_inv_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, (int)S_BETA, (const double *)target, Vq(i + 1));
            if (!rotated) hipLaunchKernelGGL(gmres_givens_kernel, dim3(1), dim3(64), sizeof(double) * (size_t)(4 * m + 6), s, st, hs, m, i);
            if (gm_update_each)
                hipLaunchKernelGGL(gmres_update_kernel, dim3(std::min(g1(nn), 512u)), dim3(kB), 0, s, nn, (const KState *)st, (const double *)hs, m, i + 1,
                                   (const double *)(flexible ? Z : V), x);
            ELAG_LAUNCH_CHECK();
        };
        // the converged step's update, once the host has read `done` (nothing where the solve ended at a restart: gmres_final_update_kernel)
        auto final_update = [&]() {
            if (gm_update_each) return;
            hipLaunchKernelGGL(gmres_final_update_kernel, dim3(std::min(g1(nn), 2048u)), dim3(kB), 0, s, nn, (const KState *)st, (const double *)hs, m,
                               (const double *)(flexible ? Z : V), x);
            ELAG_LAUNCH_CHECK();
        };
        static const bool graph_env = !(std::getenv("ELAG_HIP_GMRES_GRAPH") && !atoi(std::getenv("ELAG_HIP_GMRES_GRAPH")));
        bool graphs = graph_env && gm_graph_ok && capturable();
        // step graphs captured with the one-launch chain are thrown away once a chain of this process has given up: a replay would launch it again
        if (gm_graphs_chain && chain_form(nn) == 0) {
            for (auto g : gm_step) if (g) (void)hipGraphExecDestroy(g);
            gm_step.clear();
            gm_graphs_chain = false;
        }
        if (graphs && (int)gm_step.size() != m) { for (auto g : gm_step) if (g) (void)hipGraphExecDestroy(g); gm_step.assign((size_t)m, nullptr); gm_flex = flexible; }
        if (graphs && gm_flex != flexible) graphs = false;
        for (int j = 1; j <= max_iter && !finished;) {
            hipLaunchKernelGGL(scale_inv_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, (int)S_GBETA, (const double *)r, Vq(0));
            hipLaunchKernelGGL(gmres_start_cycle_kernel, dim3(1), dim3(1), 0, s, st, hs, m, j);
            int i = 0;
            for (; i < m && j <= max_iter; i++, j++) {
                if (graphs && !gm_step[(size_t)i]) {
                    hipGraph_t g = nullptr;
                    bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
                    std::string why;
                    if (ok) {
                        try { step(i); }
                        catch (const std::exception &e) { ok = false; why = e.what(); }
                        catch (...) { ok = false; why = "unknown exception"; }
                        const hipError_t ee = hipStreamEndCapture(s, &g);
                        if (ee != hipSuccess) { ok = false; if (why.empty()) why = std::string("hipStreamEndCapture: ") + hipGetErrorString(ee); }
                    } else why = "hipStreamBeginCapture failed";
                    hipGraphExec_t ge = nullptr;
                    if (ok && g) {
                        const hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                        if (ie != hipSuccess) { ok = false; why = std::string("hipGraphInstantiate: ") + hipGetErrorString(ie); }
                    }
                    if (ok && ge) { gm_step[(size_t)i] = ge; if (chain_form(nn) != 0) gm_graphs_chain = true; }
                    else {
                        // this solver keeps to plain launches; the reason stays with it (elag_hip_krylov_graph_note) and is printed once
                        // under ELAG_HIP_VERBOSE -- a failed capture must not leave a latched runtime error or an error text behind
                        (void)hipGetLastError();
                        graphs = false; gm_graph_ok = false;
                        gm_graph_note = "step " + std::to_string(i) + " was not captured: " + (why.empty() ? "no graph" : why);
                        if (std::getenv("ELAG_HIP_VERBOSE")) fprintf(stderr, "[elag_hip] GMRES: %s; plain launches from here on\n", gm_graph_note.c_str());
                    }
                    if (g) (void)hipGraphDestroy(g);
                }
                if (graphs) ELAG_HIP_CHECK(hipGraphLaunch(gm_step[(size_t)i], s));
                else step(i);
                if (++since_poll >= poll_after) {
                    const int steps = since_poll;
                    since_poll = 0;
                    if (poll()) { final_update(); finished = true; i++; j++; break; }
                    const double est = h_sc[S_BETANOM], thr = h_sc[S_R0];
                    poll_after = check_every;
                    if (est_prev > 0.0 && est > 0.0 && est < est_prev && thr > 0.0 && est > thr) {
                        const double rate = std::log(est_prev / est) / steps;
                        poll_after = std::max(1, std::min(check_every, (int)std::ceil(std::log(est / thr) / rate)));
                    }
                    est_prev = est;
                }
            }
            launch_mode = graphs ? "graph" : gm_graph_note.empty() ? "plain" : "plain: " + gm_graph_note;
            if (finished) break;
            // restart: Update(x, i-1, ...), true residual, convergence test on its norm
            hipLaunchKernelGGL(gmres_restart_backsolve_kernel, dim3(1), dim3(1), 0, s, st, hs, m, i - 1);
            hipLaunchKernelGGL(gmres_update_kernel, dim3(std::min(g1(nn), 512u)), dim3(kB), 0, s, nn, (const KState *)st, (const double *)hs, m, i,
                               (const double *)(flexible ? Z : V), x);
            if (!flexible && has_prec()) { residual(b, x, w); prec(w, r); }
            else residual(b, x, r);
            dot(r, r);
            hipLaunchKernelGGL(gmres_restart_norm_kernel, dim3(1), dim3(1), 0, s, st, j);
            ELAG_LAUNCH_CHECK();
            if (poll()) { final_update(); break; }      // (a step of this cycle may have converged since the last look)
        }
    }

    // mfem::BiCGSTABSolver::Mult: r, rtilde = z, p = e[0], phat = e[1], v = e[2], s = e[3], shat = e[4], t = e[5]
    void run_bicgstab(const double *b, bool iterative)
    {
        hipStream_t s = ctx().stream;
        const int nn = n();
        const dim3 g(g1(nn)), blk(kB);
        double *rt = z, *p = e[0], *phat = e[1], *v = e[2], *sv = e[3], *shat = e[4], *t = e[5];
        if (iterative) residual(b, x, r);
        else { k::vec_fill(nn, 0.0, x); dev_copy(r, b, sizeof(double) * (size_t)nn, s); }
        dev_copy(rt, r, sizeof(double) * (size_t)nn, s);
        dot(r, r);
        hipLaunchKernelGGL(bicg_init_kernel, dim3(1), dim3(1), 0, s, st, rel_tol, abs_tol, max_iter);
        ELAG_LAUNCH_CHECK();
        if (poll()) return;
        int since_poll = 0;
        for (int i = 1; i <= max_iter; i++) {
            dot(rt, r);
            hipLaunchKernelGGL(bicg_rho_kernel, dim3(1), dim3(1), 0, s, st, i);
            if (i == 1) hipLaunchKernelGGL(copy_pred_kernel, g, blk, 0, s, nn, (const KState *)st, (const double *)r, p);
            else hipLaunchKernelGGL(bicg_p_kernel, g, blk, 0, s, nn, (const KState *)st, (const double *)r, (const double *)v, p);
            if (has_prec()) prec(p, phat);
            else hipLaunchKernelGGL(copy_pred_kernel, g, blk, 0, s, nn, (const KState *)st, (const double *)p, phat);
            amul(phat, v);
            dot(rt, v);
            hipLaunchKernelGGL(bicg_alpha_kernel, dim3(1), dim3(1), 0, s, st);
            hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)r, (int)S_ALPHA, -1.0, (const double *)v, sv);   // s = r - alpha v
            dot(sv, sv);
            hipLaunchKernelGGL(bicg_mid_kernel, dim3(1), dim3(1), 0, s, st, i, i);
            hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, i, (const double *)x, (int)S_ALPHA, 1.0, (const double *)phat, x);   // x += alpha phat
            if (has_prec()) prec(sv, shat);
            else hipLaunchKernelGGL(copy_pred_kernel, g, blk, 0, s, nn, (const KState *)st, (const double *)sv, shat);
            amul(shat, t);
            dot(t, sv);
            hipLaunchKernelGGL(bicg_ts_kernel, dim3(1), dim3(1), 0, s, st);
            dot(t, t);
            hipLaunchKernelGGL(bicg_omega_kernel, dim3(1), dim3(1), 0, s, st);
            hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)x, (int)S_OMEGA, 1.0, (const double *)shat, x);   // x += omega shat
            hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)sv, (int)S_OMEGA, -1.0, (const double *)t, r);   // r = s - omega t
            dot(r, r);
            hipLaunchKernelGGL(bicg_end_kernel, dim3(1), dim3(1), 0, s, st, i);
            ELAG_LAUNCH_CHECK();
            if (++since_poll >= check_every) { since_poll = 0; if (poll()) break; }
        }
    }

    // mfem::MINRESSolver::Mult: v0 = e[0], v1 = e[1], w0 = e[2], w1 = e[3], q = e[4], u1 = e[5]
    void run_minres(const double *b, bool iterative)
    {
        hipStream_t s = ctx().stream;
        const int nn = n();
        const dim3 g(g1(nn)), blk(kB);
        double *v0 = e[0], *v1 = e[1], *w0 = e[2], *w1 = e[3], *q = e[4], *u1 = e[5];
        if (iterative) residual(b, x, v1);
        else { dev_copy(v1, b, sizeof(double) * (size_t)nn, s); k::vec_fill(nn, 0.0, x); }
        if (has_prec()) prec(v1, u1);
        dot(has_prec() ? u1 : v1, v1);
        hipLaunchKernelGGL(minres_init_kernel, dim3(1), dim3(1), 0, s, st, rel_tol, abs_tol, max_iter);
        ELAG_LAUNCH_CHECK();
        if (poll()) return;
        int since_poll = 0;
        for (int it = 1; it <= max_iter; it++) {
            double *zz = has_prec() ? u1 : v1;
            hipLaunchKernelGGL(scale_recip_kernel, g, blk, 0, s, nn, (const KState *)st, (int)S_BETA, v1);
            if (has_prec()) hipLaunchKernelGGL(scale_recip_kernel, g, blk, 0, s, nn, (const KState *)st, (int)S_BETA, u1);
            amul(zz, q);
            dot(zz, q);
            hipLaunchKernelGGL(minres_alpha_kernel, dim3(1), dim3(1), 0, s, st);
            if (it > 1) hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)q, (int)S_BETA, -1.0, (const double *)v0, q);   // q -= beta v0
            hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)q, (int)S_ALPHA, -1.0, (const double *)v1, v0);         // v0 = q - alpha v1
            if (!has_prec()) dot(v0, v0);
            else { prec(v0, q); dot(v0, q); }
            hipLaunchKernelGGL(minres_beta_kernel, dim3(1), dim3(1), 0, s, st);
            if (it == 1)
                hipLaunchKernelGGL(lin2_kernel, g, blk, 0, s, nn, (const KState *)st, 0, -1, 0.0, (const double *)nullptr, (int)S_C0, 1.0, (const double *)zz, w0);
            else if (it == 2)
                hipLaunchKernelGGL(lin2_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (int)S_C0, 1.0, (const double *)zz, (int)S_C1, 1.0, (const double *)w1, w0);
            else {
                hipLaunchKernelGGL(lin2_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (int)S_C2, 1.0, (const double *)w0, (int)S_C1, 1.0, (const double *)w1, w0);
                hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)w0, (int)S_C0, 1.0, (const double *)zz, w0);
            }
            hipLaunchKernelGGL(axpy_kernel, g, blk, 0, s, nn, (const KState *)st, 0, (const double *)x, (int)S_TS, 1.0, (const double *)w0, x);              // x += gamma1 eta w0
            hipLaunchKernelGGL(minres_eta_kernel, dim3(1), dim3(1), 0, s, st, it);
            ELAG_LAUNCH_CHECK();
            if (has_prec()) std::swap(u1, q);
            std::swap(v0, v1);
            std::swap(w0, w1);
            if (++since_poll >= check_every) { since_poll = 0; if (poll()) break; }
        }
    }

    void run_stationary(const double *b, bool iterative)
    {
        hipStream_t s = ctx().stream;
        const int nn = n();
        if (iterative) residual(b, x, r);
        else { dev_copy(r, b, sizeof(double) * (size_t)nn, s); k::vec_fill(nn, 0.0, x); }
        dot(r, r);
        hipLaunchKernelGGL(stat_init_kernel, dim3(1), dim3(1), 0, s, st, max_iter);
        for (int it = 0; it < max_iter;) {
            const int batch = std::min(check_every, max_iter - it);
            for (int q = 0; q < batch; q++) {
                prec(r, d);                                                                                       // Correction = Solver(Residual)
                hipLaunchKernelGGL(add_pred_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, x, (const double *)d);   // sol += Correction
                amul(d, z);                                                                                       // Tmp = A Correction
                hipLaunchKernelGGL(sub_pred_kernel, dim3(g1(nn)), dim3(kB), 0, s, nn, (const KState *)st, r, (const double *)z);   // Residual -= Tmp
                dot(r, r);
                hipLaunchKernelGGL(stat_step_kernel, dim3(1), dim3(1), 0, s, st, rel_tol, abs_tol);
            }
            ELAG_LAUNCH_CHECK();
            it += batch;
            if (poll()) break;
        }
    }

    void mult(const double *b, double *xio, bool iterative)
    {
        hipStream_t s = ctx().stream;
        const size_t nn = (size_t)n();
        if (d_perm) {
            // into the hierarchy's numbering: v_perm[new] = v[order[new]]
            k::vec_gather((int)nn, d_perm, b, bperm);
            if (iterative) k::vec_gather((int)nn, d_perm, xio, x);
            b = bperm;
        } else if (iterative) {
            dev_copy(x, xio, sizeof(double) * nn, s);
        }
        static const bool timing = std::getenv("ELAG_HIP_KRYLOV_TIMING") != nullptr;
        const auto t_begin = std::chrono::steady_clock::now();
        t_poll = 0.0; n_poll = 0;
        switch (solver) {
        case ELAG_HIP_KRYLOV_CG: run_cg(b, iterative); break;
        case ELAG_HIP_KRYLOV_GMRES: run_gmres(b, iterative, false); break;
        case ELAG_HIP_KRYLOV_FGMRES: run_gmres(b, iterative, true); break;
        case ELAG_HIP_KRYLOV_BICGSTAB: run_bicgstab(b, iterative); break;
        case ELAG_HIP_KRYLOV_MINRES: run_minres(b, iterative); break;
        default: run_stationary(b, iterative); break;
        }
        const double t_run = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        fetch_results();
        if (timing)
            fprintf(stderr, "[elag_hip] krylov solve: %d iterations, host %.3f ms up to the last poll: %.3f ms enqueueing, %.3f ms waiting in %d polls\n", final_iter,
                    t_run * 1e3, (t_run - t_poll) * 1e3, t_poll * 1e3, n_poll);
        if (d_perm) k::vec_scatter((int)nn, d_perm, x, xio, 0);
        else dev_copy(xio, x, sizeof(double) * nn, s);
        if (timing) {
            ELAG_HIP_CHECK(hipStreamSynchronize(s));
            fprintf(stderr, "[elag_hip] krylov mult: %.3f ms in all\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3);
        }
    }
};

extern "C" {

int elag_hip_krylov_default_options(elag_hip_krylov_options *opt)
{
    return guarded([&] {
        ELAG_REQUIRE(opt, "null options");
        // the defaults KrylovSolver falls back to (KrylovSolver.cpp:86-94) and "Restart size" 50 (:47-49)
        opt->solver = ELAG_HIP_KRYLOV_CG; opt->max_iter = 10; opt->rel_tol = 0.0; opt->abs_tol = 0.0; opt->restart = 50;
        opt->check_every = 8; opt->use_graph = 1;
    });
}

int elag_hip_krylov_create(elag_hip_krylov_t *out, elag_hip_parcsr_t A, int prec_kind, void *prec, const elag_hip_krylov_options *opt_in)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && A, "null argument");
        ELAG_REQUIRE(A->nrows() == A->ncols(), "Krylov solvers need a square operator");
        elag_hip_krylov_options opt;
        if (opt_in) opt = *opt_in; else { int rc = elag_hip_krylov_default_options(&opt); if (rc) fail(rc, "%s", elag_hip_last_error()); }
        ELAG_REQUIRE(opt.solver >= ELAG_HIP_KRYLOV_CG && opt.solver <= ELAG_HIP_KRYLOV_MINRES,
                     "KrylovSolver::KrylovSolver(...): Bad solver type (%d). Device choices are CG, GMRES, FGMRES and the stationary iteration",
                     opt.solver);
        ELAG_REQUIRE(opt.max_iter >= 0, "\"Maximum iterations\" must be >= 0");
        ELAG_REQUIRE(prec_kind >= ELAG_HIP_PREC_NONE && prec_kind <= ELAG_HIP_PREC_COARSE, "unknown preconditioner kind %d", prec_kind);
        ELAG_REQUIRE(prec_kind == ELAG_HIP_PREC_NONE || prec, "null preconditioner");
        ELAG_REQUIRE(opt.solver != ELAG_HIP_KRYLOV_STATIONARY || prec_kind != ELAG_HIP_PREC_NONE, "the stationary iteration needs a solver to apply");
        std::unique_ptr<elag_hip_krylov_s> K(new elag_hip_krylov_s());
        K->A = A; K->solver = opt.solver; K->max_iter = opt.max_iter; K->rel_tol = opt.rel_tol; K->abs_tol = opt.abs_tol;
        K->restart = opt.restart; K->check_every = std::max(1, opt.check_every); K->use_graph = opt.use_graph != 0;
        K->prec_kind = prec_kind;
        if (prec_kind == ELAG_HIP_PREC_SMOOTHER) {
            K->psm = static_cast<elag_hip_smoother_s *>(prec);
            ELAG_REQUIRE(K->psm->n() == A->nrows(), "preconditioner size %d does not match the operator (%d)", K->psm->n(), A->nrows());
        } else if (prec_kind == ELAG_HIP_PREC_HIERARCHY) {
            K->phier = static_cast<elag_hip_hierarchy_s *>(prec);
            ELAG_REQUIRE(hierarchy_height(K->phier) == A->nrows(), "hierarchy size %d does not match the operator (%d)", hierarchy_height(K->phier), A->nrows());
        } else if (prec_kind == ELAG_HIP_PREC_COARSE) {
            K->pcoarse = static_cast<elag_hip_coarse_s *>(prec);
            ELAG_REQUIRE((K->pcoarse->replicated ? K->pcoarse->nloc : K->pcoarse->n) == A->nrows(), "coarse solver size does not match the operator");
        }
        K->setup();
        *out = K.release();
    });
}

int elag_hip_krylov_mult(elag_hip_krylov_t K, const double *b, double *x, int iterative_mode, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(K && b && x, "null argument");
        ELAG_REQUIRE(b != x, "rhs and sol must not alias");
        check_loc(loc);
        const size_t n = (size_t)K->n();
        VecIn bb(b, n, loc);
        VecInOut xx(x, n, loc, iterative_mode != 0);
        K->mult(bb.d, xx.d, iterative_mode != 0);
        xx.finish();
        ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    });
}

const char *elag_hip_krylov_launch_mode(elag_hip_krylov_t K)
{
    if (!K) return "";
    return K->launch_mode.c_str();
}

int elag_hip_krylov_get_stats(elag_hip_krylov_t K, int *converged, int *final_iter, double *final_norm)
{
    return guarded([&] {
        ELAG_REQUIRE(K, "null solver");
        if (converged) *converged = K->converged;
        if (final_iter) *final_iter = K->final_iter;
        if (final_norm) *final_norm = K->final_norm;
    });
}

int elag_hip_krylov_destroy(elag_hip_krylov_t K)
{
    return guarded([&] {
        if (K) { (void)hipStreamSynchronize(ctx().stream); delete K; }
    });
}

}  // extern "C"
