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
// Two ways to run the levels.  (1) The dataflow sweep (round 5, sweep_dataflow_kernel below; the default where it applies): ONE launch per symmetric
// sweep in which every row waits for exactly the rows it reads -- a hand-off between workgroups (~1.1-1.5 us) per dependency level instead of a launch;
// for large operators the levels and the schedule themselves are computed by kernels of the same kind (setup_device).  (2) The launch plan of rounds
// 1-4, kept for Kaczmarz, hypre's weighted branches, virtual ranks with a chain per block, several rank threads on one device, as the reference the
// dataflow sweep is checked against, and behind ELAG_HIP_SWEEP_DATAFLOW=0:
// Launch plan ("segments"): a level with more than kChainRows rows is one multi-workgroup launch; a run of consecutive smaller
// levels is ONE launch of a single workgroup that walks the levels with a workgroup barrier in between (the waves of a
// workgroup share their CU's L1, so values written before the barrier are read after it without leaving the CU).  All
// launches sit in the V-cycle's hipGraph like any other kernel of the cycle.  A sweep on a small operator is a latency chain
// (one level = one dependent round trip through memory), so the kernel is built to make that trip short: a 16-byte schedule
// slot per row (row, first entry, length), kLPR lanes fetching a row's entries and their u values together, kUnroll rows in
// flight per lane group; only the subtraction itself is sequential, done from shuffled products in storage order.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <numeric>

#include "comm.hpp"
#include "smoother.hpp"

namespace elag {

namespace {

constexpr int kLPR = 16;           // lanes that fetch the entries of one row together (the row sum itself stays sequential)
constexpr int kUnroll = 4;         // rows a lane group has in flight at once
constexpr int kChainBlock = 512;   // the chain workgroup: 32 lane groups (1024 threads would halve the steps but spill: 128 VGPRs per lane there)
constexpr int kStepRows = (kChainBlock / kLPR) * kUnroll;    // rows of one chain step: 128 (a level is cut into steps of that many rows)
constexpr int kChainRows = kStepRows;                        // levels up to this many rows (one step) are walked by the chain workgroup: measured, see setup()
constexpr int kEPL = 2;            // entries per lane the chain keeps in flight for a row (kLPR * kEPL = 32 entries without a further fetch)
constexpr int kWideBlock = 256;
constexpr int kMaxChain = 4096;    // steps per chain launch (their offsets sit in LDS)
constexpr int kLdsRows = 16384;    // operators up to this many rows keep u in the chain workgroup's LDS (128 KB of the CU's 160)

typedef int int4s __attribute__((ext_vector_type(4)));

// One row, kLPR lanes: the lanes fetch a_ik, u_k of up to kLPR entries at once; the subtraction is then done in storage order
// (every lane does it, from shuffled products), so the value is the sequential loop's.
// MODE 0 (types 2 / 4):  u_i += (f_i - sum_k a_ik u_k - sum_g o_ig ughost_g) / l1_i
// MODE 1 (type 6):       u_i  = (f_i - sum_{k != i} a_ik u_k - sum_g o_ig ughost_g) / a_ii      rows with a_ii == 0 (or none) skipped
// MODE 2 (type 3):       res = (f_i - sum_k a_ik u_k - sum_g o_ig ughost_g) / |a_i|^2;  u_k += omega res a_ik for every local column k of row i
//                        (Kaczmarz: the rows of a level share no column, so the lanes' updates never meet)
// One chunk of up to kLPR entries of a row (lane q holds entry q: column j, value v, gathered x; lanes past the row's end hold
// j = -1, v = x = 0): the subtraction runs over all kLPR slots in storage order, fully unrolled so that the kLPR lane
// broadcasts are issued back to back instead of one per loop trip (a slot past the end subtracts +0.0, which leaves every
// double as it is).
// value of lane Q of the caller's 16-lane row in every lane of that row: a DPP move (row_newbcast), no LDS traffic.  (The
// generic __shfl goes through ds_bpermute: 32 LDS operations per chunk of a row, and the chain workgroup's eight waves share one
// LDS pipe -- measured 1.9 us per step, all of it there.)
template <int Q>
__device__ __forceinline__ double row_bcast(double v)
{
    static_assert(kLPR == 16, "a DPP row is 16 lanes");
    // (the 64-bit form: row_newbcast is the one DPP control the double-precision ALU takes -- one v_mov_b64_dpp where the halves took two moves:
    //  cube456 x3 level 0 260 -> 251 us per symmetric dataflow sweep, 589 -> 530 as a launch per level.  Stopping the subtraction behind the last
    //  slot any row of the wavefront uses -- a second chunk holds 1-5 entries -- was measured too: tested per slot 267 us, as 4- / 8- / 16-slot
    //  forms behind one branch 260: the tridiagonal chain gains 5 %, the operators lose what the extra code costs)
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + Q, 0xf, 0xf, false);
}
template <int Q>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + Q, 0xf, 0xf, false); }

template <int MODE, int Q>
struct OrderedSteps {
    static __device__ __forceinline__ void run(int row, int j, double v, double p, double &res, double &aii, bool &have)
    {
        const double pq = row_bcast<Q>(p);
        if (MODE == 1) {
            const int jq = row_bcast<Q>(j);
            const double vq = row_bcast<Q>(v);
            const bool diag = (jq == row) && !have;
            if (diag) { aii = vq; have = true; }
            res -= diag ? 0.0 : pq;
        } else {
            res -= pq;
        }
        OrderedSteps<MODE, Q + 1>::run(row, j, v, p, res, aii, have);
    }
};
template <int MODE>
struct OrderedSteps<MODE, kLPR> {
    static __device__ __forceinline__ void run(int, int, double, double, double &, double &, bool &) {}
};

// One chunk of up to kLPR entries of a row (lane q holds entry q: column j, value v, gathered x; lanes past the row's end hold
// j = -1, v = x = 0): the subtraction runs over all kLPR slots in storage order (a slot past the end subtracts +0.0, which
// leaves every double as it is).
template <int MODE>
__device__ __forceinline__ void ordered_chunk(int row, int j, double v, double x, double &res, double &aii, bool &have)
{
    OrderedSteps<MODE, 0>::run(row, j, v, v * x, res, aii, have);
}

// the offd terms of a row (frozen ghost values), same order
__device__ __forceinline__ void ordered_plain(double p, double &res)
{
    double aii = 0.0;
    bool have = false;
    OrderedSteps<0, 0>::run(0, 0, 0.0, p, res, aii, have);
}

// One big level as a multi-workgroup launch: slots [b, e) of the schedule, a lane group each.  A launch is a chain of dependent memory
// round trips, so what can be addressed from the slot number alone travels in the FIRST one: the slot itself and the first kLPR
// entries of its row, kept in schedule order (sJ / sa, packed at setup; a row's further entries come from the CSR arrays).  The second
// trip gathers u and the row's f, d, u; then the ordered subtraction and the store.
template <int MODE>
__global__ __launch_bounds__(kWideBlock) void sweep_levels_kernel(int b, int e, const int4s *__restrict__ sched, const int *__restrict__ sJ,
                                                                  const double *__restrict__ sa,
                                                                  const int *__restrict__ J, const double *__restrict__ a,
                                                                  const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa,
                                                                  const double *__restrict__ f, const double *__restrict__ d, const double *__restrict__ ug,
                                                                  double omega, double *u)
{
    const int lane = threadIdx.x & (kLPR - 1);
    const int t = b + (int)(blockIdx.x * blockDim.x + threadIdx.x) / kLPR;
    if (t >= e) return;
    const int4s sl = sched[t];
    const int j0 = sJ[(size_t)t * kLPR + lane];                 // -1 past the row's end
    const double v0 = sa[(size_t)t * kLPR + lane];              // 0 there
    const int row = sl.x, s = sl.y, len = sl.z;
    double res = f[row], aii = 0.0;
    const double dr = (MODE != 1) ? d[row] : 0.0, uo = (MODE == 0) ? u[row] : 0.0;
    bool have = false;
    ordered_chunk<MODE == 1 ? 1 : 0>(row, j0, v0, j0 >= 0 ? u[j0] : 0.0, res, aii, have);
    for (int k0 = kLPR; k0 < len; k0 += kLPR) {
        const bool has = k0 + lane < len;
        const int j = has ? J[s + k0 + lane] : -1;
        const double v = has ? a[s + k0 + lane] : 0.0;
        const double x = has ? u[j] : 0.0;
        ordered_chunk<MODE == 1 ? 1 : 0>(row, j, v, x, res, aii, have);
    }
    if (oI) {
        const int os = oI[row], ol = oI[row + 1] - os;
        for (int k0 = 0; k0 < ol; k0 += kLPR) {
            const bool has = k0 + lane < ol;
            const double v = has ? oa[os + k0 + lane] : 0.0;
            const double x = has ? ug[oJ[os + k0 + lane]] : 0.0;
            ordered_plain(v * x, res);
        }
    }
    if (MODE == 2) {
        res /= dr;
        if (j0 >= 0) u[j0] += omega * res * v0;
        for (int k0 = kLPR; k0 < len; k0 += kLPR)
            if (k0 + lane < len) u[J[s + k0 + lane]] += omega * res * a[s + k0 + lane];
        return;
    }
    if (lane == 0) {
        if (MODE == 0) { if (dr != 0.0) u[row] = uo + omega * res / dr; }
        else if (have && aii != 0.0) u[row] = res / aii;
    }
}

// The chain: ONE workgroup walks a run of consecutive small levels (cut into steps of <= kStepRows rows; a barrier between levels only).  What a step needs that
// does not depend on u -- its schedule slots, then the column indices and values of its rows -- is requested one and two steps
// ahead, so that the only memory round trip left on the critical path of a step is the gather of u itself (written by the
// previous step of this very workgroup, hence close by), followed by the ordered subtraction, the store and the barrier.
// LDSU (round 3, an experiment kept off by default -- see setup()): operators of up to kLdsRows rows keep the WHOLE vector u in the
// workgroup's LDS for the length of the launch, every level of the sweep joins the chain, and with `sym` the backward sweep follows the
// forward one in the same launch.  Updates are written to LDS and to memory, so the result is where it always was; the arithmetic -- and
// every bit of the result -- is the other path's (tests/test_gpu_sweeps.py runs one case with ELAG_HIP_SWEEP_LDS=1).
template <int MODE, bool LDSU>
__global__ __launch_bounds__(kChainBlock) void sweep_chain_kernel(const int *__restrict__ step_ptr, const int *__restrict__ step_first, int s_first, int s_count, int dir0,
                                                                  const int4s *__restrict__ sched, const int *__restrict__ J, const double *__restrict__ a,
                                                                  const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa,
                                                                  const double *__restrict__ f, const double *__restrict__ d, const double *__restrict__ ug,
                                                                  double omega, double *u, int n, int sym, const int *__restrict__ blk_first,
                                                                  const int *__restrict__ blk_count)
{
    __shared__ int lp[kMaxChain + 1];
    extern __shared__ double ul[];          // LDSU: u[0 .. n)
    // virtual ranks (SweepSmoother::blocks > 1): a workgroup per block, each walking the steps of ITS rows -- the blocks share nothing during a
    // sweep (the diagonal part of the split operator is block diagonal, the other blocks' values are the frozen ghosts)
    if (blk_first) { s_first = blk_first[blockIdx.x]; s_count = blk_count[blockIdx.x]; }
    if (s_count <= 0) return;
    const int s_lo = dir0 > 0 ? s_first : s_first - (s_count - 1);
    // (top bit: this step is the first one of its dependency level)
    for (int c = threadIdx.x; c <= s_count; c += kChainBlock) lp[c] = step_ptr[s_lo + c] | (step_first[s_lo + c] ? (int)0x80000000 : 0);
    if (LDSU)
        for (int i = threadIdx.x; i < n; i += kChainBlock) ul[i] = u[i];
    __syncthreads();
    constexpr int NG = kChainBlock / kLPR;
    const int lane = threadIdx.x & (kLPR - 1);
    const int grp = threadIdx.x / kLPR;
    auto U = [&](int j) -> double { return LDSU ? ul[j] : u[j]; };
    for (int pass = 0; pass < (sym ? 2 : 1); pass++) {
    const int dir = pass == 0 ? dir0 : -dir0;
    if (pass) __syncthreads();
    auto load_slots = [&](int c, int4s (&sl)[kUnroll]) {
        const int li = dir > 0 ? c : s_count - 1 - c;
        const int b = lp[li] & 0x7fffffff, e = lp[li + 1] & 0x7fffffff;
#pragma unroll
        for (int q = 0; q < kUnroll; q++) {
            const int t = b + grp + q * NG;
            sl[q] = t < e ? sched[t] : int4s{0, 0, 0, -1};          // .w = -1: no row here
        }
    };
    auto load_entries = [&](const int4s (&sl)[kUnroll], int (&jj)[kUnroll][kEPL], double (&vv)[kUnroll][kEPL]) {
#pragma unroll
        for (int q = 0; q < kUnroll; q++)
#pragma unroll
            for (int e = 0; e < kEPL; e++) {
                const bool has = e * kLPR + lane < sl[q].z;
                jj[q][e] = has ? J[sl[q].y + e * kLPR + lane] : -1;
                vv[q][e] = has ? a[sl[q].y + e * kLPR + lane] : 0.0;
            }
    };
    int4s slA[kUnroll], slB[kUnroll], slC[kUnroll];
    int jA[kUnroll][kEPL], jB[kUnroll][kEPL];
    double vA[kUnroll][kEPL], vB[kUnroll][kEPL];
#pragma unroll
    for (int q = 0; q < kUnroll; q++) {
        slB[q] = int4s{0, 0, 0, -1}; slC[q] = int4s{0, 0, 0, -1};
#pragma unroll
        for (int e = 0; e < kEPL; e++) { jB[q][e] = 0; vB[q][e] = 0.0; }
    }
    load_slots(0, slA);
    if (s_count > 1) load_slots(1, slB);
    load_entries(slA, jA, vA);
    for (int c = 0; c < s_count; c++) {
        // ---- what depends on u: this step's gathers
        double xA[kUnroll][kEPL], fr[kUnroll], dr[kUnroll], uo[kUnroll];
#pragma unroll
        for (int q = 0; q < kUnroll; q++) {
#pragma unroll
            for (int e = 0; e < kEPL; e++) xA[q][e] = (e * kLPR + lane < slA[q].z) ? U(jA[q][e]) : 0.0;
            fr[q] = f[slA[q].x];
            dr[q] = (MODE != 1) ? d[slA[q].x] : 0.0;
            uo[q] = (MODE == 0) ? U(slA[q].x) : 0.0;
        }
        // ---- what does not: the next step's entries (its slots are here already) and the slots of the step after it
        if (c + 1 < s_count) load_entries(slB, jB, vB);
        if (c + 2 < s_count) load_slots(c + 2, slC);
        // ---- the rows of this step
#pragma unroll
        for (int q = 0; q < kUnroll; q++) {
            if (slA[q].w < 0) continue;
            const int row = slA[q].x, s0 = slA[q].y, len = slA[q].z;
            double res = fr[q], aii = 0.0;
            bool have = false;
#pragma unroll
            for (int e = 0; e < kEPL; e++)
                if (e * kLPR < len) ordered_chunk<MODE == 1 ? 1 : 0>(row, jA[q][e], vA[q][e], xA[q][e], res, aii, have);
            for (int k0 = kEPL * kLPR; k0 < len; k0 += kLPR) {      // rows longer than kLPR * kEPL entries: further chunks, fetched now
                const bool has = k0 + lane < len;
                const int j = has ? J[s0 + k0 + lane] : -1;
                const double v = has ? a[s0 + k0 + lane] : 0.0;
                const double x = has ? U(j) : 0.0;
                ordered_chunk<MODE == 1 ? 1 : 0>(row, j, v, x, res, aii, have);
            }
            if (oI) {
                const int os = oI[row], ol = oI[row + 1] - os;
                for (int k0 = 0; k0 < ol; k0 += kLPR) {
                    const bool has = k0 + lane < ol;
                    const double v = has ? oa[os + k0 + lane] : 0.0;
                    const double x = has ? ug[oJ[os + k0 + lane]] : 0.0;
                    ordered_plain(v * x, res);
                }
            }
            auto put = [&](int j, double val) { u[j] = val; if (LDSU) ul[j] = val; };
            if (MODE == 2) {
                res /= dr[q];
#pragma unroll
                for (int e = 0; e < kEPL; e++)
                    if (e * kLPR + lane < len) put(jA[q][e], U(jA[q][e]) + omega * res * vA[q][e]);
                for (int k0 = kEPL * kLPR; k0 < len; k0 += kLPR)
                    if (k0 + lane < len) { const int j = J[s0 + k0 + lane]; put(j, U(j) + omega * res * a[s0 + k0 + lane]); }
            } else if (lane == 0) {
                if (MODE == 0) { if (dr[q] != 0.0) put(row, uo[q] + omega * res / dr[q]); }
                else if (have && aii != 0.0) put(row, res / aii);
            }
        }
        // the steps of ONE level are independent of each other (that is what a level is): the workgroup only has to meet where the
        // next step belongs to another level -- within a level the waves run on, and the steps' memory round trips overlap
        {
            const int li = dir > 0 ? c : s_count - 1 - c;
            if (dir > 0 ? (lp[li + 1] < 0) : (lp[li] < 0)) __syncthreads();
        }
        // ---- rotate the pipeline
#pragma unroll
        for (int q = 0; q < kUnroll; q++) {
            slA[q] = slB[q]; slB[q] = slC[q];
#pragma unroll
            for (int e = 0; e < kEPL; e++) { jA[q][e] = jB[q][e]; vA[q][e] = vB[q][e]; }
        }
    }
    }   // pass
}

// the first kLPR entries of every row in schedule order (sweep_levels_kernel's first memory round trip)
__global__ void sched_pack_kernel(int n, const int4s *__restrict__ sched, const int *__restrict__ J, const double *__restrict__ a, int *__restrict__ sJ,
                                  double *__restrict__ sa)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)n * kLPR; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / kLPR), q = (int)(i % kLPR);
        const int4s sl = sched[t];
        const bool has = q < sl.z;
        sJ[i] = has ? J[sl.y + q] : -1;
        sa[i] = has ? a[sl.y + q] : 0.0;
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
        if (option == 3) {                                       // squared 2-norm of the row (Kaczmarz)
            r = 0.0;
            for (int k = I[i]; k < I[i + 1]; k++) r += a[k] * a[k];
            if (oI) for (int k = oI[i]; k < oI[i + 1]; k++) r += oa[k] * oa[k];
            // (the sign flip: hypre tests the first stored entry, which is the diagonal one there; here the diagonal entry found above)
        } else if (option == 2) {
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

// hypre's weighted / over-relaxed branches of the same relaxations (the reference forwards "Damping Factor" and "Omega" unconditionally,
// src/linalg/solver_ops/ParELAG_HypreSmootherWrapper.cpp:28-33; no shipped list selects them), one dependency level per launch, a thread per
// row, the sums in storage order (the CPU loop's bits).  v = the vector as it was before the forward pass (both passes read it).
//   MODE 0, types 2 / 4 with relax_weight != 1 ("scaled SSOR", hypre_ParCSRRelax):  c1 = omega w, c2 = omega (1 - w),
//           dif = sum over the diag entries with j < i (forward; j > i backward) of a_ij (v_j - u_j),  u_i += (c1 res + c2 dif) / l1_i
//   MODE 1, type 6 with relax_weight != 1 or omega != 1 (hypre_BoomerAMGRelax):  res0 = -sum_{j != i} a_ij u_j,  res2 = sum_{j != i} a_ij v_j,
//           res = f_i - offd terms,  u_i = (1 - w omega) u_i + w (omega res + res0 + (1 - omega) res2) / a_ii
template <int MODE>
__global__ void sweep_sor_level_kernel(int r0, int r1, int dir, const int *__restrict__ rows, const int *__restrict__ I, const int *__restrict__ J,
                                       const double *__restrict__ a, const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa,
                                       const double *__restrict__ f, const double *__restrict__ d, const double *__restrict__ ug, double weight, double omega,
                                       const double *__restrict__ v, double *u)
{
    const int ri = r0 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (ri >= r1) return;
    const int row = rows[ri];
    double res = f[row];
    if (MODE == 0) {
        const double dr = d[row];
        if (dr == 0.0) return;
        double dif = 0.0;
        for (int k = I[row]; k < I[row + 1]; k++) {
            const int j = J[k];
            const double uj = u[j];
            res -= a[k] * uj;
            if (dir > 0 ? j < row : j > row) dif += a[k] * (v[j] - uj);
        }
        if (oI)
            for (int k = oI[row]; k < oI[row + 1]; k++) res -= oa[k] * ug[oJ[k]];
        u[row] += (omega * weight * res + omega * (1.0 - weight) * dif) / dr;
    } else {
        double res0 = 0.0, res2 = 0.0, aii = 0.0;
        bool have = false;
        for (int k = I[row]; k < I[row + 1]; k++) {
            const int j = J[k];
            if (j == row && !have) { aii = a[k]; have = true; continue; }
            res0 -= a[k] * u[j];
            res2 += a[k] * v[j];
        }
        if (!have || aii == 0.0) return;
        if (oI)
            for (int k = oI[row]; k < oI[row + 1]; k++) res -= oa[k] * ug[oJ[k]];
        double ui = u[row] * (1.0 - weight * omega);
        ui += weight * (omega * res + res0 + (1.0 - omega) * res2) / aii;
        u[row] = ui;
    }
}

// ---- the dataflow sweep -------------------------------------------------------------------------------------------------------------------
// A launch per dependency level costs ~3 us and a level of the cube456 x3 operator is ~500 independent rows: 89 levels x 4 half-sweeps are
// 1.0 of the 1.08 ms of the reference's shipped cycle, nearly all of it launch boundaries.  Here ONE launch runs a whole symmetric sweep
// (forward, then backward) and a row waits for exactly the rows it reads -- the hand-off between workgroups that
// MI355X_MICROARCH.md prices at 0.8-1.0 us (its "handoff-1to1" row) instead of a kernel boundary plus the first round trips of the next kernel.
//
// * Work: the schedule (rows by dependency level) cut into TILES of 4 rows that never straddle a level (levels are padded); tile T of a
//   half-sweep belongs to wavefront T mod W (W = all wavefronts of the grid), 16 lanes per row as in the level kernels, the subtraction in
//   storage order by the same ordered_chunk: the sequential loop's bits.  The backward half walks the tiles in reverse.
// * Hand-off (cdna_hip_programming.md, Guideline 16, form R2: the data is the flag): a row's value is published as two 8-byte granules
//   {epoch, low half} {epoch, high half}, each ONE agent-scope (sc1, write-through) store; a reader polls the two words with agent-scope
//   loads until both carry the epoch it needs -- epoch 1 = after the forward half, 2 = after the backward half; a neighbour that comes LATER in
//   the half-sweep is read at the previous epoch (its value after the previous half; in the forward half the plain input vector).  Nothing
//   else is shared: no flag, no fence, no counter on the critical path.  Two granule arrays take turns: the kernel zeroes the one it does
//   NOT use (epochs restart at 1 in every launch, so the launch works unchanged under graph replay), the last workgroup out flips
//   the selector (a word in device memory).
// * No deadlock by construction: order the tasks (half-sweep, tile); a task waits only for smaller tasks (earlier tiles of its half, any tile
//   of the half before), every wavefront takes its own tasks in increasing order, and every workgroup of the grid is resident (at most one
//   256-thread workgroup per CU) -- so the smallest unfinished task can always run.  Results do not depend on placement or dispatch order.
//   Even so every wait is bounded (ELAG_HIP_SWEEP_DATAFLOW_MS of the 100 MHz clock per wait, default 250): a wavefront that runs out sets the abort
//   word, every poller sees it and leaves, the code lands in pinned host memory and the next ABI call fails (check_device_error) -- another
//   kernel of this kind holding the device, say.  In-place safe: a row's input value is read only by its neighbours' forward tasks, all done
//   before the row's backward task can start.
typedef unsigned long long u64t;
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
struct DfArgs {
    const int4s *slots;
    const unsigned *J;
    const double *a;
    const int *oI, *oJ;
    const double *oa, *ug;
    const double *f, *d;
    double omega;
    double *u;
    u64t *gran;            // [2][2 n]
    int *state;            // [0] workgroups done, [1] array in use, [2] abort, [3] the XCD of the launch (-1 between launches), [4] next tile
    int *host_err;
    int n, npos;
    long long ticks;
    int stall_row;         // -1; a test sets the row whose value is never handed on (ELAG_HIP_SWEEP_DATAFLOW_STALL_ROW): its readers run out of time
};
// kDfWait: an entry that is only waited for (value 0, contributes +0.0): row j gets one for every row i that reads u_j without being read by
// row j (a_ij != 0, a_ji == 0) -- in the level launches the anti-dependency is kept by the levels alone, here row j has to see that row i is done
constexpr unsigned kDfLower = 1u << 30, kDfSelf = 1u << 29, kDfWait = 1u << 28, kDfNone = 1u << 31, kDfCol = (1u << 28) - 1;

// One chunk of a row in the dataflow kernel.  Type 6 leaves the row's first stored diagonal entry out of the sum and divides by it: instead of
// the level kernels' per-step test (three broadcasts and a select per entry: 0.6 us per dependency level here) the lane that holds that entry is
// found once per chunk, hands its value to the row and contributes +0.0 -- subtracting +0.0 leaves every double as it is, so the sum is the same.
template <int MODE>
__device__ __forceinline__ void df_chunk(int row, bool real, int j, double v, double x, double &res, double &aii, bool &have)
{
    if (MODE == 1) {
        const int sh = (int)(threadIdx.x & 48u);
        const unsigned m = (unsigned)((__ballot(real && j == row) >> sh) & 0xffffull);
        if (m != 0u && !have) {
            const int first = __ffs((int)m) - 1;
            aii = __shfl(v, first, kLPR);
            have = true;
            if ((int)(threadIdx.x & (kLPR - 1)) == first) { v = 0.0; x = 0.0; }
        }
    }
    if (!real) { v = 0.0; x = 0.0; }
    double a0 = 0.0;
    bool h0 = false;
    ordered_chunk<0>(row, real ? j : -1, v, x, res, a0, h0);
}

// the XCD this wavefront runs on: bits 3:0 of hardware register 20 (XCC_ID) on gfx942 / gfx950
__device__ __forceinline__ int df_xcc_id()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
#else
    return 0;
#endif
}

constexpr int kDfC = 2;            // chunks of 16 entries a row keeps in registers (32 entries: longer rows fetch the rest when their turn comes)

// XCD = true: the same sweep confined to ONE of the chip's eight XCDs.  A hand-off through agent-scope (write-through) stores costs a trip to
// memory and back, ~1.1 us of the ~1.5 us a dependency level takes above; workgroups of one XCD share its L2, where a PLAIN store of one CU is
// what an L1-bypassing (sc1) load of another CU reads ~0.3 us later.  Placement is not assumed: every workgroup reads the XCD it runs on from
// the hardware (HW_REG_XCC_ID); the first one to arrive names the XCD of the launch (a compare-and-swap on a word in device memory) and the
// workgroups that find themselves elsewhere zero their share of the idle granule array and leave -- a seventh of the grid stays under the
// observed round-robin placement, whatever stays does the work under any other.  Tiles are not dealt in advance but taken from a counter
// (the smallest tile not yet taken; a wavefront holds up to three and works them in order), which also lifts the residency condition:
// a workgroup that has not started holds no tile, so the smallest unfinished tile is always in the hands of a running wavefront.
// (launch bounds: four 256-thread workgroups per CU must fit -- the whole-chip form launches up to that many and needs every one of them resident)
template <int MODE, bool XCD, bool TICKETS = XCD>
__global__ __launch_bounds__(256, 4) void sweep_dataflow_kernel(const DfArgs A)
{
    const int tid = (int)(blockIdx.x * 256 + threadIdx.x), nthreads = (int)gridDim.x * 256;
    const int wave = tid >> 6, nwaves = nthreads >> 6;
    const int grp = (threadIdx.x & 63) >> 4, lane = threadIdx.x & (kLPR - 1);
    const int cur = A.state[1] & 1;
    u64t *g = A.gran + (size_t)cur * 2 * A.n;
    {   // the other array, for the next launch
        u64t *gz = A.gran + (size_t)(cur ^ 1) * 2 * A.n;
        for (size_t i = (size_t)tid; i < (size_t)2 * A.n; i += (size_t)nthreads) gz[i] = 0ull;
    }
    const int ntiles = A.npos >> 2;
    int cnt = wave < ntiles ? (ntiles - wave + nwaves - 1) / nwaves : 0;             // this wavefront's tiles per half-sweep
    bool dead = false;
    if (XCD) {
        // which XCD this launch runs on: the first workgroup's; the others stay only if they are there too
        __shared__ int stay;
        if (threadIdx.x == 0) {
            const int mine = df_xcc_id();
            int seen = -1;
            const bool first = __hip_atomic_compare_exchange_strong(A.state + 3, &seen, mine, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stay = (first || seen == mine) ? 1 : 0;
        }
        __syncthreads();
        cnt = stay ? ntiles : 0;                                                    // (tickets: cnt only says "in" or "out" here)
    } else if (TICKETS) {
        cnt = ntiles;
    }
    // one poll of row j's two words; true once both carry `need`
    // (one 16-byte load for the pair -- each aligned 8-byte half is a granule with its own tag, nothing is assumed about the 16 bytes as a whole --:
    //  half the requests of two 8-byte loads; cube456 x5 level 0 1 906 -> 1 810 us per symmetric sweep, x3 262 -> 260)
    auto poll = [&](int j, unsigned need, double &x) -> bool {
        u32x4s w;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(g + 2 * (size_t)j) : "memory");
        if (w.y != need || w.w != need) return false;
        x = __hiloint2double((int)w.z, (int)w.x);
        return true;
    };
    // the wavefront leaves a wait together; false: it (or somebody else) has run out of time
    // (the limit is per WAIT, counted from its 64th poll: a sweep over a very large operator may take long, a single hand-off may not)
    auto give_up = [&](unsigned &spins, long long &tw) -> bool {
        if ((++spins & 63u) != 0) { __builtin_amdgcn_s_sleep(1); return false; }
        const long long now = (long long)wall_clock64();
        if (spins == 64u) tw = now;
        const bool late = now - tw > A.ticks;
        const bool stop = __hip_atomic_load(A.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (!__any(late || stop)) return false;
        if (__any(late) && (threadIdx.x & 63) == 0) {
            __hip_atomic_store(A.state + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.host_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return true;
    };
    // What a task needs that no other row writes -- its slot, then its first kDfC chunks of entries and its f, d -- is requested two and
    // one tasks ahead: the only memory round trips left on a task's own path are the polls.
    // task number -> (half-sweep, tile); static deal: this wavefront's q-th task; tickets: the number drawn from the counter
    auto take = [&](int q) -> int {
        if (!TICKETS) return q < 2 * cnt ? (q / cnt) * ntiles + wave + (q % cnt) * nwaves : 2 * ntiles;
        if (cnt == 0) return 2 * ntiles;
        int t = 0;
        if ((threadIdx.x & 63) == 0) t = __hip_atomic_fetch_add(A.state + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        return t < 2 * ntiles ? t : 2 * ntiles;
    };
    auto load_slot = [&](int task) -> int4s {
        if (task >= 2 * ntiles) return int4s{-1, 0, 0, 0};
        const int half = task / ntiles, T = task - half * ntiles;
        return A.slots[(size_t)(half == 0 ? T : ntiles - 1 - T) * 4 + grp];
    };
    auto load_entries = [&](const int4s &sl, unsigned (&w)[kDfC], double (&v)[kDfC], double &fr, double &dr) {
#pragma unroll
        for (int c = 0; c < kDfC; c++) {
            w[c] = kDfNone; v[c] = 0.0;
            if (c < sl.z) { w[c] = A.J[(size_t)sl.y + c * kLPR + lane]; v[c] = A.a[(size_t)sl.y + c * kLPR + lane]; }
        }
        fr = 0.0; dr = 0.0;
        if (sl.x >= 0) { fr = A.f[sl.x]; if (MODE == 0) dr = A.d[sl.x]; }
    };
    int tkA = take(0), tkB = take(1), tkC = 2 * ntiles;
    int4s slA = load_slot(tkA), slB = load_slot(tkB), slC = int4s{-1, 0, 0, 0};
    unsigned wA[kDfC], wB[kDfC];
    double vA[kDfC], vB[kDfC], fA, dA, fB = 0.0, dB = 0.0;
    load_entries(slA, wA, vA, fA, dA);
#pragma unroll
    for (int c = 0; c < kDfC; c++) { wB[c] = kDfNone; vB[c] = 0.0; }
    for (int q = 0; tkA < 2 * ntiles && !dead; q++) {
        const int half = tkA / ntiles;
        const unsigned e_new = (unsigned)half + 1u;
        load_entries(slB, wB, vB, fB, dB);
        tkC = take(q + 2);
        slC = load_slot(tkC);
        const int row = slA.x, off = slA.y, nch = slA.z;
        double res = fA, uo = 0.0, aii = 0.0;
        bool have = false;
        // ---- decode; values nobody writes in this launch (the input vector) are plain loads
        int j[kDfC];
        unsigned need[kDfC];
        double x[kDfC];
        bool ok[kDfC];
#pragma unroll
        for (int c = 0; c < kDfC; c++) {
            const unsigned w = wA[c];
            const bool valid = !(w & kDfNone);
            j[c] = (int)(w & kDfCol);
            const bool earlier = valid && !(w & kDfSelf) && (half == 0 ? (w & kDfLower) != 0 : (w & kDfLower) == 0);
            need[c] = earlier ? e_new : e_new - 1u;
            x[c] = 0.0;
            ok[c] = !valid || need[c] == 0u;
            if (valid && need[c] == 0u && !(w & kDfWait)) x[c] = A.u[j[c]];
        }
        const bool self_lane = lane == 0 && row >= 0;           // the row's own value before this half
        bool ok_s = !(self_lane && e_new != 1u);
        if (self_lane && e_new == 1u) uo = A.u[row];
        // ---- the wait
        {
            unsigned spins = 0;
            long long tw = 0;
            for (;;) {
#pragma unroll
                for (int c = 0; c < kDfC; c++)
                    if (!ok[c]) ok[c] = poll(j[c], need[c], x[c]);
                if (!ok_s) ok_s = poll(row, e_new - 1u, uo);
                bool all = ok_s;
#pragma unroll
                for (int c = 0; c < kDfC; c++) all = all && ok[c];
                if (__all(all)) break;
                if (give_up(spins, tw)) { dead = true; break; }
            }
            if (dead) break;
        }
#pragma unroll
        for (int c = 0; c < kDfC; c++)
            if (c < nch) {
                df_chunk<MODE>(row, !(wA[c] & (kDfNone | kDfWait)), j[c], vA[c], x[c], res, aii, have);
            }
        // ---- rows of more than kDfC chunks: the rest, chunk by chunk (the wavefront walks its longest row)
        int maxch = nch;
        maxch = max(maxch, __shfl_xor(maxch, 16, 64));
        maxch = max(maxch, __shfl_xor(maxch, 32, 64));
        for (int c = kDfC; c < maxch && !dead; c++) {
            unsigned w = kDfNone;
            double v = 0.0, xx = 0.0;
            if (c < nch) { w = A.J[(size_t)off + c * kLPR + lane]; v = A.a[(size_t)off + c * kLPR + lane]; }
            const bool valid = !(w & kDfNone);
            const int jj = (int)(w & kDfCol);
            const bool earlier = valid && !(w & kDfSelf) && (half == 0 ? (w & kDfLower) != 0 : (w & kDfLower) == 0);
            const unsigned nd = earlier ? e_new : e_new - 1u;
            bool okx = !valid || nd == 0u;
            if (valid && nd == 0u && !(w & kDfWait)) xx = A.u[jj];
            unsigned spins = 0;
            long long tw = 0;
            for (;;) {
                if (!okx) okx = poll(jj, nd, xx);
                if (__all(okx)) break;
                if (give_up(spins, tw)) { dead = true; break; }
            }
            if (dead) break;
            if (c < nch) df_chunk<MODE>(row, valid && !(w & kDfWait), jj, v, xx, res, aii, have);
        }
        if (dead) break;
        if (row >= 0 && A.oI) {
            const int os = A.oI[row], ol = A.oI[row + 1] - os;
            for (int k0 = 0; k0 < ol; k0 += kLPR) {
                const bool has = k0 + lane < ol;
                const double v = has ? A.oa[os + k0 + lane] : 0.0;
                const double xg = has ? A.ug[A.oJ[os + k0 + lane]] : 0.0;
                ordered_plain(v * xg, res);
            }
        }
        if (row >= 0 && lane == 0 && row != A.stall_row) {
            double un = uo;
            if (MODE == 0) { if (dA != 0.0) un = uo + A.omega * res / dA; }
            else if (have && aii != 0.0) un = res / aii;
            const u64t tag = (u64t)e_new << 32;
            // (XCD: plain stores -- they stay in the XCD's L2, where every reader of this launch looks; otherwise write-through to memory)
            if (XCD) {
                __hip_atomic_store(g + 2 * (size_t)row, tag | (unsigned)__double2loint(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(g + 2 * (size_t)row + 1, tag | (unsigned)__double2hiint(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                __hip_atomic_store(g + 2 * (size_t)row, tag | (unsigned)__double2loint(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(g + 2 * (size_t)row + 1, tag | (unsigned)__double2hiint(un), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (half == 1) A.u[row] = un;
        }
        // ---- rotate the pipeline
        slA = slB; slB = slC; fA = fB; dA = dB; tkA = tkB; tkB = tkC;
#pragma unroll
        for (int c = 0; c < kDfC; c++) { wA[c] = wB[c]; vA[c] = vB[c]; }
    }
    // the last workgroup out hands the arrays over for the next launch (nobody reads state[1] any more: everybody has started)
    __syncthreads();
    if (threadIdx.x == 0 && !dead) {
        const int before = __hip_atomic_fetch_add(A.state, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == (int)gridDim.x - 1) {
            __hip_atomic_store(A.state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.state + 1, cur ^ 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.state + 3, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.state + 4, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The dataflow arrays of a structurally symmetric operator, on the device: a lane group per schedule position writes its row's entries (column |
// flags from the two rows' levels, value) chunk by chunk; every entry (i, j) also looks for (j, i) -- a row that is read without reading back
// needs a wait-only entry, which only the host path below builds (*nanti > 0: it takes over).
__global__ __launch_bounds__(256) void df_build_kernel(int npos, const int4s *__restrict__ slots, const int *__restrict__ I, const int *__restrict__ J,
                                                       const double *__restrict__ a, const int *__restrict__ level, unsigned *__restrict__ dj,
                                                       double *__restrict__ da, int *__restrict__ nanti)
{
    const int pos = (int)((blockIdx.x * 256 + threadIdx.x) / kLPR), lane = threadIdx.x & (kLPR - 1);
    if (pos >= npos) return;
    const int4s sl = slots[pos];
    const int i = sl.x;
    if (i < 0) return;
    const int s0 = I[i], len = I[i + 1] - s0, li = level[i];
    for (int q = lane; q < sl.z * kLPR; q += kLPR) {
        unsigned w = kDfNone;
        double v = 0.0;
        if (q < len) {
            const int j = J[s0 + q];
            v = a[s0 + q];
            w = (unsigned)j;
            if (j == i) w |= kDfSelf;
            else {
                const int lj = level[j];
                if (lj < li) w |= kDfLower;
                bool back = false;
                for (int k = I[j]; k < I[j + 1] && !back; k++) back = J[k] == i;
                if (!back || lj == li) atomicAdd(nanti, 1);
            }
        }
        dj[(size_t)sl.y + q] = w;
        da[(size_t)sl.y + q] = v;
    }
}

// ---- the schedule of a large operator by kernels (SweepSmoother::setup_device) -------------------------------------------------------------
// Dependency levels:  level(i) = 1 + max { level(j) : j < i, a_ij != 0 }  (structurally symmetric pattern: checked afterwards by df_build_kernel) is the
// same kind of chain as the sweep itself, so it is computed the same way: rows in NATURAL order, 64 per wavefront task, a row's lane reads the levels of
// its lower neighbours four at a time and waits for the ones that are not there yet.  lev[] holds level + 1 (0 = not yet), one agent-scope word per row.
// The lanes of a task may depend on each other (rows i and i + 1 are often neighbours), so nobody spins alone: every lane of the wavefront makes one
// attempt per trip of the loop and publishes inside it.  ctl: [0] abort, [1] the largest level + 1.
__global__ __launch_bounds__(256, 4) void df_levels_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, int *lev, int *ctl, long long ticks)
{
    const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int)(gridDim.x * 4), lane = threadIdx.x & 63;
    int wmax = 0;
    bool dead = false;
    for (int task = wave; (long long)task * 64 < n && !dead; task += nwaves) {
        const int row = task * 64 + lane;
        const bool valid = row < n;
        int k = valid ? I[row] : 0;
        const int e = valid ? I[row + 1] : 0;
        int m = 0;
        bool done = !valid;
        unsigned spins = 0;
        long long tw = 0;
        int jj[4] = {-1, -1, -1, -1};
        bool have = false;                                 // jj holds the columns of entries k .. k + 3
        for (;;) {
            if (!done) {
                if (!have) {
#pragma unroll
                    for (int u = 0; u < 4; u++) jj[u] = (k + u < e) ? J[k + u] : -1;
                    have = true;
                }
                int v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (jj[u] >= 0 && jj[u] < row) ? __hip_atomic_load(lev + jj[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
                bool all = true;
#pragma unroll
                for (int u = 0; u < 4; u++) all = all && v[u] != 0;
                if (all) {
#pragma unroll
                    for (int u = 0; u < 4; u++) m = max(m, v[u]);
                    k += 4; have = false;
                    if (k >= e) {
                        __hip_atomic_store(lev + row, m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        wmax = max(wmax, m + 1);
                        done = true;
                    }
                }
            }
            if (__all(done)) break;
            if ((++spins & 63u) == 0) {
                const long long now = (long long)wall_clock64();
                if (spins == 64u) tw = now;
                const bool stop = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (__any(now - tw > ticks || stop)) {
                    if (lane == 0) __hip_atomic_store(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    break;
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o, 64));
    if (lane == 0 && wmax > 0) atomicMax(ctl + 1, wmax);
}

// Rows by (level, row) -- the host path's schedule, as a counting sort in three steps.  (1) df_runs_kernel: a workgroup sorts its 256 consecutive rows by
// level (a bitonic network in LDS on level << 8 | thread), keeps the sorted keys, and the first row of every run of equal levels writes the run's length to
// H[level * nwg + workgroup]; also the chunks of 16 entries per row (at least one).  (2) One exclusive scan over H: S[level * nwg + workgroup] = the rows of
// lower levels plus the rows of this level in workgroups before this one.  (3) df_place_rows_kernel: position = the level's padded base + (S of the run -
// S of the level's first workgroup) + place in the run.  No atomics: the same slots every time, ascending rows inside a level (with one atomic per row the x5
// cycle was 14 % slower -- a tile's rows scattered over memory; with one per run 2 %).
__global__ __launch_bounds__(256) void df_runs_kernel(int n, int nwg, const int *__restrict__ lev, const int *__restrict__ I, unsigned *__restrict__ skey,
                                                      int *__restrict__ H, int *__restrict__ nch16)
{
    __shared__ unsigned key[256];
    const int t = threadIdx.x, i = blockIdx.x * 256 + t;
    key[t] = i < n ? ((unsigned)(lev[i] - 1) << 8) | (unsigned)t : 0xffffffffu;
    if (i < n) nch16[i] = max(1, (I[i + 1] - I[i] + kLPR - 1) / kLPR) * kLPR;
    __syncthreads();
    for (int k = 2; k <= 256; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int p = t ^ j;
            if (p > t) {
                const unsigned a = key[t], b = key[p];
                const bool up = (t & k) == 0;
                if ((a > b) == up) { key[t] = b; key[p] = a; }
            }
            __syncthreads();
        }
    const unsigned mine = key[t];
    skey[(size_t)blockIdx.x * 256 + t] = mine;
    if (mine == 0xffffffffu) return;
    const int l = (int)(mine >> 8);
    if (t == 0 || (int)(key[t - 1] >> 8) != l) {
        int len = 1;
        while (t + len < 256 && key[t + len] != 0xffffffffu && (int)(key[t + len] >> 8) == l) len++;
        H[(size_t)l * nwg + blockIdx.x] = len;
    }
}

// the first entry of every level's row of S (nlev + 1 values: the rows of the levels before it)
__global__ void df_level_starts_kernel(int nlev, int nwg, const int *__restrict__ S, int *__restrict__ start)
{
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l <= nlev) start[l] = S[(size_t)l * nwg];
}

__global__ __launch_bounds__(256) void df_place_rows_kernel(int n, int nwg, const unsigned *__restrict__ skey, const int *__restrict__ S,
                                                            const int *__restrict__ start, const int *__restrict__ base, const int *__restrict__ off,
                                                            const int *__restrict__ nch16, int4s *__restrict__ slots)
{
    __shared__ unsigned key[256];
    const int t = threadIdx.x;
    const unsigned mine = skey[(size_t)blockIdx.x * 256 + t];
    key[t] = mine;
    __syncthreads();
    if (mine == 0xffffffffu) return;
    const int l = (int)(mine >> 8);
    int h = t;                                                         // the head of my run (runs are short: a few steps)
    while (h > 0 && (int)(key[h - 1] >> 8) == l) h--;
    const int row = blockIdx.x * 256 + (int)(mine & 255u);
    const int pos = base[l] + (S[(size_t)l * nwg + blockIdx.x] - start[l]) + (t - h);
    slots[pos] = int4s{row, off[row], nch16[row] / kLPR, 0};
}

__global__ void df_fill_slots_kernel(int npos, int4s *__restrict__ slots)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npos; p += gridDim.x * blockDim.x) slots[p] = int4s{-1, 0, 0, 0};
}

__global__ void df_gather_values_kernel(long long m, const int *__restrict__ src, const double *__restrict__ a, double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) out[i] = src[i] >= 0 ? a[src[i]] : 0.0;
}
// test vectors of the self-check (any reproducible values of mixed sign and magnitude)
__global__ void df_fill_kernel(int n, unsigned salt, double *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned h = ((unsigned)i + salt) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        out[i] = ((double)(h & 0xfffffu) / 524288.0 - 1.0) * (1.0 + (double)((h >> 20) & 7u));
    }
}
__global__ void df_count_diff_kernel(int n, const double *__restrict__ x, const double *__restrict__ y, int *__restrict__ ndiff)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (__double_as_longlong(x[i]) != __double_as_longlong(y[i])) atomicAdd(ndiff, 1);
}

thread_local bool t_one_block = false;        // (the dataflow self-check's operator is swept as one rank whatever the caller has set)
int gs_hybrid_blocks()
{
    if (t_one_block) return 1;
    // elag_hip_set_gs_hybrid_blocks, else ELAG_HIP_GS_BLOCKS, else 1 (one rank: the sequential loop)
    if (ctx().gs_blocks > 0) return ctx().gs_blocks;
    const char *e = getenv("ELAG_HIP_GS_BLOCKS");
    return e && atoi(e) > 0 ? atoi(e) : 1;
}

}  // namespace

void preload_sweep_kernels()
{
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(sched_pack_kernel));
}

SweepSmoother::~SweepSmoother() { dev_free(d_rows); dev_free(d_lev_ptr); dev_free(d_l1); dev_free(d_sJ); dev_free(d_sa); dev_free(d_self_ghost); dev_free(d_blk_first); dev_free(d_blk_count); dev_free(d_lev_rows); dev_free(d_v);
    dev_free(d_df_slots); dev_free(d_df_J); dev_free(d_df_a); dev_free(d_df_gran); dev_free(d_df_state); }

void SweepSmoother::setup()
{
    ELAG_REQUIRE(type == ELAG_HIP_SMOOTHER_L1_GS || type == ELAG_HIP_SMOOTHER_L1_GS_TRUNCATED || type == ELAG_HIP_SMOOTHER_GS ||
                 type == ELAG_HIP_SMOOTHER_KACZMARZ, "internal: relax type %d is not a sweep", type);
    const bool kaczmarz = type == ELAG_HIP_SMOOTHER_KACZMARZ;
    // hypre_ParCSRRelax 2/4 and hypre_BoomerAMGRelax 6 branch on (relax_weight, omega): (1, 1) is the symmetric Gauss-Seidel of every shipped
    // parameter list ("Damping Factor" 1.0, "Omega" 1.0) and the tuned launch plan below; types 2 / 4 with omega != 1 run on it too (a factor
    // in the update); the weighted branches take one plain launch per dependency level (sweep_sor_level_kernel)
    sor = !kaczmarz && (weight != 1.0 || (omega != 1.0 && type == ELAG_HIP_SMOOTHER_GS));
    dev_free(d_v);
    if (sor) d_v = dev_alloc<double>((size_t)std::max(A->nrows(), 1));
    M = A;
    const int n = A->nrows();
    hipStream_t st = ctx().stream;
    As.reset(); dev_free(d_self_ghost); d_self_ghost = nullptr;
    dev_free(d_blk_first); dev_free(d_blk_count); d_blk_first = d_blk_count = nullptr;
    std::vector<int> &block_of = h_block_of;
    block_of.clear();
    blocks = gs_hybrid_blocks();
    if (blocks > 1 && !kaczmarz && !is_distributed(*A) && A->nrows() == A->ncols() && n >= 2 * blocks) {
        // ---- virtual ranks: A split like a ParCSR matrix on `blocks` ranks with contiguous, even row blocks ----
        std::vector<int> I = host_from_dev(A->diag.I, (size_t)n + 1), J = host_from_dev(A->diag.J, (size_t)A->diag.nnz);
        std::vector<double> a = host_from_dev(A->diag.a, (size_t)A->diag.nnz);
        const int size = n / blocks, rest = n - size * blocks;
        block_of.assign((size_t)n, 0);
        for (int b = 0, r = 0; b < blocks; b++) {
            const int len = size + (b < rest ? 1 : 0);
            for (int q = 0; q < len; q++) block_of[(size_t)r++] = b;
        }
        std::vector<int> dI((size_t)n + 1, 0), oI((size_t)n + 1, 0), dJ, oJ;
        std::vector<double> da, oa;
        std::vector<char> is_ghost((size_t)n, 0);
        for (int i = 0; i < n; i++)
            for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++)
                if (block_of[(size_t)J[(size_t)k]] != block_of[(size_t)i]) is_ghost[(size_t)J[(size_t)k]] = 1;
        std::vector<int> ghost_id((size_t)n, -1), ghost_col;
        for (int j = 0; j < n; j++)
            if (is_ghost[(size_t)j]) { ghost_id[(size_t)j] = (int)ghost_col.size(); ghost_col.push_back(j); }
        for (int i = 0; i < n; i++) {
            for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) {
                const int j = J[(size_t)k];
                if (block_of[(size_t)j] == block_of[(size_t)i]) { dJ.push_back(j); da.push_back(a[(size_t)k]); }
                else { oJ.push_back(ghost_id[(size_t)j]); oa.push_back(a[(size_t)k]); }
            }
            dI[(size_t)i + 1] = (int)dJ.size(); oI[(size_t)i + 1] = (int)oJ.size();
        }
        if (!ghost_col.empty()) {
            As.reset(new ParCSR());
            As->diag.upload(n, n, dI.data(), dJ.data(), da.data());
            As->offd.upload(n, (int)ghost_col.size(), oI.data(), oJ.data(), oa.data());
            As->d_ghost = dev_alloc<double>(ghost_col.size());
            As->first_row = As->first_col = 0; As->global_rows = As->global_cols = n;
            d_self_ghost = dev_from_host(ghost_col.data(), ghost_col.size());
            M = As.get();
        } else {
            blocks = 1;
        }
    } else {
        blocks = 1;
    }
    if (type != ELAG_HIP_SMOOTHER_GS) {
        // mfem::HypreSmoother::SetOperator: l1_norms = hypre_ParCSRComputeL1Norms(A, type) for types 1..4 (3: the rows' squared 2-norms)
        Temp<int> bad(1);
        zero_async(bad, sizeof(int));
        d_l1 = dev_alloc<double>((size_t)std::max(n, 1));
        if (n) {
            const bool ho = M->offd.nnz > 0;
            const unsigned g = (unsigned)std::min<long long>(((long long)n + 255) / 256, 2048);
            hipLaunchKernelGGL(l1_norms_ordered_kernel, dim3(g), dim3(256), 0, st, n, (const int *)M->diag.I, (const int *)M->diag.J,
                               (const double *)M->diag.a, ho ? (const int *)M->offd.I : nullptr, ho ? (const double *)M->offd.a : nullptr, type,
                               d_l1, bad.p);                     // (option = relax type: 2, 3 or 4)
            ELAG_LAUNCH_CHECK();
        }
        int nbad = 0;
        stage_d2h(&nbad, bad.p, sizeof(int));
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        if (nbad) fail(ELAG_HIP_ERR_NUMERIC, "%s: matrix has %d zero l1 row norm(s)", kaczmarz ? "Kaczmarz" : "l1 Gauss-Seidel", nbad);
    }
    host_ready = false;
    dataflow = false;
    h_lev_ptr.clear();
    // large operators: dependency levels, schedule and entries by kernels (below); everything else: the host's pass over the rows
    if (setup_device()) return;
    setup_host(true);
}

void SweepSmoother::setup_host(bool allow_dataflow)
{
    const bool kaczmarz = type == ELAG_HIP_SMOOTHER_KACZMARZ;
    const int n = M->nrows();
    hipStream_t st = ctx().stream;
    const std::vector<int> &block_of = h_block_of;
    // ---- dependency levels of the symmetrised pattern, one pass over the rows ----
    std::vector<int> I = host_from_dev(M->diag.I, (size_t)n + 1);
    std::vector<int> J = host_from_dev(M->diag.J, (size_t)M->diag.nnz);
    std::vector<int> level((size_t)n, 0);     // doubles as "pending" for rows not reached yet
    int nlev = n ? 1 : 0;
    if (kaczmarz) {
        // a Kaczmarz step reads AND writes u on the columns of its row: two rows are ordered iff they share a local column.
        //     level(i) = 1 + max { level(k) : k < i, rows i and k share a column }
        // through the highest level seen so far on every column; the rows of a level then share no column at all.
        std::vector<int> col_level((size_t)std::max(M->diag.ncols, 1), -1), seen((size_t)std::max(M->diag.ncols, 1), -1);
        for (int i = 0; i < n; i++) {
            int lv = 0;
            for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) {
                const int j = J[(size_t)k];
                if (seen[(size_t)j] == i) fail(ELAG_HIP_ERR_ARG, "Kaczmarz: row %d stores column %d twice", i, j);
                seen[(size_t)j] = i;
                lv = std::max(lv, col_level[(size_t)j] + 1);
            }
            level[(size_t)i] = lv;
            nlev = std::max(nlev, lv + 1);
            for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) col_level[(size_t)J[(size_t)k]] = lv;
        }
    }
    for (int i = 0; i < n && !kaczmarz; i++) {
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
    num_levels = nlev;
    {
        // rows by dependency level, for apply_dense
        h_lev_ptr.assign((size_t)nlev + 1, 0);
        for (int i = 0; i < n; i++) h_lev_ptr[(size_t)level[(size_t)i] + 1]++;
        std::partial_sum(h_lev_ptr.begin(), h_lev_ptr.end(), h_lev_ptr.begin());
        std::vector<int> cur(h_lev_ptr.begin(), h_lev_ptr.end() - 1), lr((size_t)std::max(n, 1));
        for (int i = 0; i < n; i++) lr[(size_t)cur[(size_t)level[(size_t)i]]++] = i;
        dev_free(d_lev_rows);
        d_lev_rows = dev_from_host(lr.data(), lr.size());
    }
    // Virtual ranks: the blocks share nothing during a sweep, so each gets a chain of its own -- rows ordered by (block, level), a "level" below
    // being one block's rows of one dependency level -- and ONE launch of a workgroup per block walks all of them, forward and backward
    // (ELAG_HIP_SWEEP_BLOCK_CHAINS=0: the blocks' levels side by side in the common schedule, one launch or chain step per level)
    // Measured on MI355X (cube456 x3 L1 Gauss-Seidel cycle, V-cycles/s, chains / common schedule): B = 8: 680 / 1 011, 16: 1 108 / 1 179,
    // 32: 1 562 / 1 294, 128: 2 557 / 1 763, 256: 3 012 / 1 979, 512: 2 760 / 2 417, 2 048: 1 638 / 4 091 -- a chain per block pays while a block
    // is not a long chain of its own (few blocks: few CUs at work) and the blocks are not thousands of workgroups for a handful of rows each:
    // blocks of at most 2 048 rows, at most 512 of them (1 024: 2 127 / ~3 000).  ELAG_HIP_SWEEP_BLOCK_CHAINS = 0 / 1 forces either.
    bool block_chains = false;
    if (As && nlev > 0 && (long long)blocks * nlev < (1ll << 27)) {
        const char *e = getenv("ELAG_HIP_SWEEP_BLOCK_CHAINS");
        const int per_block = n / blocks;
        block_chains = e ? atoi(e) != 0 : (per_block <= 2048 && blocks <= 512);
    }
    const int nkeys = block_chains ? blocks * nlev : nlev;
    auto key_of = [&](int i) { return block_chains ? block_of[(size_t)i] * nlev + level[(size_t)i] : level[(size_t)i]; };
    std::vector<int> ptr((size_t)nkeys + 1, 0), rows((size_t)n);
    for (int i = 0; i < n; i++) ptr[(size_t)key_of(i) + 1]++;
    std::partial_sum(ptr.begin(), ptr.end(), ptr.begin());
    {
        std::vector<int> cur(ptr.begin(), ptr.end() - 1);
        for (int i = 0; i < n; i++) rows[(size_t)cur[(size_t)key_of(i)]++] = i;   // ascending row inside a level
    }
    nlev = nkeys;                              // (from here on a "level" is a key)
    // steps: every level cut into pieces of at most kStepRows rows (the rows of a level are independent, so the pieces may run
    // one after the other); a segment is either one big level -- all its steps as ONE multi-workgroup launch -- or a run of
    // small levels whose steps the chain workgroup walks
    std::vector<int> step_ptr(1, 0), step_first;
    std::vector<int> first_step((size_t)nlev + 1, 0);
    for (int l = 0; l < nlev; l++) {
        first_step[(size_t)l] = (int)step_ptr.size() - 1;
        for (int b = ptr[(size_t)l]; b < ptr[(size_t)l + 1]; b += kStepRows) {
            step_ptr.push_back(std::min(b + kStepRows, ptr[(size_t)l + 1]));
            step_first.push_back(b == ptr[(size_t)l] ? 1 : 0);
        }
    }
    step_first.push_back(1);                                   // (the position after the last step)
    first_step[(size_t)nlev] = (int)step_ptr.size() - 1;
    // Levels up to chain_rows rows are walked by the chain workgroup, larger ones get a launch of their own.  Measured on MI355X
    // (ELAG_HIP_SWEEP_CHAIN_ROWS; cube456 x3 H1 cycle / H(div) Hiptmair cycle, V-cycles per second): 8192: 179 / 19, 2048: 241 / 37,
    // 512: 398 / 59, 256: 492 / 75, 128: 595 / 93, 64: 612 / 94, 16: 579 / 90, 1: 577 / 88 -- a chain step of 128 rows costs about
    // what a launch costs, so a level is chained only when it fits ONE step.
    static int chain_rows = 0;
    if (chain_rows == 0) {
        const char *e = getenv("ELAG_HIP_SWEEP_CHAIN_ROWS");
        // (never more than a chain launch can hold: a level of more steps than kMaxChain would otherwise neither be chained nor advance the loop below)
        chain_rows = e && atoi(e) > 0 ? std::min(atoi(e), kMaxChain * kStepRows) : kChainRows;
    }
    // operators whose u fits the chain workgroup's LDS: every level joins the chain.  OFF unless ELAG_HIP_SWEEP_LDS=1: measured on MI355X
    // (cube456 x3, L1 Gauss-Seidel cycle) 572 V-cycles/s against 632 for the plan below although the cycle has 50 launches fewer -- with u in
    // LDS a step no longer waits for u but for the NEXT step's matrix entries, which are requested only one step ahead (1-2 us of memory
    // latency against ~0.3 us of work), and a 500-row level costs four such steps instead of one 4 us launch.  The lever is a deeper
    // prefetch (or the level operators in LDS, which fit only below ~10^4 entries), not where u lives.  Results are the other path's bit for bit.
    {
        const char *e = getenv("ELAG_HIP_SWEEP_LDS");          // (read at every setup: a test switches it)
        lds_u = n > 0 && n <= kLdsRows && e && atoi(e);
    }
    const int cr = lds_u ? kMaxChain * kStepRows : chain_rows;
    segments.clear();
    longest_chain = 0;
    if (block_chains) {
        std::vector<int> bf((size_t)blocks), bc((size_t)blocks);
        int longest = 0;
        for (int b = 0; b < blocks; b++) {
            bf[(size_t)b] = first_step[(size_t)b * num_levels];
            bc[(size_t)b] = first_step[(size_t)(b + 1) * num_levels] - bf[(size_t)b];
            longest = std::max(longest, bc[(size_t)b]);
        }
        if (longest <= kMaxChain) {
            longest_chain = longest;
            d_blk_first = dev_from_host(bf.data(), bf.size());
            d_blk_count = dev_from_host(bc.data(), bc.size());
            lds_u = false;
            segments.push_back(Segment{0, (int)step_ptr.size() - 1, -1, 0, 0});      // max_rows < 0: a workgroup per block
        } else {
            fail(ELAG_HIP_ERR_ARG, "Gauss-Seidel on %d virtual ranks: a block's chain has %d steps (at most %d): use more blocks or ELAG_HIP_SWEEP_BLOCK_CHAINS=0",
                 blocks, longest, kMaxChain);          // (blocks of at most 2 048 rows cannot get here unless the chains were forced)
        }
    }
    for (int l = 0; l < nlev && !block_chains;) {
        const int sz = ptr[(size_t)l + 1] - ptr[(size_t)l];
        if (sz > cr) { segments.push_back(Segment{first_step[(size_t)l], first_step[(size_t)l + 1] - first_step[(size_t)l], sz, ptr[(size_t)l], ptr[(size_t)l + 1]}); l++; continue; }
        int m = l;
        while (m < nlev && first_step[(size_t)m + 1] - first_step[(size_t)l] <= kMaxChain && ptr[(size_t)m + 1] - ptr[(size_t)m] <= cr) m++;
        if (m == l) {    // a single level with more steps than one chain launch holds: a launch of its own
            segments.push_back(Segment{first_step[(size_t)l], first_step[(size_t)l + 1] - first_step[(size_t)l], sz, ptr[(size_t)l], ptr[(size_t)l + 1]});
            l++;
            continue;
        }
        segments.push_back(Segment{first_step[(size_t)l], first_step[(size_t)m] - first_step[(size_t)l], 0, 0, 0});
        l = m;
    }
    // schedule slots: (row, first entry, row length) in level order -- one 16-byte load per row instead of three dependent ones
    std::vector<int> slot((size_t)std::max(n, 1) * 4, 0);
    for (int t = 0; t < n; t++) {
        const int i = rows[(size_t)t];
        slot[(size_t)t * 4] = i; slot[(size_t)t * 4 + 1] = I[(size_t)i]; slot[(size_t)t * 4 + 2] = I[(size_t)i + 1] - I[(size_t)i];
    }
    d_rows = dev_from_host(slot.data(), slot.size());
    dev_free(d_sJ); dev_free(d_sa);
    d_sJ = dev_alloc<int>((size_t)std::max(n, 1) * kLPR);
    d_sa = dev_alloc<double>((size_t)std::max(n, 1) * kLPR);
    if (n) {
        const unsigned g = (unsigned)std::min<long long>(((long long)n * kLPR + 255) / 256, 4096);
        hipLaunchKernelGGL(sched_pack_kernel, dim3(g), dim3(256), 0, st, n, reinterpret_cast<const int4s *>(d_rows), (const int *)M->diag.J,
                           (const double *)M->diag.a, d_sJ, d_sa);
        ELAG_LAUNCH_CHECK();
    }
    n_steps = (int)step_ptr.size() - 1;
    step_ptr.insert(step_ptr.end(), step_first.begin(), step_first.end());       // [offsets (n_steps + 1) | first-of-level flags (n_steps + 1)]
    d_lev_ptr = dev_from_host(step_ptr.data(), step_ptr.size());
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    host_ready = true;
    // (virtual ranks too where their blocks share the common schedule: the diagonal part of the split operator is what the levels were computed on)
    if (allow_dataflow && !dataflow && !kaczmarz && !sor && !block_chains) setup_dataflow(I, J, level, ptr, rows);
    if (getenv("ELAG_HIP_SWEEP_VERBOSE"))
        fprintf(stderr, "[elag_hip] sweep smoother type %d: n=%d, %d dependency levels = %d steps in %d launches per direction%s%s\n", type, n, nlev,
                n_steps, (int)segments.size(), lds_u ? " (u in LDS)" : "", dataflow ? "; dataflow sweep: one launch per symmetric sweep" : "");
}

namespace {
// ELAG_HIP_SWEEP_DATAFLOW: 0 = never; unset / 1 = where it pays (below); 2 = wherever it is possible (tests)
int dataflow_mode()
{
    const char *e = getenv("ELAG_HIP_SWEEP_DATAFLOW");       // (read at every setup: tests switch it)
    return e ? atoi(e) : 1;
}
std::atomic<int> g_dataflow_checked{0};    // 0: not yet, 1: this device runs the dataflow sweep bit for bit like the level launches, -1: it does not (off for the process)
thread_local bool g_self_check_running = false;     // (this thread is inside dataflow_self_check: its own smoother must not start another)

// Once per process, before the first operator takes the dataflow sweep: a 12 x 12 x 12 seven-point operator (1 728 rows, 34 dependency levels) swept by
// the level launches and by the dataflow kernel (both granule arrays), bit for bit -- does THIS device run the hand-off protocol?  On the library's
// own operator rather than the caller's first one: the check costs the same ~1 ms whatever the caller builds, and needs no launch-per-level plan of a
// large operator (setup_device builds none).
void dataflow_self_check()
{
    if (g_dataflow_checked.load() != 0 || g_self_check_running) return;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_dataflow_checked.load() != 0) return;
    g_self_check_running = true;
    try {
        const int m = 12, n = m * m * m;
        std::vector<int> I(1, 0), J;
        std::vector<double> a;
        for (int z = 0; z < m; z++)
            for (int y = 0; y < m; y++)
                for (int x = 0; x < m; x++) {
                    const int i = (z * m + y) * m + x;
                    J.push_back(i); a.push_back(6.5 + 0.01 * (i % 7));           // the diagonal entry first (hypre's convention)
                    auto off = [&](int j, double v) { J.push_back(j); a.push_back(v); };
                    if (x > 0) off(i - 1, -1.0);
                    if (x < m - 1) off(i + 1, -1.0);
                    if (y > 0) off(i - m, -0.9);
                    if (y < m - 1) off(i + m, -0.9);
                    if (z > 0) off(i - m * m, -1.1);
                    if (z < m - 1) off(i + m * m, -1.1);
                    I.push_back((int)J.size());
                }
        DevCSR d;
        d.upload(n, n, I.data(), J.data(), a.data());
        std::unique_ptr<ParCSR> P(parcsr_from_diag(std::move(d), 0, 0, n, n));
        SweepSmoother T;
        T.A = P.get();
        T.type = ELAG_HIP_SMOOTHER_L1_GS;
        t_one_block = true;
        try { T.setup(); } catch (...) { t_one_block = false; throw; }
        t_one_block = false;                       // (small: the host path, whose setup_dataflow runs the comparison while g_dataflow_checked is 0)
        if (g_dataflow_checked.load() == 0) g_dataflow_checked.store(-1);          // (the operator did not take the dataflow form at all: leave it off)
    } catch (...) {
        g_dataflow_checked.store(-1);
    }
    g_self_check_running = false;
}
}  // namespace

void SweepSmoother::finish_dataflow(int nlev)
{
    Context &c = ctx();
    const int n = M->nrows();
    d_df_gran = dev_alloc<unsigned long long>((size_t)4 * n);
    {
        const int st0[8] = {0, 0, 0, -1, 0, 0, 0, 0};
        d_df_state = dev_from_host(st0, 8);
    }
    zero_async(d_df_gran, sizeof(unsigned long long) * (size_t)4 * n);
    // every workgroup resident whatever else the kernel needs: at most one 256-thread workgroup per CU; two tiles per wavefront and half-sweep at least
    const int ntiles = df_npos / 4, cus = c.num_cus > 0 ? c.num_cus : 64;
    df_grid = std::max(1, std::min(cus, (ntiles + 7) / 8));
    {
        // wide levels want more wavefronts than one workgroup per CU gives (a wavefront works one tile at a time, ~2 us each): about one workgroup
        // per tile of an average level, four per CU at most (all resident).  Measured, us per symmetric sweep with 256 / 512 / 1 024 workgroups:
        // cube456 x3 (119 tiles per level) 265 / 277 / 317; x4 (480 tiles per level) 693 / 600 / 614.
        const int tpl = ntiles / std::max(nlev, 1);
        if (tpl > cus) {
            int g2 = cus;
            while (g2 < tpl && g2 < 4 * cus) g2 *= 2;
            df_grid = std::min(g2, 4 * cus);
        }
    }
    {
        // One XCD or the whole chip (ELAG_HIP_SWEEP_DATAFLOW_XCD = 1 / 0 forces either).  Measured on MI355X, a symmetric sweep of the cube456 x3
        // level operators, whole chip / one XCD (a workgroup per CU launched, 32 of them stay): 42 309 rows in 89 levels (119 tiles per level)
        // 262 / 405 us; 5 739 rows in 51 levels (28 tiles per level) 148 / 121 us; 840 rows in 41 levels 118 / 93 us; a tridiagonal chain 1.08 /
        // 0.85 us per hop -- an XCD's L2 makes the hand-off shorter, its 32 CUs make a wide level longer.
        const char *e = getenv("ELAG_HIP_SWEEP_DATAFLOW_XCD");
        df_xcd = e ? atoi(e) != 0 : (ntiles <= 48 * nlev);
        if (df_xcd) df_grid = cus;
    }
    if (const char *e = getenv("ELAG_HIP_SWEEP_DATAFLOW_WGS")) df_grid = std::max(1, std::min(atoi(e), df_xcd ? 8 * cus : 4 * cus));   // (whole chip: every workgroup must be resident -- 4 per CU at most)
}

void SweepSmoother::setup_dataflow(const std::vector<int> &I, const std::vector<int> &J, const std::vector<int> &level, const std::vector<int> &ptr,
                                   const std::vector<int> &rows)
{
    dev_free(d_df_slots); dev_free(d_df_J); dev_free(d_df_a); dev_free(d_df_gran); dev_free(d_df_state);
    d_df_slots = nullptr; d_df_J = nullptr; d_df_a = nullptr; d_df_gran = nullptr; d_df_state = nullptr;
    const int mode = dataflow_mode();
    const int n = M->nrows(), nlev = (int)ptr.size() - 1;
    Context &c = ctx();
    if (mode > 0 && c.comm_kind != 2 && c.h_dev_err) dataflow_self_check();
    // one rank thread per process only (several contexts of one process would run such kernels side by side: each waits for workgroups the
    // other ones' spinning workgroups keep off the CUs); local square diagonal block; and enough dependency levels for a launch per level to hurt
    if (mode <= 0 || g_dataflow_checked.load() < 0 || c.comm_kind == 2 || n <= 0 || M->diag.ncols != n || n >= (1 << 28) || !c.h_dev_err) return;
    if (g_dataflow_checked.load() == 0 && !g_self_check_running) return;      // (the self-check could not run)
    if (mode == 1 && (nlev < 4 || n < 256)) return;
    hipStream_t st = c.stream;
    // ---- the device path: slots from the row lengths alone, entries and flags by a kernel; taken unless some row is read without reading back ----
    {
        long long npos0 = 0, nent0 = 0;
        for (int l = 0; l < nlev; l++) npos0 += ((long long)(ptr[(size_t)l + 1] - ptr[(size_t)l]) + 3) / 4 * 4;
        for (int i = 0; i < n; i++) nent0 += (long long)std::max(1, (I[(size_t)i + 1] - I[(size_t)i] + kLPR - 1) / kLPR) * kLPR;
        if (npos0 >= (1ll << 30) || nent0 >= (1ll << 31)) return;
        std::vector<int> slots0((size_t)npos0 * 4, 0);
        long long pos0 = 0, off0 = 0;
        for (int l = 0; l < nlev; l++) {
            for (int t = ptr[(size_t)l]; t < ptr[(size_t)l + 1]; t++, pos0++) {
                const int i = rows[(size_t)t], nch = std::max(1, (I[(size_t)i + 1] - I[(size_t)i] + kLPR - 1) / kLPR);
                slots0[(size_t)pos0 * 4] = i; slots0[(size_t)pos0 * 4 + 1] = (int)off0; slots0[(size_t)pos0 * 4 + 2] = nch;
                off0 += (long long)nch * kLPR;
            }
            for (; pos0 % 4 != 0; pos0++) slots0[(size_t)pos0 * 4] = -1;
        }
        d_df_slots = dev_from_host(slots0.data(), slots0.size());
        int *d_level = dev_from_host(level.data(), (size_t)n);
        d_df_J = dev_alloc<unsigned>((size_t)nent0);
        d_df_a = dev_alloc<double>((size_t)nent0);
        Temp<int> nanti(1);
        zero_async(nanti, sizeof(int));
        hipLaunchKernelGGL(df_build_kernel, dim3((unsigned)((npos0 * kLPR + 255) / 256)), dim3(256), 0, st, (int)npos0, reinterpret_cast<const int4s *>(d_df_slots),
                           (const int *)M->diag.I, (const int *)M->diag.J, (const double *)M->diag.a, (const int *)d_level, d_df_J, d_df_a, nanti.p);
        ELAG_LAUNCH_CHECK();
        const int na = fetch_int(nanti);
        dev_free(d_level);
        if (na == 0) df_npos = (int)npos0;
        else { dev_free(d_df_slots); dev_free(d_df_J); dev_free(d_df_a); d_df_slots = nullptr; d_df_J = nullptr; d_df_a = nullptr; }
    }
    if (!d_df_slots) {
    // ---- the host path: anti-dependencies -- rows i that store column j while row j does not store column i ----
    std::vector<int> tptr((size_t)n + 1, 0), tidx((size_t)J.size()), xptr((size_t)n + 1, 0), xidx;
    for (int i = 0; i < n; i++)
        for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) tptr[(size_t)J[(size_t)k] + 1]++;
    std::partial_sum(tptr.begin(), tptr.end(), tptr.begin());
    {
        std::vector<int> fill(tptr.begin(), tptr.end() - 1);
        for (int i = 0; i < n; i++)
            for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) tidx[(size_t)fill[(size_t)J[(size_t)k]]++] = i;
        std::vector<int> mark((size_t)n, -1);
        for (int j = 0; j < n; j++) {
            for (int k = I[(size_t)j]; k < I[(size_t)j + 1]; k++) mark[(size_t)J[(size_t)k]] = j;
            for (int q = tptr[(size_t)j]; q < tptr[(size_t)j + 1]; q++) {
                const int i = tidx[(size_t)q];
                if (i != j && mark[(size_t)i] != j) { mark[(size_t)i] = j; xidx.push_back(i); }
            }
            xptr[(size_t)j + 1] = (int)xidx.size();
        }
    }
    // ---- tiles: the levels padded to whole wave tiles of 4 positions; rows padded to whole chunks of 16 entries (at least one) ----
    auto chunks_of = [&](int i) { return std::max(1, (I[(size_t)i + 1] - I[(size_t)i] + xptr[(size_t)i + 1] - xptr[(size_t)i] + kLPR - 1) / kLPR); };
    long long npos = 0, nent = 0;
    for (int l = 0; l < nlev; l++) npos += ((long long)(ptr[(size_t)l + 1] - ptr[(size_t)l]) + 3) / 4 * 4;
    for (int i = 0; i < n; i++) nent += (long long)chunks_of(i) * kLPR;
    if (npos >= (1ll << 30) || nent >= (1ll << 31)) return;
    std::vector<int> slots((size_t)npos * 4, 0), src((size_t)std::max<long long>(nent, 1), -1);
    std::vector<unsigned> dj((size_t)std::max<long long>(nent, 1), kDfNone);
    long long pos = 0, off = 0;
    for (int l = 0; l < nlev; l++) {
        for (int t = ptr[(size_t)l]; t < ptr[(size_t)l + 1]; t++, pos++) {
            const int i = rows[(size_t)t], len = I[(size_t)i + 1] - I[(size_t)i], nch = chunks_of(i);
            slots[(size_t)pos * 4] = i; slots[(size_t)pos * 4 + 1] = (int)off; slots[(size_t)pos * 4 + 2] = nch;
            auto flags = [&](int j) -> unsigned {
                if (j == i) return kDfSelf;
                return level[(size_t)j] < level[(size_t)i] ? kDfLower : 0u;
            };
            for (int q = 0; q < len; q++) {
                const int k = I[(size_t)i] + q, j = J[(size_t)k];
                if (j != i && level[(size_t)j] == level[(size_t)i]) return;      // (cannot be: two rows of one level share no entry)
                dj[(size_t)off + q] = (unsigned)j | flags(j);
                src[(size_t)off + q] = k;
            }
            for (int q = xptr[(size_t)i]; q < xptr[(size_t)i + 1]; q++) {
                const int j = xidx[(size_t)q];
                if (level[(size_t)j] == level[(size_t)i]) return;
                dj[(size_t)off + len + (q - xptr[(size_t)i])] = (unsigned)j | flags(j) | kDfWait;
            }
            off += (long long)nch * kLPR;
        }
        for (; pos % 4 != 0; pos++) { slots[(size_t)pos * 4] = -1; slots[(size_t)pos * 4 + 2] = 0; }
    }
    df_npos = (int)npos;
    d_df_slots = dev_from_host(slots.data(), slots.size());
    d_df_J = dev_from_host(dj.data(), dj.size());
    d_df_a = dev_alloc<double>(dj.size());
    {
        int *d_src = dev_from_host(src.data(), src.size());
        const unsigned gsz = (unsigned)std::min<long long>(((long long)dj.size() + 255) / 256, 4096);
        hipLaunchKernelGGL(df_gather_values_kernel, dim3(gsz), dim3(256), 0, st, (long long)dj.size(), (const int *)d_src, (const double *)M->diag.a, d_df_a);
        ELAG_LAUNCH_CHECK();
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        dev_free(d_src);
    }
    }   // host path
    finish_dataflow(nlev);
    dataflow = true;
    // ---- the same symmetric sweep by the level launches and by the dataflow kernel, bit for bit: on the process's self-check operator
    //      (dataflow_self_check below), or on every operator that comes this way with ELAG_HIP_SWEEP_DATAFLOW_CHECK=1 ----
    const char *always = getenv("ELAG_HIP_SWEEP_DATAFLOW_CHECK");
    if (g_dataflow_checked.load() == 0 || (always && atoi(always))) {
        Temp<double> tf((size_t)n), ua((size_t)n), ub((size_t)n);
        Temp<int> nd(1);
        const unsigned gsz = (unsigned)std::min<long long>(((long long)n + 255) / 256, 2048);
        hipLaunchKernelGGL(df_fill_kernel, dim3(gsz), dim3(256), 0, st, n, 17u, tf.p);
        hipLaunchKernelGGL(df_fill_kernel, dim3(gsz), dim3(256), 0, st, n, 90001u, ua.p);
        hipLaunchKernelGGL(df_fill_kernel, dim3(gsz), dim3(256), 0, st, n, 90001u, ub.p);
        zero_async(nd, sizeof(int));
        const bool ghosts = M->offd.nnz > 0;
        if (ghosts) zero_async(M->d_ghost, sizeof(double) * (size_t)M->offd.ncols);      // (any frozen ghost values, the same for both)
        for (size_t q = 0; q < segments.size(); q++) launch_segment(segments[q], +1, tf, ua);
        for (size_t q = segments.size(); q-- > 0;) launch_segment(segments[q], -1, tf, ua);
        launch_dataflow(tf, ub);
        launch_dataflow(tf, ub);                                                          // (twice: the second launch runs on the other granule array)
        for (size_t q = 0; q < segments.size(); q++) launch_segment(segments[q], +1, tf, ua);
        for (size_t q = segments.size(); q-- > 0;) launch_segment(segments[q], -1, tf, ua);
        hipLaunchKernelGGL(df_count_diff_kernel, dim3(gsz), dim3(256), 0, st, n, (const double *)ua.p, (const double *)ub.p, nd.p);
        ELAG_LAUNCH_CHECK();
        const int ndiff = fetch_int(nd);
        const int gave_up = *reinterpret_cast<volatile int *>(c.h_dev_err);
        if (ndiff != 0 || gave_up != 0) {
            *reinterpret_cast<volatile int *>(c.h_dev_err) = 0;
            g_dataflow_checked.store(-1);
            dataflow = false;
            if (!getenv("ELAG_HIP_QUIET"))
                fprintf(stderr, "[elag_hip] the Gauss-Seidel dataflow sweep %s on this device: the launch-per-level sweeps stay (n = %d, %d levels, %d workgroups)\n",
                        gave_up ? "gave up waiting" : "did not reproduce the level launches bit for bit", n, nlev, df_grid);
        } else {
            g_dataflow_checked.store(1);
        }
    }
}

bool SweepSmoother::setup_device()
{
    // from 20 000 rows: measured on the 42 309-row level of cube456 x3 the kernels take 1.8 ms against the host's 2.5 (same schedule, same sweep), on 5 739 rows 0.42
    // against 0.46; below, the host path also leaves the launch-per-level plan a dense tail is built with
    int min_rows = 20000;                              // (read at every setup: a test forces the kernels on a small operator)
    if (const char *e = getenv("ELAG_HIP_SWEEP_DEVICE_SETUP_ROWS")) min_rows = atoi(e);
    const int mode = dataflow_mode();
    Context &c = ctx();
    const int n = M->nrows();
    const char *always = getenv("ELAG_HIP_SWEEP_DATAFLOW_CHECK");
    if (type == ELAG_HIP_SMOOTHER_KACZMARZ || sor || blocks != 1 || mode <= 0 || c.comm_kind == 2 || !c.h_dev_err || n < min_rows || M->diag.ncols != n ||
        n >= (1 << 28) || (always && atoi(always)))
        return false;
    dataflow_self_check();
    if (g_dataflow_checked.load() <= 0) return false;
    static long long ticks = 0;
    if (ticks == 0) { const char *e = getenv("ELAG_HIP_SWEEP_DATAFLOW_MS"); ticks = std::max(1ll, (long long)((e && atof(e) > 0 ? atof(e) : 250.0) * 1.0e5)); }
    hipStream_t st = c.stream;
    const int cus = c.num_cus > 0 ? c.num_cus : 64;
    // ---- 1. dependency levels ----
    int *d_level = dev_alloc<int>((size_t)n);
    Temp<int> ctl(2);
    zero_async(d_level, sizeof(int) * (size_t)n);
    zero_async(ctl, sizeof(int) * 2);
    {
        const int tasks = (n + 63) / 64;
        const unsigned g = (unsigned)std::max(1, std::min(4 * cus, (tasks + 3) / 4));      // (every workgroup resident: 4 per CU at most)
        hipLaunchKernelGGL(df_levels_kernel, dim3(g), dim3(256), 0, st, n, (const int *)M->diag.I, (const int *)M->diag.J, d_level, ctl.p, ticks);
        ELAG_LAUNCH_CHECK();
    }
    int hc[2] = {0, 0};
    fetch_ints(ctl, 2, hc);
    const int nlev = hc[1];
    auto give_up = [&]() { dev_free(d_level); dev_free(d_df_slots); dev_free(d_df_J); dev_free(d_df_a); d_df_slots = nullptr; d_df_J = nullptr; d_df_a = nullptr; return false; };
    if (hc[0] != 0 || nlev <= 0) return give_up();                                  // (ran out of time)
    if (mode == 1 && nlev < 4) return give_up();
    // ---- 2. rows by (level, row): runs per workgroup, one scan, placement (df_runs_kernel) ----
    const int nwg = (n + 255) / 256;
    const long long hsize = (long long)nlev * nwg;
    if (hsize + 1 >= (1ll << 30)) return give_up();
    Temp<unsigned> skey((size_t)nwg * 256);
    Temp<int> H((size_t)hsize + 1), S((size_t)hsize + 2), nch16((size_t)n), off((size_t)n + 1), tot(1), dstart((size_t)nlev + 1);
    zero_async(H, sizeof(int) * ((size_t)hsize + 1));
    hipLaunchKernelGGL(df_runs_kernel, dim3((unsigned)nwg), dim3(256), 0, st, n, nwg, (const int *)d_level, (const int *)M->diag.I, skey.p, H.p, nch16.p);
    ELAG_LAUNCH_CHECK();
    k::exclusive_scan(H, S, (int)hsize + 1);                                          // (S[nlev * nwg] = n)
    k::exclusive_scan(nch16, off, n, nullptr, tot);
    hipLaunchKernelGGL(df_level_starts_kernel, dim3((unsigned)((nlev + 256) / 256)), dim3(256), 0, st, nlev, nwg, (const int *)S.p, dstart.p);
    ELAG_LAUNCH_CHECK();
    std::vector<int> hstart((size_t)nlev + 1);
    stage_d2h(hstart.data(), dstart.p, sizeof(int) * ((size_t)nlev + 1));
    const int nent = fetch_int(tot);
    if (nent <= 0 || hstart[(size_t)nlev] != n) return give_up();                    // (the scan's total does not fit an int; or the runs do not add up)
    std::vector<int> base((size_t)nlev + 1, 0);
    long long npos = 0;
    for (int l = 0; l < nlev; l++) { base[(size_t)l] = (int)npos; npos += ((long long)(hstart[(size_t)l + 1] - hstart[(size_t)l]) + 3) / 4 * 4; }
    if (npos >= (1ll << 30)) return give_up();
    base[(size_t)nlev] = (int)npos;
    // ---- 3. the slots ----
    Temp<int> dbase((size_t)nlev + 1);
    stage_h2d(dbase.p, base.data(), sizeof(int) * ((size_t)nlev + 1));
    d_df_slots = dev_alloc<int>((size_t)npos * 4);
    hipLaunchKernelGGL(df_fill_slots_kernel, dim3((unsigned)std::min<long long>((npos + 255) / 256, 2048)), dim3(256), 0, st, (int)npos, reinterpret_cast<int4s *>(d_df_slots));
    hipLaunchKernelGGL(df_place_rows_kernel, dim3((unsigned)nwg), dim3(256), 0, st, n, nwg, (const unsigned *)skey.p, (const int *)S.p, (const int *)dstart.p,
                       (const int *)dbase.p, (const int *)off.p, (const int *)nch16.p, reinterpret_cast<int4s *>(d_df_slots));
    ELAG_LAUNCH_CHECK();
    // ---- 4. entries and flags; a row that is read without reading back sends the operator to the host path ----
    d_df_J = dev_alloc<unsigned>((size_t)nent);
    d_df_a = dev_alloc<double>((size_t)nent);
    Temp<int> nanti(1);
    zero_async(nanti, sizeof(int));
    hipLaunchKernelGGL(df_build_kernel, dim3((unsigned)((npos * kLPR + 255) / 256)), dim3(256), 0, st, (int)npos, reinterpret_cast<const int4s *>(d_df_slots),
                       (const int *)M->diag.I, (const int *)M->diag.J, (const double *)M->diag.a, (const int *)d_level, d_df_J, d_df_a, nanti.p);
    ELAG_LAUNCH_CHECK();
    const int na = fetch_int(nanti);
    if (na != 0) return give_up();
    dev_free(d_level);
    df_npos = (int)npos;
    num_levels = nlev;
    finish_dataflow(nlev);
    dataflow = true;
    if (getenv("ELAG_HIP_SWEEP_VERBOSE"))
        fprintf(stderr, "[elag_hip] sweep smoother type %d: n=%d, %d dependency levels; dataflow sweep: one launch per symmetric sweep (schedule built on the device)\n", type, n, nlev);
    return true;
}

void SweepSmoother::launch_dataflow(const double *f, double *u)
{
    static long long ticks = 0;
    if (ticks == 0) { const char *e = getenv("ELAG_HIP_SWEEP_DATAFLOW_MS"); ticks = std::max(1ll, (long long)((e && atof(e) > 0 ? atof(e) : 250.0) * 1.0e5)); }   // (100 MHz)
    const bool ho = M->offd.nnz > 0;
    DfArgs A;
    A.slots = reinterpret_cast<const int4s *>(d_df_slots);
    A.J = d_df_J; A.a = d_df_a;
    A.oI = ho ? M->offd.I : nullptr; A.oJ = ho ? M->offd.J : nullptr; A.oa = ho ? M->offd.a : nullptr; A.ug = M->d_ghost;
    A.f = f; A.d = (type == ELAG_HIP_SMOOTHER_GS) ? nullptr : d_l1; A.omega = omega;
    A.u = u; A.gran = d_df_gran; A.state = d_df_state; A.host_err = ctx().h_dev_err;
    A.n = M->nrows(); A.npos = df_npos; A.ticks = ticks;
    { const char *e = getenv("ELAG_HIP_SWEEP_DATAFLOW_STALL_ROW"); A.stall_row = e ? atoi(e) : -1; }
    if (df_xcd) {
        if (type == ELAG_HIP_SMOOTHER_GS) hipLaunchKernelGGL((sweep_dataflow_kernel<1, true>), dim3((unsigned)df_grid), dim3(256), 0, ctx().stream, A);
        else hipLaunchKernelGGL((sweep_dataflow_kernel<0, true>), dim3((unsigned)df_grid), dim3(256), 0, ctx().stream, A);
    } else {
        // (the whole chip keeps the static deal: one tile counter for 1 024 wavefronts on eight XCDs was measured at 532 us against 262 for the
        // cube456 x3 level-0 sweep -- the counter is the chip's slowest word)
        if (type == ELAG_HIP_SMOOTHER_GS) hipLaunchKernelGGL((sweep_dataflow_kernel<1, false>), dim3((unsigned)df_grid), dim3(256), 0, ctx().stream, A);
        else hipLaunchKernelGGL((sweep_dataflow_kernel<0, false>), dim3((unsigned)df_grid), dim3(256), 0, ctx().stream, A);
    }
    ELAG_LAUNCH_CHECK();
}

namespace {
// One dependency level of a sweep on m right-hand sides: a thread per (row of the level, column).  MODE as in the sweep kernels (0: types
// 2 / 4, 1: type 6); F == nullptr: the identity; Xs: the frozen copy the off-block terms read (virtual ranks), ghost_col: their rows in it.
template <int MODE>
__global__ void sweep_dense_level_kernel(int r0, int r1, const int *__restrict__ rows, const int *__restrict__ I, const int *__restrict__ J,
                                         const double *__restrict__ a, const int *__restrict__ oI, const int *__restrict__ oJ, const double *__restrict__ oa,
                                         const int *__restrict__ ghost_col, const double *__restrict__ F, const double *__restrict__ d, double omega,
                                         const double *__restrict__ Xs, double *X, int m)
{
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int ri = r0 + (int)(id / m), c = (int)(id % m);
    if (ri >= r1) return;
    const int row = rows[ri];
    double res = F ? F[(size_t)row * m + c] : (row == c ? 1.0 : 0.0), aii = 0.0;
    bool have = false;
    for (int k = I[row]; k < I[row + 1]; k++) {
        const int j = J[k];
        if (MODE == 1 && j == row && !have) { aii = a[k]; have = true; continue; }
        res -= a[k] * X[(size_t)j * m + c];
    }
    if (oI)
        for (int k = oI[row]; k < oI[row + 1]; k++) res -= oa[k] * Xs[(size_t)ghost_col[oJ[k]] * m + c];
    if (MODE == 0) { if (d[row] != 0.0) X[(size_t)row * m + c] += omega * res / d[row]; }
    else if (have && aii != 0.0) X[(size_t)row * m + c] = res / aii;
}

// The same sweeps with the COLUMNS of X dealt to the workgroups: the columns are independent systems, so a workgroup that owns a strip of 32 of
// them walks all dependency levels of all sweeps by itself with a workgroup barrier between levels -- one launch where the level kernel above takes
// two per level and sweep (368 launches of ~16 us for the 5 739 x 5 739 block of cube456 x3's level 1: 5.9 of a 13 ms Gauss-Seidel hierarchy
// build).  Same operations per entry, same order inside a row (operators without ghosts; 64 rows of a level at a time).
constexpr int kStripCols = 32, kStripThreads = 1024;     // (32 columns: 256-byte pieces of a row of X; 16 read the 263 MB block 6 % slower, 64 leave too few workgroups: 9.5 against 6.1 ms)
template <int MODE>
__global__ __launch_bounds__(kStripThreads) void sweep_dense_strip_kernel(int nlev, const int *__restrict__ lev_ptr, const int *__restrict__ rows,
                                                                          const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                                                          const double *__restrict__ F, const double *__restrict__ d, double omega, double *X, int m,
                                                                          int sweeps)
{
    constexpr int kSlots = kStripThreads / kStripCols;
    const int c = blockIdx.x * kStripCols + (int)(threadIdx.x % kStripCols), rs = (int)(threadIdx.x / kStripCols);
    const bool on = c < m;
    for (int sw = 0; sw < sweeps; sw++)
        for (int pass = 0; pass < 2; pass++)
            for (int q = 0; q < nlev; q++) {
                const int l = pass == 0 ? q : nlev - 1 - q;
                const int r0 = lev_ptr[l], r1 = lev_ptr[l + 1];
                for (int ri = r0 + rs; ri < r1; ri += kSlots) {
                    if (!on) continue;
                    const int row = rows[ri];
                    double res = F ? F[(size_t)row * m + c] : (row == c ? 1.0 : 0.0), aii = 0.0;
                    bool have = false;
                    // eight entries of the row at a time: their columns and values, then their X entries, requested together (entry by entry a row
                    // is a chain of two dependent loads per entry); the sum is taken in the row's order as before.  (What is left is bandwidth: a sweep reads
                    // nnz(A) x m entries of X -- 3.6 GB per direction for the 5 739 x 5 739 block, whose 263 MB fit no cache -- at ~3 TB/s in 128 B
                    // pieces.  Requesting the next level's row ahead and keeping the schedule in LDS were tried: slower / no change.)
                    const int ke = I[row + 1];
                    for (int k = I[row]; k < ke; k += 8) {
                        int jj[8];
                        double aa[8], xx[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { const bool in = k + u < ke; jj[u] = in ? J[k + u] : row; aa[u] = in ? a[k + u] : 0.0; }
#pragma unroll
                        for (int u = 0; u < 8; u++) xx[u] = X[(size_t)jj[u] * m + c];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            if (k + u >= ke) continue;
                            if (MODE == 1 && jj[u] == row && !have) { aii = aa[u]; have = true; continue; }
                            res -= aa[u] * xx[u];
                        }
                    }
                    if (MODE == 0) { if (d[row] != 0.0) X[(size_t)row * m + c] += omega * res / d[row]; }
                    else if (have && aii != 0.0) X[(size_t)row * m + c] = res / aii;
                }
                __syncthreads();          // (the level's rows of this strip are written before the next level reads them: columns never cross strips)
            }
}
}  // namespace

bool SweepSmoother::apply_dense(const double *F, double *X, int m, bool zero_guess)
{
    if (!host_ready) setup_host(false);          // (the rows by dependency level: a schedule built on the device does not keep them on the host)
    if (type == ELAG_HIP_SMOOTHER_KACZMARZ || sor || h_lev_ptr.empty()) return false;
    const int n = M->nrows();
    if (n == 0 || m == 0) return true;
    hipStream_t st = ctx().stream;
    if (zero_guess) zero_async(X, sizeof(double) * (size_t)n * m);
    const bool ho = M->offd.nnz > 0;
    ELAG_REQUIRE(!ho || As, "internal: apply_dense on an operator with real ghosts");
    Temp<double> snap(ho ? (size_t)n * m : 1);
    const int nlev = (int)h_lev_ptr.size() - 1;
    {
        const char *e = getenv("ELAG_HIP_DENSE_STRIPS");              // (read at every call: a test compares the two routes)
        if (!ho && !(e && !atoi(e)) && m >= 64) {
            Temp<int> dlp((size_t)nlev + 1);
            stage_h2d(dlp.p, h_lev_ptr.data(), sizeof(int) * ((size_t)nlev + 1));
            const unsigned g = (unsigned)((m + kStripCols - 1) / kStripCols);
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(g), dim3(kStripThreads), 0, st, nlev, (const int *)dlp.p, (const int *)d_lev_rows, (const int *)M->diag.I,
                                   (const int *)M->diag.J, (const double *)M->diag.a, F, (const double *)d_l1, omega, X, m, sweeps);
            };
            if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_dense_strip_kernel<1>);
            else go(sweep_dense_strip_kernel<0>);
            ELAG_LAUNCH_CHECK();
            return true;
        }
    }
    auto level = [&](int l) {
        const int r0 = h_lev_ptr[(size_t)l], r1 = h_lev_ptr[(size_t)l + 1];
        if (r1 == r0) return;
        const long long tot = (long long)(r1 - r0) * m;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, r0, r1, (const int *)d_lev_rows, (const int *)M->diag.I,
                               (const int *)M->diag.J, (const double *)M->diag.a, ho ? (const int *)M->offd.I : nullptr, ho ? (const int *)M->offd.J : nullptr,
                               ho ? (const double *)M->offd.a : nullptr, (const int *)d_self_ghost, F, (const double *)d_l1, omega, (const double *)snap.p, X, m);
        };
        if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_dense_level_kernel<1>);
        else go(sweep_dense_level_kernel<0>);
    };
    for (int sw = 0; sw < sweeps; sw++) {
        if (ho) dev_copy(snap.p, X, sizeof(double) * (size_t)n * m, st);   // the other blocks' values, frozen
        for (int l = 0; l < nlev; l++) level(l);
        for (int l = nlev - 1; l >= 0; l--) level(l);
    }
    ELAG_LAUNCH_CHECK();
    return true;
}

void SweepSmoother::launch_segment(const Segment &s, int dir, const double *f, double *u, bool sym)
{
    hipStream_t st = ctx().stream;
    const bool ho = M->offd.nnz > 0;
    const int *oI = ho ? M->offd.I : nullptr, *oJ = ho ? M->offd.J : nullptr;
    const double *oa = ho ? M->offd.a : nullptr;
    const int4s *sched = reinterpret_cast<const int4s *>(d_rows);
    const double *dd = (type == ELAG_HIP_SMOOTHER_GS) ? nullptr : d_l1;
    const int *lev = d_lev_ptr, *Jd = M->diag.J;
    const double *ad = M->diag.a, *ug = M->d_ghost;
    if (s.max_rows < 0) {
        // virtual ranks: a workgroup per block, the symmetric sweep in one launch
        const int n = M->nrows();
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kChainBlock), 0, st, lev, lev + n_steps + 1, 0, 0, +1, sched, Jd, ad, oI, oJ, oa, f, dd, ug, omega, u,
                               n, 1, (const int *)d_blk_first, (const int *)d_blk_count);
        };
        if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_chain_kernel<1, false>);
        else go(sweep_chain_kernel<0, false>);
        ELAG_LAUNCH_CHECK();
        return;
    }
    if (s.max_rows > 0) {
        // one big level (all its steps as one range): enough workgroups for one row per lane group
        const unsigned grid = (unsigned)(((long long)s.max_rows * kLPR + kWideBlock - 1) / kWideBlock);
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(kWideBlock), 0, st, s.row_begin, s.row_end, sched, (const int *)d_sJ, (const double *)d_sa, Jd, ad, oI, oJ, oa, f,
                               dd, ug, omega, u);
        };
        if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_levels_kernel<1>);
        else if (type == ELAG_HIP_SMOOTHER_KACZMARZ) go(sweep_levels_kernel<2>);
        else go(sweep_levels_kernel<0>);
    } else {
        const int s_first = dir > 0 ? s.first : s.first + s.count - 1;
        const int n = M->nrows();
        const size_t lds = lds_u ? sizeof(double) * (size_t)n : 0;
        auto go = [&](auto kern) {
            if (lds > (size_t)(48 << 10)) {
                // (beyond the default dynamic LDS limit: raise it once per kernel)
                static thread_local std::vector<const void *> raised;
                const void *fn = reinterpret_cast<const void *>(kern);
                if (std::find(raised.begin(), raised.end(), fn) == raised.end()) {
                    ELAG_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kLdsRows)));
                    raised.push_back(fn);
                }
            }
            hipLaunchKernelGGL(kern, dim3(1), dim3(kChainBlock), lds, st, lev, lev + n_steps + 1, s_first, s.count, dir, sched, Jd, ad, oI, oJ, oa, f, dd, ug, omega, u,
                               n, sym ? 1 : 0, (const int *)nullptr, (const int *)nullptr);
        };
        if (lds_u) {
            if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_chain_kernel<1, true>);
            else if (type == ELAG_HIP_SMOOTHER_KACZMARZ) go(sweep_chain_kernel<2, true>);
            else go(sweep_chain_kernel<0, true>);
        } else {
            if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_chain_kernel<1, false>);
            else if (type == ELAG_HIP_SMOOTHER_KACZMARZ) go(sweep_chain_kernel<2, false>);
            else go(sweep_chain_kernel<0, false>);
        }
    }
    ELAG_LAUNCH_CHECK();
}

void SweepSmoother::apply(const double *f, double *&cur, double *&, bool zero_guess, bool /*transpose: the symmetric sweep is self-adjoint*/)
{
    const int n = M->nrows();
    if (zero_guess) k::vec_fill(n, 0.0, cur);
    for (int sw = 0; sw < sweeps; sw++) {
        // "Copy off-diag values of u to the current processor": once per symmetric sweep
        if (As) k::vec_gather(As->offd.ncols, d_self_ghost, cur, As->d_ghost);        // (virtual ranks: the other blocks' values, frozen here)
        else if (M->needs_exchange()) par_halo_exchange(*M, cur);
        if (sor) { sweep_sor(f, cur); continue; }
        if (dataflow) { launch_dataflow(f, cur); continue; }
        if (!host_ready) {
            if (stream_is_capturing()) fail(ELAG_HIP_ERR_STATE, "internal: the launch-per-level plan of a Gauss-Seidel smoother is missing inside a graph capture");
            setup_host(false);
        }
        if (segments.size() == 1 && segments[0].max_rows < 0) { launch_segment(segments[0], +1, f, cur, true); continue; }
        if (lds_u && segments.size() == 1 && segments[0].max_rows == 0) {
            launch_segment(segments[0], +1, f, cur, true);        // forward then backward in one launch, u in LDS throughout
            continue;
        }
        for (size_t s = 0; s < segments.size(); s++) launch_segment(segments[s], +1, f, cur);
        for (size_t s = segments.size(); s-- > 0;) launch_segment(segments[s], -1, f, cur);
    }
}

void SweepSmoother::sweep_sor(const double *f, double *u)
{
    const int n = M->nrows(), nlev = (int)h_lev_ptr.size() - 1;
    if (n == 0) return;
    hipStream_t st = ctx().stream;
    const bool ho = M->offd.nnz > 0;
    dev_copy(d_v, u, sizeof(double) * (size_t)n, st);          // (d_v: allocated by setup -- a cycle may be under capture here)
    auto level = [&](int l, int dir) {
        const int r0 = h_lev_ptr[(size_t)l], r1 = h_lev_ptr[(size_t)l + 1];
        if (r1 == r0) return;
        auto go = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, st, r0, r1, dir, (const int *)d_lev_rows, (const int *)M->diag.I,
                               (const int *)M->diag.J, (const double *)M->diag.a, ho ? (const int *)M->offd.I : nullptr, ho ? (const int *)M->offd.J : nullptr,
                               ho ? (const double *)M->offd.a : nullptr, f, (const double *)d_l1, (const double *)M->d_ghost, weight, omega, (const double *)d_v, u);
        };
        if (type == ELAG_HIP_SMOOTHER_GS) go(sweep_sor_level_kernel<1>);
        else go(sweep_sor_level_kernel<0>);
    };
    for (int l = 0; l < nlev; l++) level(l, +1);
    for (int l = nlev - 1; l >= 0; l--) level(l, -1);
    ELAG_LAUNCH_CHECK();
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
