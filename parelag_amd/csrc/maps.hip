// Host-side integer maps of the agglomeration and the integer / pattern ParCSR helpers of
// src/hypreExtension.  The three index-map routines never touch the GPU.
#include <algorithm>
#include <numeric>

#include "comm.hpp"
#include "parcsr.hpp"
#include "staging.hpp"

using namespace elag;

namespace {

// union-find with path halving
struct DisjointSets {
    std::vector<int> parent;
    explicit DisjointSets(int n) : parent((size_t)n) { std::iota(parent.begin(), parent.end(), 0); }
    int find(int x)
    {
        while (parent[(size_t)x] != x) { parent[(size_t)x] = parent[(size_t)parent[(size_t)x]]; x = parent[(size_t)x]; }
        return x;
    }
    void unite(int a, int b)
    {
        a = find(a); b = find(b);
        if (a == b) return;
        if (a < b) parent[(size_t)b] = a; else parent[(size_t)a] = b;  // the smaller index stays the representative
    }
};

__global__ void boolint_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const int *__restrict__ oI,
                               const int *__restrict__ oJ, const int *__restrict__ x, const int *__restrict__ xg, int alpha,
                               int beta, int *__restrict__ y)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    // hypre_CSRMatrixMatvecBoolInt: y = (beta/alpha) y; y += sum x[j]; y *= alpha   (integer arithmetic, :60-133)
    int v = y[row];
    if (alpha == 0) { y[row] = v * beta; return; }
    const int temp = beta / alpha;
    v = (temp == 0) ? 0 : v * temp;
    int t = 0;
    for (int k = I[row]; k < I[row + 1]; k++) t += x[J[k]];
    v += t;
    v *= alpha;
    if (oI) {  // offd pass with beta = 1 (:196): y = (1/alpha) y (integer division, as the reference does); += ; *= alpha
        int t2 = 0;
        for (int k = oI[row]; k < oI[row + 1]; k++) t2 += xg[oJ[k]];
        const int temp2 = 1 / alpha;
        if (temp2 != 1) v = (temp2 == 0) ? 0 : v * temp2;
        v += t2;
        v *= alpha;
    }
    y[row] = v;
}

__global__ void gather_int_kernel(int n, const int *__restrict__ idx, const int *__restrict__ x, int *__restrict__ buf)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) buf[i] = x[idx[i]];
}

__global__ void keep_equal_kernel(long long nnz, double *__restrict__ a, const double *__restrict__ b)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long long)gridDim.x * blockDim.x)
        a[k] = (fabs(a[k] - b[k]) < 1e-9) ? 1.0 : 0.0;
}

__global__ void pattern_differs_kernel(long long n, const int *__restrict__ a, const int *__restrict__ b, int *__restrict__ flag)
{
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x)
        if (a[k] != b[k]) atomicOr(flag, 1);
}

bool same_pattern(const DevCSR &A, const DevCSR &B)
{
    if (A.nrows != B.nrows || A.ncols != B.ncols || A.nnz != B.nnz) return false;
    Temp<int> flag(1);
    hipStream_t st = ctx().stream;
    zero_async(flag, sizeof(int));
    hipLaunchKernelGGL(pattern_differs_kernel, dim3(256), dim3(256), 0, st, (long long)A.nrows + 1, (const int *)A.I, (const int *)B.I, flag.p);
    if (A.nnz) hipLaunchKernelGGL(pattern_differs_kernel, dim3(1024), dim3(256), 0, st, (long long)A.nnz, (const int *)A.J, (const int *)B.J, flag.p);
    ELAG_LAUNCH_CHECK();
    int h = 0;
    stage_d2h(&h, flag.p, sizeof(int));
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    return h == 0;
}

}  // namespace

extern "C" {

int elag_hip_partition_refined(int nElements, int nParts, int nDim, int legacy, int *p)
{
    return guarded([&] {
        ELAG_REQUIRE(p && nElements > 0 && nParts > 0 && nDim >= 1 && nDim <= 3, "bad argument");
        ELAG_REQUIRE(nElements > nParts && nElements % nParts == 0, "nElements (%d) must be a proper multiple of nParts (%d)", nElements, nParts);
        const int nsplits = 1 << nDim, factor = nElements / nParts;
        ELAG_REQUIRE(factor % nsplits == 0, "coarsening factor %d is not a multiple of 2^nDim = %d", factor, nsplits);
        if (!legacy) {
            for (int j = 0; j < nElements; j++) p[j] = j / factor;   // children of element i are i*factor .. (i+1)*factor-1
            return;
        }
        // refinement levels appended pass by pass: after each pass the first `len` entries are known and every known
        // element contributes (nsplits - 1) appended children, in order
        for (int i = 0; i < nParts; i++) p[i] = i;
        int len = nParts;
        while (len < nElements) {
            int w = len;
            for (int i = 0; i < len; i++)
                for (int c = 1; c < nsplits; c++) p[w++] = p[i];
            len = w;
        }
    });
}

int elag_hip_connected_components(int n, int *part, const int *ci, const int *cj, int *num_components)
{
    return guarded([&] {
        ELAG_REQUIRE(n >= 0 && (n == 0 || (part && ci && cj)), "bad argument");
        if (n == 0) { if (num_components) *num_components = 0; return; }
        int nparts = 0;
        // nodes with a negative id belong to no part: the reference's sweep skips them (connectedComponents.cpp:50-51,
        // 62-63, like transposePartitioning does); they keep their id
        for (int i = 0; i < n; i++) nparts = std::max(nparts, part[i] + 1);
        // components = classes of "same part and adjacent"; representative = smallest node of the class
        DisjointSets ds(n);
        for (int i = 0; i < n; i++)
            for (int k = ci[i]; k < ci[i + 1]; k++) {
                const int j = cj[k];
                ELAG_REQUIRE(j >= 0 && j < n, "connectivity column %d out of range", j);
                if (part[i] >= 0 && part[j] == part[i]) ds.unite(i, j);
            }
        // The reference numbers the components of a part in the order in which a sweep over the nodes first meets them,
        // i.e. by their smallest node; ids of part q start after all components of parts < q.
        std::vector<int> ncomp((size_t)nparts + 1, 0), local((size_t)n, -1);
        for (int i = 0; i < n; i++)
            if (part[i] >= 0 && ds.find(i) == i) local[(size_t)i] = ncomp[(size_t)part[i] + 1]++;
        for (int q = 0; q < nparts; q++) ncomp[(size_t)q + 1] += ncomp[(size_t)q];
        const std::vector<int> old(part, part + n);   // representatives are rewritten before their members read them
        for (int i = 0; i < n; i++) {
            if (old[(size_t)i] < 0) continue;
            const int r = ds.find(i);
            part[i] = ncomp[(size_t)old[(size_t)r]] + local[(size_t)r];
        }
        if (num_components) *num_components = ncomp[(size_t)nparts];
    });
}

int elag_hip_transpose_partitioning(int n, const int *part, int nparts, int *io, int *jo)
{
    return guarded([&] {
        ELAG_REQUIRE(n >= 0 && nparts >= 0 && io && (n == 0 || part), "bad argument");
        std::fill(io, io + nparts + 1, 0);
        for (int i = 0; i < n; i++) {
            ELAG_REQUIRE(part[i] >= -1 && part[i] < nparts, "part id %d out of range at element %d", part[i], i);
            if (part[i] >= 0) io[part[i] + 1]++;
        }
        std::partial_sum(io, io + nparts + 1, io);
        std::vector<int> cursor(io, io + nparts);
        for (int i = 0; i < n; i++)
            if (part[i] >= 0) jo[cursor[(size_t)part[i]]++] = i;
    });
}

int elag_hip_parcsr_spmv_boolint(elag_hip_parcsr_t A, int alpha, const int *x, int beta, int *y, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && x && y && x != y, "bad argument");
        check_loc(loc);
        hipStream_t st = ctx().stream;
        const int n = A->nrows(), nc = A->ncols();
        const int *dx = x; int *dy = y;
        int *tx = nullptr, *ty = nullptr;
        if (loc == ELAG_HIP_HOST) {
            tx = dev_from_host(x, (size_t)nc);
            ty = dev_from_host(y, (size_t)n);
            dx = tx; dy = ty;
        }
        int *ghost = nullptr, *sbuf = nullptr;
        const bool halo = A->needs_exchange();
        if (halo) {
            ghost = dev_alloc<int>((size_t)A->offd.ncols);
            const int stn = A->send_total();
            sbuf = dev_alloc<int>((size_t)stn);
            if (stn) hipLaunchKernelGGL(gather_int_kernel, dim3((unsigned)std::min((stn + 255) / 256, 1024)), dim3(256), 0, st, stn,
                                        (const int *)A->d_send_elmts, dx, sbuf);
            Context &c = ctx();
            ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, st));
            ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
            comm_neighbor_exchange(sbuf, (int)A->send_procs.size(), A->send_procs.data(), A->send_starts.data(), ghost,
                                   (int)A->recv_procs.size(), A->recv_procs.data(), A->recv_starts.data(), sizeof(int), c.comm_stream);
            ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
            ELAG_HIP_CHECK(hipStreamWaitEvent(st, c.ev_halo, 0));
        }
        if (n)
            hipLaunchKernelGGL(boolint_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const int *)A->diag.I,
                               (const int *)A->diag.J, A->offd.ncols > 0 ? (const int *)A->offd.I : nullptr, A->offd.ncols > 0 ? (const int *)A->offd.J : nullptr, dx,
                               (const int *)ghost, alpha, beta, dy);
        ELAG_LAUNCH_CHECK();
        if (loc == ELAG_HIP_HOST && n) stage_d2h(y, ty, sizeof(int) * (size_t)n);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        dev_free(tx); dev_free(ty); dev_free(ghost); dev_free(sbuf);
    });
}

int elag_hip_parcsr_keep_equal_entries(elag_hip_parcsr_t A, elag_hip_parcsr_t B)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && B, "null matrix");
        ELAG_REQUIRE(same_pattern(A->diag, B->diag) && same_pattern(A->offd, B->offd), "KeepEqualEntries: A and B must have the same sparsity pattern");
        hipStream_t st = ctx().stream;
        A->diag.release_sign(); A->offd.release_sign();
        if (A->diag.nnz) hipLaunchKernelGGL(keep_equal_kernel, dim3(1024), dim3(256), 0, st, (long long)A->diag.nnz, A->diag.a, (const double *)B->diag.a);
        if (A->offd.nnz) hipLaunchKernelGGL(keep_equal_kernel, dim3(1024), dim3(256), 0, st, (long long)A->offd.nnz, A->offd.a, (const double *)B->offd.a);
        ELAG_LAUNCH_CHECK();
        // hypre_ParCSRMatrixDeleteZeros(A, 1e-9)
        DevCSR d;
        k::csr_delete_zeros(A->diag, 1e-9, d);
        A->diag = std::move(d);
        if (A->offd.nnz) {
            DevCSR o;
            k::csr_delete_zeros(A->offd, 1e-9, o);
            const int nco = A->offd.ncols;
            A->offd = std::move(o);
            A->offd.ncols = nco;
        }
        A->diagT.reset(); A->offdT.reset();
        A->finish_setup();
    });
}

}  // extern "C"
