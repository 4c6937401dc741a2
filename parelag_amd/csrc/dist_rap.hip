// Distributed triple product  C = RT^T A P  on ParCSR operands (hypre_BoomerAMGBuildCoarseOperator):
//   1. the rows of P that A's ghost columns refer to are fetched from their owners along A's comm package
//      (one variable-length exchange: row lengths, then columns and values)                     [P_ext]
//   2. AP = [A_diag | A_offd] * [P ; P_ext] with GLOBAL coarse column ids (ESC SpGEMM)
//   3. C_all = [RT_diag | RT_offd]^T * AP : rows of this rank's coarse dofs and of the ghost coarse dofs of RT
//   4. the ghost rows are contributions to coarse rows owned elsewhere: sent back along RT's comm package (reverse
//      direction) and added in through the deterministic scatter matrix                        [RAP_ext]
//   5. the result is split into diag / offd by the coarse column partition, col_map_offd is built and the comm
//      package of C is negotiated (what hypre_MatvecCommPkgCreate does).
// Steps 1 and 4 vanish on a single rank; IgnoreNonLocalRange (MatrixUtils.cpp:878-916) is the same routine with
// RT's offd emptied.  All exchanges go through comm_neighbor_exchange / comm_allgatherv, i.e. RCCL across GPUs or the
// in-process hub in the single-GPU tests.
#include <algorithm>
#include <numeric>

#include "comm.hpp"
#include "parcsr.hpp"

namespace elag {

namespace {

constexpr int kB = 256;
inline unsigned g1(long long n) { return (unsigned)std::max<long long>(1, (n + kB - 1) / kB); }

__global__ void concat_len_kernel(int n, const int *__restrict__ XI, const int *__restrict__ YI, int *__restrict__ len)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) len[r] = (XI[r + 1] - XI[r]) + (YI ? YI[r + 1] - YI[r] : 0);
}

// out row r = X row r (cols: mapX ? mapX[j] : j + offX) followed by Y row r (same with mapY / offY)
__global__ void concat_fill_kernel(int n, const int *__restrict__ XI, const int *__restrict__ XJ, const double *__restrict__ Xa,
                                   const int *__restrict__ mapX, int offX, const int *__restrict__ YI, const int *__restrict__ YJ,
                                   const double *__restrict__ Ya, const int *__restrict__ mapY, int offY, const int *__restrict__ OI,
                                   int *__restrict__ OJ, double *__restrict__ Oa)
{
    const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int lane = threadIdx.x & 7;
    if (r >= n) return;
    int d = OI[r];
    const int xs = XI[r], xl = XI[r + 1] - xs;
    for (int q = lane; q < xl; q += 8) { const int j = XJ[xs + q]; OJ[d + q] = mapX ? mapX[j] : j + offX; Oa[d + q] = Xa[xs + q]; }
    d += xl;
    if (YI) {
        const int ys = YI[r], yl = YI[r + 1] - ys;
        for (int q = lane; q < yl; q += 8) { const int j = YJ[ys + q]; OJ[d + q] = mapY ? mapY[j] : j + offY; Oa[d + q] = Ya[ys + q]; }
    }
}

void concat_rows(const DevCSR &X, const int *mapX, int offX, const DevCSR *Y, const int *mapY, int offY, int ncols_out, bool sort,
                 DevCSR &out)
{
    hipStream_t st = ctx().stream;
    const int n = X.nrows;
    out.release();
    out.nrows = n; out.ncols = ncols_out;
    out.I = dev_alloc<int>((size_t)n + 1);
    {
        Temp<int> len((size_t)n);
        if (n) hipLaunchKernelGGL(concat_len_kernel, dim3(g1(n)), dim3(kB), 0, st, n, (const int *)X.I, Y ? (const int *)Y->I : nullptr, len.p);
        k::exclusive_scan(len, out.I, n);
    }
    const int64_t nnz = X.nnz + (Y ? Y->nnz : 0);
    out.alloc_entries(nnz);
    if (n && nnz)
        hipLaunchKernelGGL(concat_fill_kernel, dim3(g1((long long)n * 8)), dim3(kB), 0, st, n, (const int *)X.I, (const int *)X.J,
                           (const double *)X.a, mapX, offX, Y ? (const int *)Y->I : nullptr, Y ? (const int *)Y->J : nullptr,
                           Y ? (const double *)Y->a : nullptr, mapY, offY, (const int *)out.I, out.J, out.a);
    ELAG_LAUNCH_CHECK();
    if (sort && nnz) k::csr_sort_rows(out);
    else out.choose_spmv_shape();
}

__global__ void add_offset_kernel(int n, const int *__restrict__ in, int off, int *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + off;
}

// rows of X followed by rows of Y
void vstack(const DevCSR &X, const DevCSR &Y, DevCSR &out)
{
    hipStream_t st = ctx().stream;
    out.alloc(X.nrows + Y.nrows, X.ncols, X.nnz + Y.nnz);
    dev_copy(out.I, X.I, sizeof(int) * ((size_t)X.nrows + 1), st);
    if (Y.nrows)
        hipLaunchKernelGGL(add_offset_kernel, dim3(g1(Y.nrows)), dim3(kB), 0, st, Y.nrows, (const int *)(Y.I + 1), (int)X.nnz, out.I + X.nrows + 1);
    if (X.nnz) {
        dev_copy(out.J, X.J, sizeof(int) * (size_t)X.nnz, st);
        dev_copy(out.a, X.a, sizeof(double) * (size_t)X.nnz, st);
    }
    if (Y.nnz) {
        dev_copy(out.J + X.nnz, Y.J, sizeof(int) * (size_t)Y.nnz, st);
        dev_copy(out.a + X.nnz, Y.a, sizeof(double) * (size_t)Y.nnz, st);
    }
    ELAG_LAUNCH_CHECK();
    out.choose_spmv_shape();
}

// rows [r0, r1) of X
void slice_rows(const DevCSR &X, int r0, int r1, DevCSR &out)
{
    hipStream_t st = ctx().stream;
    int b[2] = {0, 0};
    stage_d2h(&b[0], X.I + r0, sizeof(int));
    stage_d2h(&b[1], X.I + r1, sizeof(int));
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    const int n = r1 - r0, nnz = b[1] - b[0];
    out.alloc(n, X.ncols, nnz);
    hipLaunchKernelGGL(add_offset_kernel, dim3(g1(n + 1)), dim3(kB), 0, st, n + 1, (const int *)(X.I + r0), -b[0], out.I);
    if (nnz) {
        dev_copy(out.J, X.J + b[0], sizeof(int) * (size_t)nnz, st);
        dev_copy(out.a, X.a + b[0], sizeof(double) * (size_t)nnz, st);
    }
    ELAG_LAUNCH_CHECK();
    out.choose_spmv_shape();
}

// ---- variable-length row exchange -----------------------------------------------------------------------------------
__global__ void row_len_kernel(int n, const int *__restrict__ I, const int *__restrict__ idx, int *__restrict__ len)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { const int r = idx ? idx[k] : k; len[k] = I[r + 1] - I[r]; }
}

__global__ void row_pack_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a,
                                const int *__restrict__ idx, const int *__restrict__ off, int *__restrict__ OJ, double *__restrict__ Oa)
{
    const int k = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int lane = threadIdx.x & 7;
    if (k >= n) return;
    const int r = idx ? idx[k] : k;
    const int s = I[r], l = I[r + 1] - s, d = off[k];
    for (int q = lane; q < l; q += 8) { OJ[d + q] = J[s + q]; Oa[d + q] = a[s + q]; }
}

void exchange_sync(const void *sb, int ns, const int *sp, const int *ss, void *rb, int nr, const int *rp, const int *rs, size_t eb)
{
    // every RCCL call on the comm stream, fork/join with the compute stream; setup path, so wait for completion
    Context &c = ctx();
    ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));
    comm_neighbor_exchange(sb, ns, sp, ss, rb, nr, rp, rs, eb, c.comm_stream);
    ELAG_HIP_CHECK(hipStreamSynchronize(c.comm_stream));
}

// Send the rows src[idx[k]] (k in the neighbour's segment of send_starts) to send_procs, receive rows in the order of
// recv_starts from recv_procs.  Returns the received rows as a CSR with src.ncols columns.
void exchange_rows(const DevCSR &src, const int *d_idx, const std::vector<int> &send_procs, const std::vector<int> &send_starts,
                   const std::vector<int> &recv_procs, const std::vector<int> &recv_starts, DevCSR &out)
{
    hipStream_t st = ctx().stream;
    const int ns = send_starts.empty() ? 0 : send_starts.back();
    const int nr = recv_starts.empty() ? 0 : recv_starts.back();
    Temp<int> len_s((size_t)ns), len_r((size_t)nr), off_s((size_t)ns + 1);
    if (ns) hipLaunchKernelGGL(row_len_kernel, dim3(g1(ns)), dim3(kB), 0, st, ns, (const int *)src.I, d_idx, len_s.p);
    ELAG_LAUNCH_CHECK();
    exchange_sync(len_s, (int)send_procs.size(), send_procs.data(), send_starts.data(), len_r, (int)recv_procs.size(), recv_procs.data(),
                  recv_starts.data(), sizeof(int));
    std::vector<int> hs = host_from_dev((const int *)len_s.p, (size_t)ns), hr = host_from_dev((const int *)len_r.p, (size_t)nr);
    std::vector<int> snz(send_procs.size() + 1, 0), rnz(recv_procs.size() + 1, 0);
    for (size_t i = 0; i < send_procs.size(); i++)
        snz[i + 1] = snz[i] + std::accumulate(hs.begin() + send_starts[i], hs.begin() + send_starts[i + 1], 0);
    for (size_t i = 0; i < recv_procs.size(); i++)
        rnz[i + 1] = rnz[i] + std::accumulate(hr.begin() + recv_starts[i], hr.begin() + recv_starts[i + 1], 0);
    k::exclusive_scan(len_s, off_s, ns);
    Temp<int> sJ((size_t)snz.back());
    Temp<double> sa((size_t)snz.back());
    if (ns && snz.back())
        hipLaunchKernelGGL(row_pack_kernel, dim3(g1((long long)ns * 8)), dim3(kB), 0, st, ns, (const int *)src.I, (const int *)src.J,
                           (const double *)src.a, d_idx, (const int *)off_s.p, sJ.p, sa.p);
    ELAG_LAUNCH_CHECK();
    out.release();
    out.nrows = nr; out.ncols = src.ncols;
    out.I = dev_alloc<int>((size_t)nr + 1);
    k::exclusive_scan(len_r, out.I, nr);
    out.alloc_entries(rnz.back());
    exchange_sync(sJ, (int)send_procs.size(), send_procs.data(), snz.data(), out.J, (int)recv_procs.size(), recv_procs.data(), rnz.data(), sizeof(int));
    exchange_sync(sa, (int)send_procs.size(), send_procs.data(), snz.data(), out.a, (int)recv_procs.size(), recv_procs.data(), rnz.data(), sizeof(double));
    out.choose_spmv_shape();
}

// ---- split a global-column matrix into diag / offd ------------------------------------------------------------------
__global__ void split_count_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, int c0, int c1, int *__restrict__ nd,
                                   int *__restrict__ no)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int d = 0, o = 0;
    for (int k = I[r]; k < I[r + 1]; k++) { const int j = J[k]; if (j >= c0 && j < c1) d++; else o++; }
    nd[r] = d; no[r] = o;
}

__global__ void split_fill_kernel(int n, const int *__restrict__ I, const int *__restrict__ J, const double *__restrict__ a, int c0, int c1,
                                  const int *__restrict__ cmap, int ncm, const int *__restrict__ DI, int *__restrict__ DJ,
                                  double *__restrict__ Da, const int *__restrict__ OI, int *__restrict__ OJ, double *__restrict__ Oa)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int d = DI[r], o = OI[r];
    for (int k = I[r]; k < I[r + 1]; k++) {
        const int j = J[k];
        if (j >= c0 && j < c1) { DJ[d] = j - c0; Da[d] = a[k]; d++; }
        else {
            int lo = j;                                  // (without a map: the global id stays; split_remap_kernel follows)
            if (cmap) {
                int hi = ncm;  // position of j in the ascending col_map
                lo = 0;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (cmap[mid] < j) lo = mid + 1; else hi = mid; }
            }
            OJ[o] = lo; Oa[o] = a[k]; o++;
        }
    }
}

__global__ void split_remap_kernel(int nnz, int *__restrict__ OJ, const int *__restrict__ cmap, int ncm)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int j = OJ[k];
    int lo = 0, hi = ncm;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cmap[mid] < j) lo = mid + 1; else hi = mid; }
    OJ[k] = lo;
}

std::vector<double> allgather_scalar(double v)
{
    Context &c = ctx();
    const int nr = c.nranks;
    std::vector<double> out((size_t)nr, v);
    if (nr == 1) return out;
    Temp<double> dv(1), dall((size_t)nr);
    stage_h2d(dv.p, &v, sizeof(double));
    ELAG_HIP_CHECK(hipStreamSynchronize(c.stream));
    std::vector<int> counts((size_t)nr, 1), displs((size_t)nr);
    std::iota(displs.begin(), displs.end(), 0);
    comm_allgatherv(dv, dall, counts.data(), displs.data(), c.comm_stream);
    ELAG_HIP_CHECK(hipStreamSynchronize(c.comm_stream));
    out = host_from_dev((const double *)dall.p, (size_t)nr);
    return out;
}

}  // namespace

// Builds the ParCSR matrix (rows = this rank's rows, first_row given) from a CSR with GLOBAL column ids:
// diag / offd split by [first_col, first_col + ncols_local), col_map_offd, and the negotiated comm package.
ParCSR *parcsr_from_global_columns(const DevCSR &Cg, int64_t first_row, int64_t global_rows, int64_t first_col, int ncols_local,
                                   int64_t global_cols)
{
    Context &c = ctx();
    hipStream_t st = c.stream;
    const int n = Cg.nrows, nr = c.nranks, me = c.rank;
    const int c0 = (int)first_col, c1 = (int)first_col + ncols_local;
    // The split first, with the off-rank entries keeping their GLOBAL column ids; then col_map_offd = their ascending unique values (on the
    // host, like hypre -- but only the off-rank ids travel: round 3 pulled every column id of the operator to the host for this, 150 MB at
    // cube456 x5), and the off-rank block's ids become positions in that map.
    std::unique_ptr<ParCSR> M(new ParCSR());
    std::vector<int> cmap;
    int ncm = 0;
    {
        Temp<int> nd((size_t)n), no((size_t)n);
        if (n) hipLaunchKernelGGL(split_count_kernel, dim3(g1(n)), dim3(kB), 0, st, n, (const int *)Cg.I, (const int *)Cg.J, c0, c1, nd.p, no.p);
        ELAG_LAUNCH_CHECK();
        M->diag.nrows = n; M->diag.ncols = ncols_local; M->diag.I = dev_alloc<int>((size_t)n + 1);
        M->offd.nrows = n; M->offd.I = dev_alloc<int>((size_t)n + 1);
        k::exclusive_scan(nd, M->diag.I, n);
        k::exclusive_scan(no, M->offd.I, n);
        int tot[2] = {0, 0};
        stage_d2h(&tot[0], M->diag.I + n, sizeof(int));
        stage_d2h(&tot[1], M->offd.I + n, sizeof(int));
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        M->diag.alloc_entries(tot[0]);
        M->offd.alloc_entries(tot[1]);
        if (n) hipLaunchKernelGGL(split_fill_kernel, dim3(g1(n)), dim3(kB), 0, st, n, (const int *)Cg.I, (const int *)Cg.J, (const double *)Cg.a, c0,
                                  c1, (const int *)nullptr, 0, (const int *)M->diag.I, M->diag.J, M->diag.a, (const int *)M->offd.I, M->offd.J, M->offd.a);
        ELAG_LAUNCH_CHECK();
        if (tot[1]) {
            cmap = host_from_dev((const int *)M->offd.J, (size_t)tot[1]);
            std::sort(cmap.begin(), cmap.end());
            cmap.erase(std::unique(cmap.begin(), cmap.end()), cmap.end());
            ncm = (int)cmap.size();
            Temp<int> dcm((size_t)ncm);
            stage_h2d(dcm.p, cmap.data(), sizeof(int) * (size_t)ncm);
            hipLaunchKernelGGL(split_remap_kernel, dim3(g1(tot[1])), dim3(kB), 0, st, tot[1], M->offd.J, (const int *)dcm.p, ncm);
            ELAG_LAUNCH_CHECK();
        }
        M->offd.ncols = ncm;
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
    }
    M->diag.choose_spmv_shape();
    M->offd.choose_spmv_shape();
    M->col_map_offd.assign(cmap.begin(), cmap.end());
    M->first_row = first_row; M->first_col = first_col; M->global_rows = global_rows; M->global_cols = global_cols;
    M->send_starts.assign(1, 0);
    M->recv_starts.assign(1, 0);
    if (nr > 1) {
        // column partition of every rank, owners of my ghosts, and who needs what from me (hypre_MatvecCommPkgCreate)
        std::vector<double> starts = allgather_scalar((double)first_col);
        std::vector<int> need((size_t)nr, 0);
        for (int j : cmap) {
            int q = (int)(std::upper_bound(starts.begin(), starts.end(), (double)j) - starts.begin()) - 1;
            if (q < 0 || q >= nr || q == me) fail(ELAG_HIP_ERR_COMM, "ghost column %d has no owner", j);
            need[(size_t)q]++;
        }
        for (int q = 0; q < nr; q++)
            if (need[(size_t)q]) { M->recv_procs.push_back(q); M->recv_starts.push_back(M->recv_starts.back() + need[(size_t)q]); }
        // all-gather the need matrix (nr x nr): need_all[r * nr + q] = number of columns rank r wants from rank q
        Temp<double> dn((size_t)nr), dall((size_t)nr * nr);
        std::vector<double> needd(need.begin(), need.end());
        stage_h2d(dn.p, needd.data(), sizeof(double) * (size_t)nr);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        std::vector<int> counts((size_t)nr, nr), displs((size_t)nr);
        for (int r = 0; r < nr; r++) displs[(size_t)r] = r * nr;
        comm_allgatherv(dn, dall, counts.data(), displs.data(), c.comm_stream);
        ELAG_HIP_CHECK(hipStreamSynchronize(c.comm_stream));
        std::vector<double> need_all = host_from_dev((const double *)dall.p, (size_t)nr * nr);
        for (int r = 0; r < nr; r++) {
            const int cnt = (int)need_all[(size_t)r * nr + me];
            if (r != me && cnt) { M->send_procs.push_back(r); M->send_starts.push_back(M->send_starts.back() + cnt); }
        }
        // requests: I send the global ids I need to their owners and receive the ids others need from me
        const int st_total = M->send_starts.back();
        Temp<int> dreq((size_t)ncm), dgot((size_t)st_total);
        if (ncm) stage_h2d(dreq.p, cmap.data(), sizeof(int) * (size_t)ncm);
        exchange_sync(dreq, (int)M->recv_procs.size(), M->recv_procs.data(), M->recv_starts.data(), dgot, (int)M->send_procs.size(),
                      M->send_procs.data(), M->send_starts.data(), sizeof(int));
        std::vector<int> got = host_from_dev((const int *)dgot.p, (size_t)st_total);
        M->send_elmts.resize((size_t)st_total);
        for (int k = 0; k < st_total; k++) {
            const int loc = got[(size_t)k] - c0;
            if (loc < 0 || loc >= ncols_local) fail(ELAG_HIP_ERR_COMM, "rank %d was asked for column %d it does not own", me, got[(size_t)k]);
            M->send_elmts[(size_t)k] = loc;
        }
    } else if (ncm) {
        fail(ELAG_HIP_ERR_COMM, "single-rank product produced %d off-range columns", ncm);
    }
    M->finish_setup();
    return M.release();
}

// this rank's rows with GLOBAL column ids (hypre_MergeDiagAndOffd), columns ascending
static void to_global(ParCSR &M, DevCSR &out)
{
    hipStream_t st = ctx().stream;
    ELAG_REQUIRE(M.global_cols < (int64_t(1) << 31), "global sizes beyond 2^31 are not supported");
    std::vector<int> cm(M.col_map_offd.begin(), M.col_map_offd.end());
    Temp<int> dcm(cm.size());
    if (!cm.empty()) stage_h2d(dcm.p, cm.data(), sizeof(int) * cm.size());
    concat_rows(M.diag, nullptr, (int)M.first_col, M.offd.ncols ? &M.offd : nullptr, dcm.p, 0, (int)M.global_cols, true, out);
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
}

void parcsr_to_global(ParCSR &M, DevCSR &out) { to_global(M, out); }
void csr_hstack(const DevCSR &X, const int *mapX, int offX, const DevCSR *Y, const int *mapY, int offY, int ncols_out, bool sort, DevCSR &out)
{
    concat_rows(X, mapX, offX, Y, mapY, offY, ncols_out, sort, out);
}
void csr_vstack(const DevCSR &X, const DevCSR &Y, DevCSR &out) { vstack(X, Y, out); }
std::vector<double> comm_allgather_scalar(double v) { return allgather_scalar(v); }

ParCSR *parcsr_clone(ParCSR &M)
{
    std::unique_ptr<ParCSR> C(new ParCSR());
    k::csr_copy(M.diag, C->diag);
    if (M.offd.nnz) k::csr_copy(M.offd, C->offd);
    else C->offd.set_empty(M.nrows(), M.offd.ncols);
    C->col_map_offd = M.col_map_offd;
    C->first_row = M.first_row; C->first_col = M.first_col; C->global_rows = M.global_rows; C->global_cols = M.global_cols;
    C->send_procs = M.send_procs; C->send_starts = M.send_starts; C->send_elmts = M.send_elmts;
    C->recv_procs = M.recv_procs; C->recv_starts = M.recv_starts;
    C->finish_setup();
    return C.release();
}

// hypre_ParCSRMatrixAdd2 across ranks (parcsr-add.c:199-245: merge diag+offd, add, GenerateDiagAndOffd)
ParCSR *distributed_add(double a, ParCSR &A, double b, ParCSR &B)
{
    DevCSR Ag, Bg, Cg;
    to_global(A, Ag);
    to_global(B, Bg);
    k::csr_add(a, Ag, b, Bg, Cg);
    return parcsr_from_global_columns(Cg, A.first_row, A.global_rows, A.first_col, A.ncols(), A.global_cols);
}

// hypre_ParMatmul: C = A*B, rows of B for A's ghost columns fetched along A's comm package
ParCSR *distributed_matmul(ParCSR &A, ParCSR &B)
{
    const int g = A.offd.ncols;
    DevCSR Bg, Bext, Bst, Aloc, Cg;
    to_global(B, Bg);
    exchange_rows(Bg, A.d_send_elmts, A.send_procs, A.send_starts, A.recv_procs, A.recv_starts, Bext);
    vstack(Bg, Bext, Bst);
    concat_rows(A.diag, nullptr, 0, g ? &A.offd : nullptr, nullptr, A.ncols(), A.ncols() + g, false, Aloc);
    k::csr_spgemm(Aloc, Bst, Cg);
    return parcsr_from_global_columns(Cg, A.first_row, A.global_rows, B.first_col, B.ncols(), B.global_cols);
}

// hypre_ParCSRMatrixTranspose2 (par_Tmatmul.c:41-79): rows of A^T for this rank's columns; the transposed ghost
// columns are rows owned elsewhere and travel home along the reverse comm package
ParCSR *distributed_transpose(ParCSR &A)
{
    hipStream_t st = ctx().stream;
    ELAG_REQUIRE(A.global_rows < (int64_t(1) << 31), "global sizes beyond 2^31 are not supported");
    const int nc = A.ncols(), g = A.offd.ncols;
    DevCSR Aloc, Tall, Tg;
    concat_rows(A.diag, nullptr, 0, g ? &A.offd : nullptr, nullptr, nc, nc + g, false, Aloc);
    k::csr_transpose(Aloc, Tall);                       // (nc + g) x n, columns = local row ids
    // local row ids -> global row ids
    {
        DevCSR shifted;
        DevCSR *none = nullptr;
        concat_rows(Tall, nullptr, (int)A.first_row, none, nullptr, 0, (int)A.global_rows, false, shifted);
        Tall = std::move(shifted);
    }
    if (A.needs_exchange()) {
        DevCSR Town, Text, Trecv;
        slice_rows(Tall, 0, nc, Town);
        slice_rows(Tall, nc, nc + g, Text);
        exchange_rows(Text, nullptr, A.recv_procs, A.recv_starts, A.send_procs, A.send_starts, Trecv);
        if (A.send_total()) {
            A.ensure_transpose();
            DevCSR E;
            k::csr_spgemm(*A.scatT, Trecv, E);
            k::csr_add(1.0, Town, 1.0, E, Tg);
        } else {
            Tg = std::move(Town);
        }
    } else {
        Tg = std::move(Tall);
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    return parcsr_from_global_columns(Tg, A.first_col, A.global_cols, A.first_row, A.nrows(), A.global_rows);
}

// local part of || A - B ||_max for hypre_ParCSRMatrixCompare (merge, subtract, max)
double distributed_max_abs_diff(ParCSR &A, ParCSR &B)
{
    DevCSR Ag, Bg, Dg;
    to_global(A, Ag);
    to_global(B, Bg);
    k::csr_add(1.0, Ag, -1.0, Bg, Dg);
    return k::reduce_max_abs(Dg.nnz, Dg.a);
}

// Every rank receives every other rank's rows (global columns): the whole matrix as a rank-local, replicated ParCSR.
// Used for the coarsest operator(s) of a distributed hierarchy ("gather the coarsest 1-2 levels onto every GPU").
ParCSR *replicate_matrix(ParCSR &A, std::vector<int64_t> *row_starts_out)
{
    Context &c = ctx();
    hipStream_t st = c.stream;
    const int nr = c.nranks, me = c.rank, n = A.nrows();
    ELAG_REQUIRE(A.global_cols < (int64_t(1) << 31), "global sizes beyond 2^31 are not supported");
    DevCSR Ag;
    {
        std::vector<int> cm(A.col_map_offd.begin(), A.col_map_offd.end());
        Temp<int> dcm(cm.size());
        if (!cm.empty()) stage_h2d(dcm.p, cm.data(), sizeof(int) * cm.size());
        concat_rows(A.diag, nullptr, (int)A.first_col, A.offd.ncols ? &A.offd : nullptr, dcm.p, 0, (int)A.global_cols, true, Ag);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
    }
    std::vector<double> cnt = allgather_scalar((double)n);
    std::vector<int64_t> starts((size_t)nr + 1, 0);
    for (int r = 0; r < nr; r++) starts[(size_t)r + 1] = starts[(size_t)r] + (int64_t)cnt[(size_t)r];
    if (row_starts_out) *row_starts_out = starts;
    DevCSR full;
    if (nr == 1) {
        k::csr_copy(Ag, full);
    } else {
        // send all my rows to every other rank; receive theirs in rank order
        std::vector<int> procs, sst{0}, rst{0}, idx;
        for (int r = 0; r < nr; r++) {
            if (r == me) continue;
            procs.push_back(r);
            sst.push_back(sst.back() + n);
            rst.push_back(rst.back() + (int)cnt[(size_t)r]);
            for (int i = 0; i < n; i++) idx.push_back(i);
        }
        Temp<int> didx(idx.size());
        if (!idx.empty()) stage_h2d(didx.p, idx.data(), sizeof(int) * idx.size());
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        DevCSR others;
        exchange_rows(Ag, didx.p, procs, sst, procs, rst, others);
        // rank order: rows of ranks < me, my rows, rows of ranks > me
        const int before = (int)starts[(size_t)me];
        DevCSR lo, hi, tmp;
        slice_rows(others, 0, before, lo);
        slice_rows(others, before, others.nrows, hi);
        vstack(lo, Ag, tmp);
        vstack(tmp, hi, full);
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    // a replicated matrix is rank-local: global shape = local shape, no comm package
    std::unique_ptr<ParCSR> M(new ParCSR());
    M->diag = std::move(full);
    M->offd.set_empty(M->diag.nrows, 0);
    M->first_row = 0; M->first_col = 0; M->global_rows = A.global_rows; M->global_cols = A.global_cols;
    M->send_starts.assign(1, 0);
    M->recv_starts.assign(1, 0);
    M->finish_setup();
    return M.release();
}

ParCSR *distributed_rap(ParCSR &RT, ParCSR &A, ParCSR &P, bool ignore_nonlocal_range)
{
    Context &c = ctx();
    hipStream_t st = c.stream;
    ELAG_REQUIRE(P.global_cols < (int64_t(1) << 31) && A.global_cols < (int64_t(1) << 31), "global sizes beyond 2^31 are not supported");
    const int g = A.offd.ncols;
    const int ncR = RT.ncols(), gR = ignore_nonlocal_range ? 0 : RT.offd.ncols;
    // 1. P with global columns, external rows
    DevCSR Pg;
    {
        std::vector<int> cm(P.col_map_offd.begin(), P.col_map_offd.end());
        Temp<int> dcm(cm.size());
        if (!cm.empty()) stage_h2d(dcm.p, cm.data(), sizeof(int) * cm.size());
        concat_rows(P.diag, nullptr, (int)P.first_col, P.offd.ncols ? &P.offd : nullptr, dcm.p, 0, (int)P.global_cols, true, Pg);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
    }
    DevCSR Pext;
    exchange_rows(Pg, A.d_send_elmts, A.send_procs, A.send_starts, A.recv_procs, A.recv_starts, Pext);
    // 2. AP
    DevCSR AP;
    {
        DevCSR B, Aloc;
        vstack(Pg, Pext, B);
        concat_rows(A.diag, nullptr, 0, g ? &A.offd : nullptr, nullptr, A.ncols(), A.ncols() + g, false, Aloc);
        k::csr_spgemm(Aloc, B, AP);
    }
    // 3. C_all = [RT_diag | RT_offd]^T * AP
    DevCSR Call;
    {
        DevCSR RTloc, Rloc;
        concat_rows(RT.diag, nullptr, 0, gR ? &RT.offd : nullptr, nullptr, ncR, ncR + gR, false, RTloc);
        k::csr_transpose(RTloc, Rloc);
        k::csr_spgemm(Rloc, AP, Call);
    }
    // 4. contributions to remote coarse rows go home (reverse direction of RT's comm package)
    DevCSR Cg;
    if (RT.needs_exchange() && !ignore_nonlocal_range) {
        DevCSR Cown, Cext, Crecv;
        slice_rows(Call, 0, ncR, Cown);
        slice_rows(Call, ncR, ncR + gR, Cext);
        exchange_rows(Cext, nullptr, RT.recv_procs, RT.recv_starts, RT.send_procs, RT.send_starts, Crecv);
        if (RT.send_total()) {
            RT.ensure_transpose();
            DevCSR E;
            k::csr_spgemm(*RT.scatT, Crecv, E);
            k::csr_add(1.0, Cown, 1.0, E, Cg);
        } else {
            Cg = std::move(Cown);
        }
    } else {
        Cg = std::move(Call);
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(st));
    // 5. ParCSR assembly
    return parcsr_from_global_columns(Cg, RT.first_col, RT.global_cols, P.first_col, P.ncols(), P.global_cols);
}

}  // namespace elag
