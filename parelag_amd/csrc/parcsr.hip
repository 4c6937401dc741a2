// ParCSR object: creation, halo-exchanging SpMV / SpMV^T, and the setup-side sparse algebra ABI.
#include "parcsr.hpp"

#include <algorithm>
#include <cmath>

#include "comm.hpp"
#include "staging.hpp"

using namespace elag;

elag_hip_parcsr_s::~elag_hip_parcsr_s()
{
    dev_free(d_send_elmts); dev_free(d_send_buf); dev_free(d_ghost); dev_free(d_boundary_rows);
    dev_free(d_rev_send); dev_free(d_rev_recv);
}

void elag_hip_parcsr_s::finish_setup()
{
    const int st = send_total();
    dev_free(d_send_elmts); dev_free(d_send_buf); dev_free(d_ghost); dev_free(d_boundary_rows);
    if (st) {
        d_send_elmts = dev_from_host(send_elmts.data(), (size_t)st);
        d_send_buf = dev_alloc<double>((size_t)st);
    }
    n_boundary = 0;
    if (offd.ncols > 0) {
        d_ghost = dev_alloc<double>((size_t)offd.ncols);
        // boundary rows = rows with a non-empty offd row (list built on the host once per matrix)
        std::vector<int> oi = host_from_dev(offd.I, (size_t)offd.nrows + 1);
        std::vector<int> rows;
        for (int i = 0; i < offd.nrows; i++)
            if (oi[(size_t)i + 1] > oi[(size_t)i]) rows.push_back(i);
        n_boundary = (int)rows.size();
        if (n_boundary) d_boundary_rows = dev_from_host(rows.data(), rows.size());
    }
    ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
}

void elag_hip_parcsr_s::ensure_transpose()
{
    if (diagT) return;
    diagT.reset(new DevCSR());
    k::csr_transpose(diag, *diagT);
    if (offd.ncols > 0) {
        offdT.reset(new DevCSR());
        k::csr_transpose(offd, *offdT);
        d_rev_send = dev_alloc<double>((size_t)offd.ncols);
    }
    const int st = send_total();
    if (st) {
        // scatT: row r lists the positions k with send_elmts[k] == r (ascending k): deterministic scatter-add
        std::vector<int> I((size_t)diag.ncols + 1, 0), J((size_t)st);
        std::vector<double> a((size_t)st, 1.0);
        for (int k = 0; k < st; k++) I[(size_t)send_elmts[(size_t)k] + 1]++;
        for (int r = 0; r < diag.ncols; r++) I[(size_t)r + 1] += I[(size_t)r];
        std::vector<int> pos(I.begin(), I.end() - 1);
        for (int k = 0; k < st; k++) J[(size_t)pos[(size_t)send_elmts[(size_t)k]]++] = k;
        scatT.reset(new DevCSR());
        scatT->upload(diag.ncols, st, I.data(), J.data(), a.data());
        d_rev_recv = dev_alloc<double>((size_t)st);
    }
}

namespace elag {

namespace {
__global__ void gather_scaled_kernel(int n, const int *__restrict__ idx, const double *__restrict__ x,
                                     const double *__restrict__ cs, double *__restrict__ buf)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = idx[i];
        buf[i] = cs ? x[j] * cs[j] : x[j];
    }
}
}  // namespace

void par_spmv(ParCSR &A, k::SpmvArgs a)
{
    Context &c = ctx();
    if (!A.needs_exchange()) {
        a.xg = nullptr; a.csg = nullptr; a.rows = nullptr; a.skip_rows_with_offd = 0;
        k::spmv_fused(A.diag, nullptr, a, c.stream);
        return;
    }
    // 1. pack x[send_map_elmts] (pre-multiplied by the column scale) on the compute stream
    const int st = A.send_total();
    if (st) {
        const unsigned g = (unsigned)std::min<long long>(((long long)st + 255) / 256, 1024);
        hipLaunchKernelGGL(gather_scaled_kernel, dim3(g), dim3(256), 0, c.stream, st, (const int *)A.d_send_elmts, a.x, a.cs, A.d_send_buf);
        ELAG_LAUNCH_CHECK();
    }
    ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
    // 2. neighbour exchange on the comm stream (RCCL send/recv group over xGMI)
    comm_neighbor_exchange(A.d_send_buf, (int)A.send_procs.size(), A.send_procs.data(), A.send_starts.data(), A.d_ghost,
                           (int)A.recv_procs.size(), A.recv_procs.data(), A.recv_starts.data(), sizeof(double), c.comm_stream);
    ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
    // 3. interior rows (empty offd row) overlap with the exchange
    k::SpmvArgs in = a;
    in.xg = nullptr; in.csg = nullptr; in.rows = nullptr; in.skip_rows_with_offd = 1;
    k::spmv_fused(A.diag, &A.offd, in, c.stream);
    // 4. boundary rows once the ghost values have landed
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
    if (A.n_boundary) {
        k::SpmvArgs bd = a;
        bd.xg = A.d_ghost; bd.csg = nullptr; bd.rows = A.d_boundary_rows; bd.nrows_sel = A.n_boundary; bd.skip_rows_with_offd = 0;
        k::spmv_fused(A.diag, &A.offd, bd, c.stream);
    }
}

void par_halo_exchange(ParCSR &A, const double *x)
{
    Context &c = ctx();
    if (!A.needs_exchange()) return;
    const int st = A.send_total();
    if (st) {
        const unsigned g = (unsigned)std::min<long long>(((long long)st + 255) / 256, 1024);
        hipLaunchKernelGGL(gather_scaled_kernel, dim3(g), dim3(256), 0, c.stream, st, (const int *)A.d_send_elmts, x, (const double *)nullptr,
                           A.d_send_buf);
        ELAG_LAUNCH_CHECK();
    }
    ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
    comm_neighbor_exchange(A.d_send_buf, (int)A.send_procs.size(), A.send_procs.data(), A.send_starts.data(), A.d_ghost,
                           (int)A.recv_procs.size(), A.recv_procs.data(), A.recv_starts.data(), sizeof(double), c.comm_stream);
    ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
}

// y += alpha * (reverse halo of A.d_rev_send): ghost-column contributions go back to their owners
// (hypre_ParCSRMatrixMatvecT, comm job 2, hypre_ParCSRMatrixMatvecBoolInt.c:296-323).
void reverse_halo_start(ParCSR &A)
{
    Context &c = ctx();
    ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
    comm_neighbor_exchange(A.d_rev_send, (int)A.recv_procs.size(), A.recv_procs.data(), A.recv_starts.data(), A.d_rev_recv,
                           (int)A.send_procs.size(), A.send_procs.data(), A.send_starts.data(), sizeof(double), c.comm_stream);
    ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
}

void reverse_halo_finish(ParCSR &A, double alpha, double *y)
{
    Context &c = ctx();
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
    if (A.scatT) {
        k::SpmvArgs s;
        s.x = A.d_rev_recv; s.alpha = alpha; s.yin = y; s.beta = 1.0; s.out = y;
        k::spmv_fused(*A.scatT, nullptr, s, c.stream);
    }
}

void reverse_halo_add(ParCSR &A, double alpha, double *y)
{
    reverse_halo_start(A);
    reverse_halo_finish(A, alpha, y);
}

void par_spmvT(ParCSR &A, double alpha, const double *x, double beta, double *y, const double *row_scale)
{
    Context &c = ctx();
    A.ensure_transpose();
    const bool halo = A.needs_exchange();
    ELAG_REQUIRE(!(halo && row_scale), "internal: scaled transposed product needs a rank-local operator");
    if (halo) {
        // y_ghost = offd^T x, then the reverse exchange (roles of the send/recv lists swapped)
        if (A.offd.ncols > 0) {
            k::SpmvArgs g;
            g.x = x; g.alpha = 1.0; g.out = A.d_rev_send;
            k::spmv_fused(*A.offdT, nullptr, g, c.stream);
        }
        reverse_halo_start(A);
    }
    k::SpmvArgs d;
    d.x = x; d.alpha = alpha; d.yin = (beta != 0.0) ? y : nullptr; d.beta = beta; d.out = y; d.s = row_scale;
    k::spmv_fused(*A.diagT, nullptr, d, c.stream);
    if (halo) reverse_halo_finish(A, alpha, y);
}

ParCSR *parcsr_from_diag(DevCSR &&d, int64_t first_row, int64_t first_col, int64_t grows, int64_t gcols)
{
    std::unique_ptr<ParCSR> M(new ParCSR());
    M->diag = std::move(d);
    M->offd.set_empty(M->diag.nrows, 0);
    M->first_row = first_row; M->first_col = first_col; M->global_rows = grows; M->global_cols = gcols;
    M->send_starts.assign(1, 0);
    M->recv_starts.assign(1, 0);
    M->finish_setup();
    return M.release();
}

}  // namespace elag



extern "C" {

int elag_hip_parcsr_create_from_global_csr(elag_hip_parcsr_t *out, int num_rows, const int *row_ptr, const int *global_col, const double *values,
                                           int64_t first_row, int64_t global_num_rows, int64_t first_col, int num_cols_local, int64_t global_num_cols)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && num_rows >= 0 && num_cols_local >= 0 && (num_rows == 0 || row_ptr), "bad argument");
        ELAG_REQUIRE(global_num_cols < (int64_t(1) << 31) && global_num_rows < (int64_t(1) << 31), "global sizes beyond 2^31 are not supported");
        ELAG_REQUIRE(first_col >= 0 && first_col + num_cols_local <= global_num_cols && first_row >= 0 && first_row + num_rows <= global_num_rows,
                     "this rank's rows / columns lie outside the global matrix");
        DevCSR Cg;
        Cg.upload(num_rows, (int)global_num_cols, row_ptr, global_col, values);
        *out = parcsr_from_global_columns(Cg, first_row, global_num_rows, first_col, num_cols_local, global_num_cols);
    });
}

int elag_hip_parcsr_create(elag_hip_parcsr_t *out, const elag_hip_parcsr_desc *d)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && d, "null argument");
        ELAG_REQUIRE(d->num_rows >= 0 && d->num_cols_diag >= 0 && d->num_cols_offd >= 0, "negative dimension");
        std::unique_ptr<ParCSR> M(new ParCSR());
        M->diag.upload(d->num_rows, d->num_cols_diag, d->diag_i, d->diag_j, d->diag_a);
        if (d->num_cols_offd > 0) {
            ELAG_REQUIRE(d->offd_i && d->col_map_offd, "offd block given without row pointers / col_map_offd");
            M->offd.upload(d->num_rows, d->num_cols_offd, d->offd_i, d->offd_j, d->offd_a);
            M->col_map_offd.assign(d->col_map_offd, d->col_map_offd + d->num_cols_offd);
            for (int i = 1; i < d->num_cols_offd; i++)
                ELAG_REQUIRE(M->col_map_offd[(size_t)i] > M->col_map_offd[(size_t)i - 1], "col_map_offd must ascend strictly");
        } else {
            M->offd.set_empty(d->num_rows, 0);
        }
        M->first_row = d->first_row; M->first_col = d->first_col_diag;
        M->global_rows = d->global_num_rows ? d->global_num_rows : d->num_rows;
        M->global_cols = d->global_num_cols ? d->global_num_cols : d->num_cols_diag;
        // comm package
        ELAG_REQUIRE(d->num_sends >= 0 && d->num_recvs >= 0, "negative neighbour count");
        M->send_starts.assign(1, 0);
        M->recv_starts.assign(1, 0);
        if (d->num_sends) {
            ELAG_REQUIRE(d->send_procs && d->send_map_starts && d->send_map_elmts, "incomplete send lists");
            M->send_procs.assign(d->send_procs, d->send_procs + d->num_sends);
            M->send_starts.assign(d->send_map_starts, d->send_map_starts + d->num_sends + 1);
            ELAG_REQUIRE(M->send_starts[0] == 0, "send_map_starts[0] must be 0");
            const int st = M->send_starts.back();
            M->send_elmts.assign(d->send_map_elmts, d->send_map_elmts + st);
            for (int k = 0; k < st; k++)
                ELAG_REQUIRE(M->send_elmts[(size_t)k] >= 0 && M->send_elmts[(size_t)k] < d->num_cols_diag, "send_map_elmts[%d] out of range", k);
        }
        if (d->num_recvs) {
            ELAG_REQUIRE(d->recv_procs && d->recv_vec_starts, "incomplete recv lists");
            M->recv_procs.assign(d->recv_procs, d->recv_procs + d->num_recvs);
            M->recv_starts.assign(d->recv_vec_starts, d->recv_vec_starts + d->num_recvs + 1);
            ELAG_REQUIRE(M->recv_starts[0] == 0 && M->recv_starts.back() == d->num_cols_offd,
                         "recv_vec_starts must cover exactly the %d ghost columns", d->num_cols_offd);
        } else {
            ELAG_REQUIRE(d->num_cols_offd == 0, "%d ghost columns without a receive list", d->num_cols_offd);
        }
        // neighbour ranks must exist in the communicator the exchange will run on (a descriptor built before
        // elag_hip_comm_init is checked against the size it states itself: at least non-negative)
        const int nr = ctx().nranks;
        for (int p : M->send_procs) ELAG_REQUIRE(p >= 0 && (nr <= 1 || p < nr), "send proc %d outside the communicator of %d ranks", p, nr);
        for (int p : M->recv_procs) ELAG_REQUIRE(p >= 0 && (nr <= 1 || p < nr), "recv proc %d outside the communicator of %d ranks", p, nr);
        M->finish_setup();
        *out = M.release();
    });
}

int elag_hip_parcsr_destroy(elag_hip_parcsr_t A)
{
    return guarded([&] {
        if (A) { (void)hipStreamSynchronize(ctx().stream); delete A; }
    });
}

int elag_hip_parcsr_get_info(elag_hip_parcsr_t A, elag_hip_parcsr_info *info)
{
    return guarded([&] {
        ELAG_REQUIRE(A && info, "null argument");
        info->num_rows = A->nrows(); info->num_cols_diag = A->ncols(); info->num_cols_offd = A->ncols_offd();
        info->nnz_diag = A->diag.nnz; info->nnz_offd = A->offd.nnz;
        info->first_row = A->first_row; info->first_col_diag = A->first_col;
        info->global_num_rows = A->global_rows; info->global_num_cols = A->global_cols;
        info->num_sends = (int)A->send_procs.size(); info->num_recvs = (int)A->recv_procs.size();
        info->send_total = A->send_total();
    });
}

static void export_csr(const DevCSR &M, bool diag_first, int *hi, int *hj, double *ha)
{
    std::vector<int> I = host_from_dev(M.I, (size_t)M.nrows + 1);
    std::vector<int> J = host_from_dev(M.J, (size_t)M.nnz);
    std::vector<double> a = host_from_dev(M.a, (size_t)M.nnz);
    if (diag_first && M.nrows == M.ncols) {
        for (int i = 0; i < M.nrows; i++) {
            const int s = I[(size_t)i], e = I[(size_t)i + 1];
            for (int k = s; k < e; k++)
                if (J[(size_t)k] == i) {
                    const double v = a[(size_t)k];
                    for (int q = k; q > s; q--) { J[(size_t)q] = J[(size_t)q - 1]; a[(size_t)q] = a[(size_t)q - 1]; }
                    J[(size_t)s] = i; a[(size_t)s] = v;
                    break;
                }
        }
    }
    if (hi) memcpy(hi, I.data(), sizeof(int) * I.size());
    if (hj && M.nnz) memcpy(hj, J.data(), sizeof(int) * J.size());
    if (ha && M.nnz) memcpy(ha, a.data(), sizeof(double) * a.size());
}

int elag_hip_parcsr_export(elag_hip_parcsr_t A, int diag_first, int *diag_i, int *diag_j, double *diag_a,
                           int *offd_i, int *offd_j, double *offd_a, int64_t *col_map_offd)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A, "null matrix");
        export_csr(A->diag, diag_first != 0, diag_i, diag_j, diag_a);
        if (offd_i) export_csr(A->offd, false, offd_i, offd_j, offd_a);
        if (col_map_offd && !A->col_map_offd.empty())
            memcpy(col_map_offd, A->col_map_offd.data(), sizeof(int64_t) * A->col_map_offd.size());
    });
}

int elag_hip_parcsr_spmv(elag_hip_parcsr_t A, int trans, double alpha, const double *x, double beta, double *y, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && x && y, "null argument");
        check_loc(loc);
        const size_t nx = trans ? (size_t)A->nrows() : (size_t)A->ncols();
        const size_t ny = trans ? (size_t)A->ncols() : (size_t)A->nrows();
        ELAG_REQUIRE(x != y, "x and y must not alias");
        VecIn xin(x, nx, loc);
        VecInOut yio(y, ny, loc, beta != 0.0);
        if (trans) {
            par_spmvT(*A, alpha, xin.d, beta, yio.d);
        } else {
            k::SpmvArgs a;
            a.x = xin.d; a.alpha = alpha; a.beta = beta; a.yin = (beta != 0.0) ? yio.d : nullptr; a.out = yio.d;
            par_spmv(*A, a);
        }
        yio.finish();
    });
}

int elag_hip_parcsr_residual(elag_hip_parcsr_t A, const double *b, const double *x, double *r, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && b && x && r, "null argument");
        check_loc(loc);
        ELAG_REQUIRE(r != x, "r and x must not alias");
        VecIn bin(b, (size_t)A->nrows(), loc), xin(x, (size_t)A->ncols(), loc);
        VecInOut rio(r, (size_t)A->nrows(), loc, false);
        k::SpmvArgs a;
        a.x = xin.d; a.alpha = -1.0; a.f = bin.d; a.gamma = 1.0; a.out = rio.d;
        par_spmv(*A, a);
        rio.finish();
    });
}

int elag_hip_parcsr_transpose(elag_hip_parcsr_t A, elag_hip_parcsr_t *At)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && At, "null argument");
        if (is_distributed(*A)) { *At = distributed_transpose(*A); return; }
        DevCSR T;
        k::csr_transpose(A->diag, T);
        *At = parcsr_from_diag(std::move(T), A->first_col, A->first_row, A->global_cols, A->global_rows);
    });
}

int elag_hip_parcsr_matmul(elag_hip_parcsr_t A, elag_hip_parcsr_t B, elag_hip_parcsr_t *C)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && B && C, "null argument");
        ELAG_REQUIRE(A->ncols() == B->nrows(), "matmul: A has %d columns, B has %d rows", A->ncols(), B->nrows());
        if (is_distributed(*A) || is_distributed(*B)) {
            ELAG_REQUIRE(A->global_cols == B->global_rows && A->first_col == B->first_row, "matmul: distributed shapes do not chain");
            *C = distributed_matmul(*A, *B);
            return;
        }
        DevCSR out;
        k::csr_spgemm(A->diag, B->diag, out);
        *C = parcsr_from_diag(std::move(out), A->first_row, B->first_col, A->global_rows, B->global_cols);
    });
}

int elag_hip_parcsr_rap(elag_hip_parcsr_t RT, elag_hip_parcsr_t A, elag_hip_parcsr_t P, elag_hip_parcsr_t *RAP)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(RT && A && P && RAP, "null argument");
        if (is_distributed(*RT) || is_distributed(*A) || is_distributed(*P)) {
            ELAG_REQUIRE(RT->nrows() == A->nrows() && A->ncols() == P->nrows() && RT->global_rows == A->global_rows &&
                         A->global_cols == P->global_rows, "rap: distributed shapes do not chain");
            *RAP = distributed_rap(*RT, *A, *P);
            return;
        }
        ELAG_REQUIRE(RT->nrows() == A->nrows() && A->ncols() == P->nrows(),
                     "rap: shapes RT %dx%d, A %dx%d, P %dx%d do not chain", RT->nrows(), RT->ncols(), A->nrows(), A->ncols(),
                     P->nrows(), P->ncols());
        DevCSR out;
        RT->ensure_transpose();
        k::csr_rap(*RT->diagT, A->diag, P->diag, out);
        *RAP = parcsr_from_diag(std::move(out), RT->first_col, P->first_col, RT->global_cols, P->global_cols);
    });
}

int elag_hip_parcsr_rap_ignore_nonlocal_range(elag_hip_parcsr_t RT, elag_hip_parcsr_t A, elag_hip_parcsr_t P, elag_hip_parcsr_t *out)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(RT && A && P && out, "null argument");
        ELAG_REQUIRE(RT->nrows() == A->nrows() && A->ncols() == P->nrows(), "IgnoreNonLocalRange: shapes do not chain");
        *out = distributed_rap(*RT, *A, *P, true);
    });
}

int elag_hip_parcsr_spmvT_ignore_nonlocal(elag_hip_parcsr_t A, const double *x, double *y, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && x && y && x != y, "bad argument");
        check_loc(loc);
        VecIn xin(x, (size_t)A->nrows(), loc);
        VecInOut yio(y, (size_t)A->ncols(), loc, false);
        A->ensure_transpose();
        k::SpmvArgs d;
        d.x = xin.d; d.alpha = 1.0; d.out = yio.d;
        k::spmv_fused(*A->diagT, nullptr, d);
        yio.finish();
    });
}

int elag_hip_parcsr_replicate(elag_hip_parcsr_t A, elag_hip_parcsr_t *Arep, int64_t *row_starts)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && Arep, "null argument");
        std::vector<int64_t> st;
        *Arep = replicate_matrix(*A, &st);
        if (row_starts) std::copy(st.begin(), st.end(), row_starts);
    });
}

int elag_hip_parcsr_add(double a, elag_hip_parcsr_t A, double b, elag_hip_parcsr_t B, elag_hip_parcsr_t *C)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && B && C, "null argument");
        if (is_distributed(*A) || is_distributed(*B)) {
            ELAG_REQUIRE(A->nrows() == B->nrows() && A->ncols() == B->ncols() && A->global_rows == B->global_rows &&
                         A->global_cols == B->global_cols && A->first_row == B->first_row && A->first_col == B->first_col,
                         "add: incompatible matrix dimensions");
            *C = distributed_add(a, *A, b, *B);
            return;
        }
        DevCSR out;
        k::csr_add(a, A->diag, b, B->diag, out);
        *C = parcsr_from_diag(std::move(out), A->first_row, A->first_col, A->global_rows, A->global_cols);
    });
}

int elag_hip_parcsr_rdp(elag_hip_parcsr_t RT, const double *d, int loc, elag_hip_parcsr_t P, elag_hip_parcsr_t *RDP)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(RT && d && P && RDP, "null argument");
        check_loc(loc);
        ELAG_REQUIRE(RT->nrows() == P->nrows(), "rdp: RT and P must have the same number of rows");
        if (is_distributed(*RT) || is_distributed(*P)) {
            // across ranks: the triple product with the diagonal operator diag(d) in the middle
            ELAG_REQUIRE(RT->first_row == P->first_row && RT->global_rows == P->global_rows, "rdp: RT and P must share the row partition");
            std::vector<double> dh((size_t)P->nrows());
            if (loc == ELAG_HIP_DEVICE) dh = host_from_dev(d, (size_t)P->nrows());
            else std::copy(d, d + P->nrows(), dh.begin());
            elag_hip_parcsr_t D = nullptr;
            int rc = elag_hip_parcsr_diagonal(&D, P->nrows(), P->first_row, P->global_rows, dh.data());
            if (rc) fail(rc, "%s", elag_hip_last_error());
            std::unique_ptr<ParCSR> Dg(D);
            *RDP = distributed_rap(*RT, *Dg, *P);
            return;
        }
        VecIn din(d, (size_t)P->nrows(), loc);
        DevCSR DP, out;
        k::csr_copy(P->diag, DP);
        k::csr_scale_rows(DP, din.d);
        RT->ensure_transpose();
        k::csr_spgemm(*RT->diagT, DP, out);
        *RDP = parcsr_from_diag(std::move(out), RT->first_col, P->first_col, RT->global_cols, P->global_cols);
    });
}

int elag_hip_parcsr_fix_zero_rows(elag_hip_parcsr_t A, int *num_fixed)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A, "null matrix");
        const int n = k::fix_zero_rows(A->diag, &A->offd);
        A->diagT.reset(); A->offdT.reset();  // cached transposes are stale now
        if (num_fixed) *num_fixed = n;
    });
}

int elag_hip_parcsr_delete_zeros(elag_hip_parcsr_t A, double tol)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A, "null matrix");
        DevCSR out;
        k::csr_delete_zeros(A->diag, tol, out);
        A->diag = std::move(out);
        if (A->offd.nnz) {   // the ghost column set (and the comm package built on it) is kept, as the reference rebuilds it on the same col_map
            DevCSR o;
            k::csr_delete_zeros(A->offd, tol, o);
            const int nco = A->offd.ncols;
            A->offd = std::move(o);
            A->offd.ncols = nco;
        }
        A->diagT.reset(); A->offdT.reset();
        A->finish_setup();
    });
}

int elag_hip_parcsr_sign_transform(elag_hip_parcsr_t A)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A, "null matrix");
        A->diag.release_sign(); A->offd.release_sign();
        k::values_sign(A->diag.nnz, A->diag.a);
        k::values_sign(A->offd.nnz, A->offd.a);
        A->diagT.reset(); A->offdT.reset();
    });
}

int elag_hip_parcsr_diagonal(elag_hip_parcsr_t *out, int num_rows, int64_t first_row, int64_t global_num_rows, const double *d_host)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && num_rows >= 0, "bad argument");
        std::vector<int> I((size_t)num_rows + 1), J((size_t)num_rows);
        std::vector<double> a((size_t)num_rows, 1.0);
        for (int i = 0; i <= num_rows; i++) I[(size_t)i] = i;
        for (int i = 0; i < num_rows; i++) { J[(size_t)i] = i; if (d_host) a[(size_t)i] = d_host[i]; }
        DevCSR D;
        D.upload(num_rows, num_rows, I.data(), J.data(), a.data());
        const int64_t g = global_num_rows ? global_num_rows : num_rows;
        *out = parcsr_from_diag(std::move(D), first_row, first_row, g, g);
    });
}

int elag_hip_parcsr_norm(elag_hip_parcsr_t A, int kind, double *value)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && value, "null argument");
        double v = 0.0;
        if (kind == 2) {
            v = std::max(k::reduce_max_abs(A->diag.nnz, A->diag.a), k::reduce_max_abs(A->offd.nnz, A->offd.a));
            comm_allreduce_scalar(&v, 1);
        } else if (kind == 3) {
            v = k::reduce_sum_sq(A->diag.nnz, A->diag.a) + k::reduce_sum_sq(A->offd.nnz, A->offd.a);
            comm_allreduce_scalar(&v, 0);
            v = std::sqrt(v);
        } else if (kind == 1) {
            // linf: max_i sum_j |a_ij| -- row sums of |diag| + |offd| (hypre forms |A| 1 with a matvec)
            const int n = A->nrows();
            Temp<double> rs((size_t)n);
            Temp<int> bad(1);
            zero_async(bad, sizeof(int));
            k::l1_norms(A->diag, &A->offd, 1, rs, bad);
            v = k::reduce_max_abs(n, rs);
            comm_allreduce_scalar(&v, 1);
        } else if (kind == 0) {
            // l1: max_j sum_i |a_ij| = row sums of the explicit transposes, ghost columns sent to their owners
            A->ensure_transpose();
            const int nc = A->ncols();
            Temp<double> cs((size_t)nc);
            Temp<int> bad(1);
            zero_async(bad, sizeof(int));
            k::l1_norms(*A->diagT, nullptr, 1, cs, bad);
            k::values_abs(nc, cs, cs);
            if (A->offd.ncols > 0) {
                k::l1_norms(*A->offdT, nullptr, 1, A->d_rev_send, bad);
                k::values_abs(A->offd.ncols, A->d_rev_send, A->d_rev_send);
                reverse_halo_add(*A, 1.0, cs);
            }
            v = k::reduce_max_abs(nc, cs);
            comm_allreduce_scalar(&v, 1);
        } else {
            fail(ELAG_HIP_ERR_ARG, "unknown norm kind %d", kind);
        }
        *value = v;
    });
}

int elag_hip_parcsr_compare(elag_hip_parcsr_t A, elag_hip_parcsr_t B, double tol, int *mask)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && B && mask, "null argument");
        int cmp = 0;
        if (A->global_rows != B->global_rows) cmp |= 1;
        if (A->global_cols != B->global_cols) cmp |= 2;
        if (A->first_row != B->first_row) cmp |= 4;
        if (A->first_row + A->nrows() != B->first_row + B->nrows()) cmp |= 8;
        if (A->first_col != B->first_col) cmp |= 16;
        if (A->first_col + A->ncols() != B->first_col + B->ncols()) cmp |= 32;
        if (cmp == 0) {
            double diff;
            if (is_distributed(*A) || is_distributed(*B)) {
                diff = distributed_max_abs_diff(*A, *B);
            } else {
                DevCSR D;
                k::csr_add(1.0, A->diag, -1.0, B->diag, D);
                diff = k::reduce_max_abs(D.nnz, D.a);
            }
            comm_allreduce_scalar(&diff, 0);
            if (diff > tol) cmp |= 64;
        } else {
            cmp |= 64;
        }
        *mask = cmp;
    });
}

int elag_hip_parcsr_set_spmv_lanes(elag_hip_parcsr_t A, int lanes)
{
    return guarded([&] {
        ELAG_REQUIRE(A, "null matrix");
        if (lanes == -1 || lanes == -2) {
            // the LDS-staged stream kernel regardless of size; -1 additionally forces its windowed x-cache form
            if (!A->diag.block_row) k::build_stream_plan(A->diag);         // (small operators get their window plan on request)
            ELAG_REQUIRE(A->diag.block_row && A->diag.max_row_nnz <= DevCSR::kStreamMaxRow, "rows too long for the stream kernel");
            A->diag.lanes_per_row = 0;
            A->diag.release_xc();
            if (lanes == -1) k::prepare_spmv(A->diag, true);
            else A->diag.xc_state = -1;
            return;
        }
        ELAG_REQUIRE(lanes >= 0 && lanes <= 64 && (lanes & (lanes - 1)) == 0, "lanes must be -2, -1, 0 or a power of two <= 64");
        if (lanes == 0) { A->diag.choose_spmv_shape(); A->offd.choose_spmv_shape(); }
        else A->diag.lanes_per_row = lanes;
    });
}

int elag_hip_parcsr_get_spmv_form(elag_hip_parcsr_t A, int *form, int *lanes)
{
    return guarded([&] {
        ELAG_REQUIRE(A && form, "null argument");
        const DevCSR &D = A->diag;
        const bool stream = D.lanes_per_row == 0 && D.block_row;
        *form = stream ? (D.xc_state == 1 ? 2 : 1) : 0;
        // row-group kernel: lanes per row; stream / x-cache kernels: entries per window (workgroup)
        if (lanes) *lanes = stream ? (D.xc_state == 1 ? D.xc_window : DevCSR::kStreamNnz) : D.lanes_per_row;
    });
}

int elag_hip_parcsr_get_diagonal(elag_hip_parcsr_t A, double *d, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && d, "null argument");
        check_loc(loc);
        VecInOut dio(d, (size_t)A->nrows(), loc, false);
        k::extract_diagonal(A->diag, dio.d);
        dio.finish();
    });
}

}  // extern "C"
