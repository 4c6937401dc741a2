// Device-resident ParCSR matrix (hypre_ParCSRMatrix equivalent) and its halo-exchanging products.
#pragma once
#include "kernels.hpp"

struct elag_hip_parcsr_s {
    elag::DevCSR diag, offd;
    std::vector<int64_t> col_map_offd;
    int64_t first_row = 0, first_col = 0, global_rows = 0, global_cols = 0;

    // comm package (hypre_ParCSRCommPkg) -- host copies drive the RCCL calls
    std::vector<int> send_procs, send_starts, send_elmts, recv_procs, recv_starts;
    int *d_send_elmts = nullptr;
    double *d_send_buf = nullptr;  // send_total
    double *d_ghost = nullptr;     // num_cols_offd
    // rows whose offd part is non-empty (boundary rows): processed after the halo has arrived
    int *d_boundary_rows = nullptr;
    int n_boundary = 0;

    // explicit transposes, built on first MatvecT (deterministic gather form instead of scatter-add)
    std::unique_ptr<elag::DevCSR> diagT, offdT;
    std::unique_ptr<elag::DevCSR> scatT;  // num_cols_diag x send_total, unit entries: y[send_elmts[k]] += buf[k]
    double *d_rev_send = nullptr;         // num_cols_offd  (offd^T x)
    double *d_rev_recv = nullptr;         // send_total

    elag_hip_parcsr_s() = default;
    ~elag_hip_parcsr_s();
    elag_hip_parcsr_s(const elag_hip_parcsr_s &) = delete;
    elag_hip_parcsr_s &operator=(const elag_hip_parcsr_s &) = delete;

    int nrows() const { return diag.nrows; }
    int ncols() const { return diag.ncols; }
    int ncols_offd() const { return offd.ncols; }
    int send_total() const { return send_starts.empty() ? 0 : send_starts.back(); }
    // a rank takes part in the exchange if it has ghosts OR a neighbour needs its values (the two can differ for
    // rectangular operators such as P)
    bool needs_exchange() const { return offd.ncols > 0 || send_total() > 0; }

    void finish_setup();      // device copies of the comm lists, boundary row list, buffers
    void ensure_transpose();  // diagT / offdT / scatT
};

namespace elag {
typedef elag_hip_parcsr_s ParCSR;

// out_i = beta*yin_i + s_i*(gamma*f_i + alpha*s2_i*(A x)_i) with the halo exchange overlapped with the
// interior rows.  `a.x` is the local vector (length ncols of diag); a.xg / a.csg are filled in here.
void par_spmv(ParCSR &A, k::SpmvArgs a);
// A.d_ghost <- the ghost values of the local vector x (pack, neighbour exchange on the comm stream, compute stream waits)
void par_halo_exchange(ParCSR &A, const double *x);
// y = alpha*A^T x + beta*y  (x: local rows, y: local columns of diag)
// `row_scale` (optional, rank-local operators only): y_i = row_scale_i * (alpha*(A^T x)_i) + beta*y_i
void par_spmvT(ParCSR &A, double alpha, const double *x, double beta, double *y, const double *row_scale = nullptr);

// dist_rap.hip
ParCSR *distributed_rap(ParCSR &RT, ParCSR &A, ParCSR &P, bool ignore_nonlocal_range = false);
ParCSR *replicate_matrix(ParCSR &A, std::vector<int64_t> *row_starts_out);
ParCSR *distributed_add(double a, ParCSR &A, double b, ParCSR &B);
ParCSR *distributed_matmul(ParCSR &A, ParCSR &B);
ParCSR *distributed_transpose(ParCSR &A);
double distributed_max_abs_diff(ParCSR &A, ParCSR &B);
ParCSR *parcsr_from_global_columns(const DevCSR &Cg, int64_t first_row, int64_t global_rows, int64_t first_col, int ncols_local,
                                   int64_t global_cols);
// this rank's rows with GLOBAL column ids (hypre_MergeDiagAndOffd), columns ascending
void parcsr_to_global(ParCSR &M, DevCSR &out);
// [X | Y] row by row; column ids: map ? map[j] : j + off; optional per-row sort
void csr_hstack(const DevCSR &X, const int *mapX, int offX, const DevCSR *Y, const int *mapY, int offY, int ncols_out, bool sort, DevCSR &out);
void csr_vstack(const DevCSR &X, const DevCSR &Y, DevCSR &out);
// one double from every rank, in rank order
std::vector<double> comm_allgather_scalar(double v);
// deep copy (matrix, comm package, buffers)
ParCSR *parcsr_clone(ParCSR &M);
inline bool is_distributed(const ParCSR &M) { return M.needs_exchange() || M.global_rows != M.nrows() || M.global_cols != M.ncols(); }

// Build a single-rank ParCSR from a device CSR (takes ownership of `d`).
ParCSR *parcsr_from_diag(DevCSR &&d, int64_t first_row, int64_t first_col, int64_t grows, int64_t gcols);
}  // namespace elag
