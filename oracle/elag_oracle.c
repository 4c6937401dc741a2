/*
 * elag_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic on ParElag's AMGe V-cycle hot path, with the loop
 * structure of the reference so that summation order matches.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path (libelag_hip.so) never does.
 *
 * PARITY UNPINNED at kernel granularity: the arithmetic of this path lives in hypre / MFEM, which are
 * third-party, un-vendored (INSTALL "Dependencies": hypre >= 2.10.0b, MFEM; no lock file) and absent
 * from /root/reference and from this image.  The reference's own tests only pin whole-solver residuals
 * through BoomerAMG/L1-GS/PCG (examples/CMakeLists.txt:146-182), which cannot be reproduced without
 * those libraries.  Each function below therefore restates the published hypre 2.x / MFEM 4.x algorithm
 * and cites the ParElag call site (or the in-repo twin in src/hypreExtension) it follows; the oracle is
 * validated by dense recomputation (scipy), closed-form cases and the reference's algebraic invariants
 * (src/amge/DeRhamSequence.cpp:694-970) in tests/test_oracle_*.py.
 *
 * Conventions: CSR i/j/data with int32 indices (HYPRE_Int), fp64 values; a ParCSR
 * matrix is a `diag` CSR (local columns) plus an `offd` CSR (compressed ghost columns).  The halo
 * exchange itself is performed by the caller (oracle/oracle.py) between the two CSR calls, exactly where
 * src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:189-196 places it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* The same source builds the multi-core timing baseline (oracle/Makefile target libelag_oracle_omp.so, -fopenmp): the
 * row loops of the V-cycle are split statically over the cores, which is the work split of the reference's
 * one-MPI-rank-per-core runs without their message passing (so it flatters the CPU).  Row sums keep their storage
 * order; only P^T r is formed from an explicit transpose (as newer hypre does) instead of the serial scatter-add, so
 * that result differs from the serial oracle in the last bits.  The parity checks always use the serial build. */
#ifdef _OPENMP
#include <omp.h>
#define ORC_PFOR _Pragma("omp parallel for schedule(static)")
#else
#define ORC_PFOR
#endif
ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * CSR matvec  y = alpha*A*x + beta*y
 * Follows the double-precision twin of hypre_CSRMatrixMatvecBoolInt
 * (src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:16-137): scale y by beta/alpha, accumulate
 * each row in storage order into a temporary, add, scale by alpha.  (The rownnz short-cut at :92-107
 * performs the same additions, so it is not restated.)
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_csr_matvec(int n, const int *I, const int *J, const double *a,
                            double alpha, const double *x, double beta, double *y)
{
    int i;
    if (alpha == 0.0) { for (i = 0; i < n; i++) y[i] *= beta; return; }
    double temp = beta / alpha;
    if (temp != 1.0) {
        if (temp == 0.0) { ORC_PFOR for (i = 0; i < n; i++) y[i] = 0.0; }
        else             { ORC_PFOR for (i = 0; i < n; i++) y[i] *= temp; }
    }
    ORC_PFOR
    for (i = 0; i < n; i++) {
        double t = 0.0;
        for (int jj = I[i]; jj < I[i + 1]; jj++) t += a[jj] * x[J[jj]];
        y[i] += t;
    }
    if (alpha != 1.0) { ORC_PFOR for (i = 0; i < n; i++) y[i] *= alpha; }
}

/* y = alpha*A^T*x + beta*y ; scatter-add form of hypre_CSRMatrixMatvecT, the kernel behind
 * P->MultTranspose (src/linalg/solver_ops/ParELAG_Hierarchy.cpp:199-203); the parallel wrapper is the
 * disabled copy at hypre_ParCSRMatrixMatvecBoolInt.c:212-345. */
ORC_API void orc_csr_matvecT(int nrows, int ncols, const int *I, const int *J, const double *a,
                             double alpha, const double *x, double beta, double *y)
{
    int i, jj;
    if (alpha == 0.0) { for (i = 0; i < ncols; i++) y[i] *= beta; return; }
    double temp = beta / alpha;
    if (temp != 1.0) {
        if (temp == 0.0) for (i = 0; i < ncols; i++) y[i] = 0.0;
        else             for (i = 0; i < ncols; i++) y[i] *= temp;
    }
    for (i = 0; i < nrows; i++)
        for (jj = I[i]; jj < I[i + 1]; jj++) y[J[jj]] += a[jj] * x[i];
    if (alpha != 1.0) for (i = 0; i < ncols; i++) y[i] *= alpha;
}

/* Integer twin, verbatim semantics of hypre_CSRMatrixMatvecBoolInt (values of A ignored = 1). */
ORC_API void orc_csr_matvec_boolint(int n, const int *I, const int *J, int alpha, const int *x,
                                    int beta, int *y)
{
    int i, jj;
    if (alpha == 0) { for (i = 0; i < n; i++) y[i] *= beta; return; }
    int temp = beta / alpha;
    if (temp != 1) {
        if (temp == 0) for (i = 0; i < n; i++) y[i] = 0;
        else           for (i = 0; i < n; i++) y[i] *= temp;
    }
    for (i = 0; i < n; i++) {
        int t = 0;
        for (jj = I[i]; jj < I[i + 1]; jj++) t += x[J[jj]];
        y[i] += t;
    }
    if (alpha != 1) for (i = 0; i < n; i++) y[i] *= alpha;
}

/* pack the send buffer: x_buf[k] = x[send_map_elmts[k]]  (hypre_ParCSRMatrixMatvecBoolInt.c:178-187) */
ORC_API void orc_gather(int n, const int *elmts, const double *x, double *buf)
{ for (int k = 0; k < n; k++) buf[k] = x[elmts[k]]; }

/* reverse halo: y[send_map_elmts[k]] += y_buf[k]  (same file :316-323) */
ORC_API void orc_scatter_add(int n, const int *elmts, const double *buf, double *y)
{ for (int k = 0; k < n; k++) y[elmts[k]] += buf[k]; }

/* ------------------------------------------------------------------------------------------------
 * hypre_ParCSRComputeL1Norms(A, option, NULL, &l1)  [3P hypre parcsr_ls/ams.c], reached from
 * mfem::HypreSmoother::SetOperator for types 1-4 (ParELAG_HypreSmootherWrapper.cpp:28-34).
 * option 1: sum|diag row| + sum|offd row|; 2: |a_ii| + sum|offd row|; 3: sum a^2;
 * option 4: d=|a_ii|, l = d + 0.5*sum|offd|, l<=4/3 d -> d.  Then l1 = -l1 where a_ii<0 (a_ii is the
 * first stored entry of the diag row).  Returns the number of (near) zero norms (an error in hypre).
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_l1_norms(int n, const int *dI, const int *dJ, const double *da,
                         const int *oI, const double *oa, int option, double *l1)
{
    int i, j, bad = 0;
    for (i = 0; i < n; i++) {
        double s = 0.0;
        if (option == 1) {
            for (j = dI[i]; j < dI[i + 1]; j++) s += fabs(da[j]);
            if (oI) for (j = oI[i]; j < oI[i + 1]; j++) s += fabs(oa[j]);
        } else if (option == 2) {
            for (j = dI[i]; j < dI[i + 1]; j++) if (dJ[j] == i) s += fabs(da[j]);
            if (oI) for (j = oI[i]; j < oI[i + 1]; j++) s += fabs(oa[j]);
        } else if (option == 3) {
            for (j = dI[i]; j < dI[i + 1]; j++) s += da[j] * da[j];
            if (oI) for (j = oI[i]; j < oI[i + 1]; j++) s += oa[j] * oa[j];
        } else if (option == 4) {
            double d = 0.0, o = 0.0;
            for (j = dI[i]; j < dI[i + 1]; j++) if (dJ[j] == i) d += fabs(da[j]);
            if (oI) for (j = oI[i]; j < oI[i + 1]; j++) o += 0.5 * fabs(oa[j]);
            s = d + o;
            if (s <= 4.0 / 3.0 * d) s = d;
        }
        l1[i] = s;
    }
    for (i = 0; i < n; i++) {
        if (dI[i + 1] > dI[i] && da[dI[i]] < 0.0) l1[i] = -l1[i];
        if (fabs(l1[i]) < 2.220446049250313e-16) bad++;
    }
    return bad;
}

/* u[i] += v[i]/l1[i] -- tail of hypre_ParCSRRelax relax_type 1 (l1-Jacobi) [3P par_relax_more.c];
 * the preceding step v = -w*A*u + w*f is a ParCSR matvec done by the caller. */
ORC_API void orc_jacobi_update(int n, const double *v, const double *l1, double *u)
{ ORC_PFOR for (int i = 0; i < n; i++) u[i] += v[i] / l1[i]; }

/* hypre_ParCSRRelax relax_type 2/4 (l1 Gauss-Seidel, the reference's shipped default smoother:
 * examples/testing_helpers/Create0FormParameterList.hpp:103-116), relax_weight = omega = 1 branch:
 * ghost values frozen for the sweep, forward then backward over local rows. */
ORC_API void orc_l1gs_sweep_omega(int n, const int *dI, const int *dJ, const double *da,
                                  const int *oI, const int *oJ, const double *oa,
                                  const double *f, const double *l1, const double *uext, double omega, double *u)
{
    /* relax_weight = 1, omega != 1 branch of the same relaxation: u_i += omega * res / l1_i */
    int i, jj;
    for (i = 0; i < n; i++) {
        if (l1[i] != 0.0) {
            double res = f[i];
            for (jj = dI[i]; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            u[i] += omega * res / l1[i];
        }
    }
    for (i = n - 1; i > -1; i--) {
        if (l1[i] != 0.0) {
            double res = f[i];
            for (jj = dI[i]; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            u[i] += omega * res / l1[i];
        }
    }
}

/* hypre_ParCSRRelax relax_type 3 (Kaczmarz; "Kaczmarz" in src/linalg/factories/ParELAG_HypreSmootherFactory.cpp:95,113) [3P par_relax_more.c]:
 * l1 = squared 2-norms of the rows (hypre_ParCSRComputeL1Norms option 3); per row, forward then backward over the local rows with the
 * ghost values frozen: res = (f_i - a_i . u) / l1_i, then u_j += omega * res * a_ij for the local columns of the row. */
ORC_API void orc_kaczmarz_sweep(int n, const int *dI, const int *dJ, const double *da,
                                const int *oI, const int *oJ, const double *oa,
                                const double *f, const double *l1, const double *uext, double omega, double *u)
{
    int i, jj, pass;
    for (pass = 0; pass < 2; pass++)
        for (i = pass ? n - 1 : 0; pass ? i > -1 : i < n; i += pass ? -1 : 1) {
            double res = f[i];
            for (jj = dI[i]; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            res /= l1[i];
            for (jj = dI[i]; jj < dI[i + 1]; jj++) u[dJ[jj]] += omega * res * da[jj];
        }
}

ORC_API void orc_l1gs_sweep(int n, const int *dI, const int *dJ, const double *da,
                            const int *oI, const int *oJ, const double *oa,
                            const double *f, const double *l1, const double *uext, double *u)
{
    int i, jj;
    for (i = 0; i < n; i++) {
        if (l1[i] != 0.0) {
            double res = f[i];
            for (jj = dI[i]; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            u[i] += res / l1[i];
        }
    }
    for (i = n - 1; i > -1; i--) {
        if (l1[i] != 0.0) {
            double res = f[i];
            for (jj = dI[i]; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            u[i] += res / l1[i];
        }
    }
}

/* hypre_BoomerAMGRelax relax_type 6 [3P parcsr_ls/par_relax.c], reached from hypre_ParCSRRelax's default branch for
 * mfem::HypreSmoother::GS = 6 ("Gauss-Seidel", ParELAG_HypreSmootherFactory.cpp:99), relax_points = 0, relax_weight = omega = 1,
 * one thread: hybrid symmetric Gauss-Seidel -- ghost values frozen for the sweep, forward then backward over the local rows,
 * u_i = (f_i - sum_{j != i} a_ij u_j) / a_ii with a_ii the FIRST stored entry of the diag row (hypre's storage convention),
 * rows with a zero diagonal left alone. */
ORC_API void orc_gs_sweep(int n, const int *dI, const int *dJ, const double *da,
                          const int *oI, const int *oJ, const double *oa,
                          const double *f, const double *uext, double *u)
{
    int i, jj;
    for (i = 0; i < n; i++) {
        if (dI[i + 1] > dI[i] && da[dI[i]] != 0.0) {
            double res = f[i];
            for (jj = dI[i] + 1; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            u[i] = res / da[dI[i]];
        }
    }
    for (i = n - 1; i > -1; i--) {
        if (dI[i + 1] > dI[i] && da[dI[i]] != 0.0) {
            double res = f[i];
            for (jj = dI[i] + 1; jj < dI[i + 1]; jj++) res -= da[jj] * u[dJ[jj]];
            if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
            u[i] = res / da[dI[i]];
        }
    }
}

/* hypre_ParCSRRelax relax_type 2/4, relax_weight != 1: the "scaled SSOR" branch [3P parcsr_ls/ams.c] (c1 = omega w,
 * c2 = omega (1 - w) and the saved pre-sweep vector; the reference forwards "Damping Factor" / "Omega" unconditionally,
 * src/linalg/solver_ops/ParELAG_HypreSmootherWrapper.cpp:28-33).  v receives the vector the forward pass started from (hypre saves
 * v_i = u_i when row i's turn comes, i.e. before that row is written), and BOTH passes read it:
 *   forward   dif = sum_{j < i} a_ij (v_j - u_j)      backward   dif = sum_{j > i} a_ij (v_j - u_j)
 *   u_i += (c1 res + c2 dif) / l1_i,   res = f_i - sum_diag a_ij u_j - sum_offd a_ij ughost_j   (ghost values frozen for the sweep) */
ORC_API void orc_l1gs_sweep_scaled(int n, const int *dI, const int *dJ, const double *da,
                                   const int *oI, const int *oJ, const double *oa,
                                   const double *f, const double *l1, const double *uext, double weight, double omega, double *u, double *v)
{
    const double c1 = omega * weight, c2 = omega * (1.0 - weight);
    int i, jj;
    for (i = 0; i < n; i++) {
        double dif = 0.0, res = f[i];
        v[i] = u[i];
        if (l1[i] == 0.0) continue;
        for (jj = dI[i]; jj < dI[i + 1]; jj++) {
            res -= da[jj] * u[dJ[jj]];
            if (dJ[jj] < i) dif += da[jj] * (v[dJ[jj]] - u[dJ[jj]]);
        }
        if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
        u[i] += (c1 * res + c2 * dif) / l1[i];
    }
    for (i = n - 1; i > -1; i--) {
        double dif = 0.0, res = f[i];
        if (l1[i] == 0.0) continue;
        for (jj = dI[i]; jj < dI[i + 1]; jj++) {
            res -= da[jj] * u[dJ[jj]];
            if (dJ[jj] > i) dif += da[jj] * (v[dJ[jj]] - u[dJ[jj]]);
        }
        if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
        u[i] += (c1 * res + c2 * dif) / l1[i];
    }
}

/* hypre_BoomerAMGRelax relax_type 6, relax_weight != 1 or omega != 1 [3P parcsr_ls/par_relax.c], relax_points = 0, one thread:
 * Vtemp = u once before the forward pass, prod = 1 - w omega, then forward and backward over the local rows with a nonzero diagonal
 *   res0 = - sum_{j != i} a_ij u_j,  res2 = + sum_{j != i} a_ij Vtemp_j,  res = f_i - sum_offd a_ij ughost_j
 *   u_i = prod u_i + w (omega res + res0 + (1 - omega) res2) / a_ii          (a_ii = the first stored entry of the diag row) */
ORC_API void orc_gs_sweep_sor(int n, const int *dI, const int *dJ, const double *da,
                              const int *oI, const int *oJ, const double *oa,
                              const double *f, const double *uext, double weight, double omega, double *u, double *vtemp)
{
    const double prod = 1.0 - weight * omega, one_minus_omega = 1.0 - omega;
    int i, jj, pass;
    for (i = 0; i < n; i++) vtemp[i] = u[i];
    for (pass = 0; pass < 2; pass++)
        for (i = pass ? n - 1 : 0; pass ? i > -1 : i < n; i += pass ? -1 : 1) {
            if (dI[i + 1] > dI[i] && da[dI[i]] != 0.0) {
                double res0 = 0.0, res2 = 0.0, res = f[i];
                for (jj = dI[i] + 1; jj < dI[i + 1]; jj++) {
                    res0 -= da[jj] * u[dJ[jj]];
                    res2 += da[jj] * vtemp[dJ[jj]];
                }
                if (oI) for (jj = oI[i]; jj < oI[i + 1]; jj++) res -= oa[jj] * uext[oJ[jj]];
                u[i] *= prod;
                u[i] += weight * (omega * res + res0 + one_minus_omega * res2) / da[dI[i]];
            }
        }
}

/* HYPRE_ParCSRDiagScale(NULL, A, y, x) [3P parcsr_ls/HYPRE_parcsr_pcg.c]: x_i = y_i / a_ii (first stored entry), whatever x held --
 * mfem::HypreDiagScale, the solver behind "Hypre Jacobi" = 413 (ParELAG_HypreSmootherFactory.cpp:28-31). */
ORC_API void orc_diag_scale(int n, const int *dI, const double *da, const double *y, double *x)
{ for (int i = 0; i < n; i++) x[i] = y[i] / da[dI[i]]; }

/* ------------------------------------------------------------------------------------------------
 * Chebyshev coefficients of hypre_ParCSRRelax_Cheby(A,f,max,min,fraction,order,scale=1,variant=0,...)
 * [3P par_cheby.c], reached with "Cheby Poly Order"/"Cheby Poly Fraction"
 * (ParELAG_HypreSmootherWrapper.cpp:32-33).  Returns cheby_order = order-1; coefs[0..cheby_order].
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_cheby_coefs(double max_eig, double min_eig, double fraction, int order, double *coefs)
{
    if (order > 4) order = 4;
    if (order < 1) order = 1;
    int co = order - 1;
    double ub = max_eig * 1.1;
    double lb = (ub - min_eig) * fraction + min_eig;
    double theta = (ub + lb) / 2, delta = (ub - lb) / 2, den;
    switch (co) {
    case 0: coefs[0] = 1.0 / theta; break;
    case 1: den = delta * delta - 2 * theta * theta;
            coefs[0] = -4 * theta / den; coefs[1] = 2 / den; break;
    case 2: den = 3 * (delta * delta) * theta - 4 * (theta * theta * theta);
            coefs[0] = (3 * delta * delta - 12 * theta * theta) / den;
            coefs[1] = 12 * theta / den; coefs[2] = -4 / den; break;
    case 3: den = pow(delta, 4) - 8 * delta * delta * theta * theta + 8 * pow(theta, 4);
            coefs[0] = (32 * pow(theta, 3) - 16 * delta * delta * theta) / den;
            coefs[1] = (8 * delta * delta - 48 * theta * theta) / den;
            coefs[2] = 32 * theta / den; coefs[3] = -8 / den; break;
    }
    return co;
}

/* ds[j] = 1/sqrt(a_jj), a_jj = first stored entry of the diag row (A.0 convention) */
ORC_API void orc_inv_sqrt_diag(int n, const int *dI, const double *da, double *ds)
{ for (int j = 0; j < n; j++) ds[j] = 1.0 / sqrt(da[dI[j]]); }

/* vector pieces of the scaled Chebyshev sweep (the matvecs in between are done by the caller) */
ORC_API void orc_cheby_r(int n, const double *ds, const double *f, const double *tmp, double *r)
{ ORC_PFOR for (int j = 0; j < n; j++) r[j] = ds[j] * (f[j] + tmp[j]); }
ORC_API void orc_cheby_start(int n, double c, const double *r, double *u, double *orig)
{ ORC_PFOR for (int j = 0; j < n; j++) { orig[j] = u[j]; u[j] = r[j] * c; } }
ORC_API void orc_vec_mul(int n, const double *a, const double *b, double *out)
{ ORC_PFOR for (int j = 0; j < n; j++) out[j] = a[j] * b[j]; }
ORC_API void orc_cheby_step(int n, double c, const double *r, const double *ds, const double *v, double *u)
{ ORC_PFOR for (int j = 0; j < n; j++) { double t = ds[j] * v[j]; u[j] = c * r[j] + t; } }
ORC_API void orc_cheby_finish(int n, const double *orig, const double *ds, double *u)
{ ORC_PFOR for (int j = 0; j < n; j++) u[j] = orig[j] + ds[j] * u[j]; }

/* hypre_Rand / hypre_SeqVectorSetRandomValues [3P utilities/random.c]: Lehmer LCG a=16807,
 * m=2^31-1 via Schrage (q=127773, r=2836); value = 2*rand-1.  Used for the CG eigen-estimate start. */
ORC_API void orc_hypre_random(int n, int seed, double *v)
{
    const int a = 16807, m = 2147483647, q = 127773, r = 2836;
    int s = seed;
    for (int i = 0; i < n; i++) {
        int low = s % q, high = s / q;
        int test = a * low - r * high;
        s = (test > 0) ? test : test + m;
        v[i] = 2.0 * ((double)s / m) - 1.0;
    }
}

/* Eigenvalues of the symmetric tridiagonal (d[0..n-1], e[1..n-1]) by implicit QL (EISPACK tql1, the
 * routine hypre_LINPACKcgtql1 wraps); d is overwritten with ascending eigenvalues.  Returns 0 on ok. */
ORC_API int orc_tql1(int n, double *d, double *e)
{
    int i, j, l, m, ii;
    double c, c2, c3 = 0, dl1, el1, f, g, h, p, r, s, s2 = 0, tst1, tst2;
    if (n == 1) return 0;
    for (i = 1; i < n; i++) e[i - 1] = e[i];
    f = 0.0; tst1 = 0.0; e[n - 1] = 0.0;
    for (l = 0; l < n; l++) {
        j = 0;
        h = fabs(d[l]) + fabs(e[l]);
        if (tst1 < h) tst1 = h;
        for (m = l; m < n; m++) { tst2 = tst1 + fabs(e[m]); if (tst2 == tst1) break; }
        if (m != l) {
            do {
                if (j == 30) return l + 1;
                j++;
                int l1 = l + 1, l2 = l1 + 1;
                g = d[l];
                p = (d[l1] - g) / (2.0 * e[l]);
                r = hypot(p, 1.0);
                d[l] = e[l] / (p + copysign(r, p));
                d[l1] = e[l] * (p + copysign(r, p));
                dl1 = d[l1];
                h = g - d[l];
                for (i = l2; i < n; i++) d[i] -= h;
                f += h;
                p = d[m]; c = 1.0; c2 = c; el1 = e[l1]; s = 0.0;
                for (ii = 1; ii <= m - l; ii++) {
                    c3 = c2; c2 = c; s2 = s;
                    i = m - ii;
                    g = c * e[i];
                    h = c * p;
                    r = hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
                tst2 = tst1 + fabs(e[l]);
            } while (tst2 > tst1);
        }
        p = d[l] + f;
        /* order eigenvalues */
        for (i = l; i > 0; i--) { if (p >= d[i - 1]) break; d[i] = d[i - 1]; }
        d[i] = p;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * hypre_ParCSRMatrixFixZeroRows [3P], call sites Hierarchy.cpp:366-371, HiptmairSmootherFactory.cpp:162
 * rows with sum|a| (diag+offd) <= 0: diagonal entry <- 1, everything else <- 0.
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_fix_zero_rows(int n, const int *dI, const int *dJ, double *da, const int *oI, double *oa)
{
    int i, j, nfixed = 0;
    for (i = 0; i < n; i++) {
        double l1 = 0.0;
        for (j = dI[i]; j < dI[i + 1]; j++) l1 += fabs(da[j]);
        if (oI) for (j = oI[i]; j < oI[i + 1]; j++) l1 += fabs(oa[j]);
        if (l1 <= 0.0) {
            for (j = dI[i]; j < dI[i + 1]; j++) da[j] = (dJ[j] == i) ? 1.0 : 0.0;
            if (oI) for (j = oI[i]; j < oI[i + 1]; j++) oa[j] = 0.0;
            nfixed++;
        }
    }
    return nfixed;
}

/* ------------------------------------------------------------------------------------------------
 * Sparse transpose (hypre_CSRMatrixTranspose(A,&AT,1) / mfem::Transpose): counting sort, so each
 * transposed row lists its entries by ascending source row.  par_Tmatmul.c:41-79 is the call site.
 * ---------------------------------------------------------------------------------------------- */
ORC_API void orc_csr_transpose(int nrows, int ncols, const int *I, const int *J, const double *a,
                               int *TI, int *TJ, double *Ta)
{
    int i, jj;
    int nnz = I[nrows];
    memset(TI, 0, sizeof(int) * (size_t)(ncols + 1));
    for (jj = 0; jj < nnz; jj++) TI[J[jj] + 1]++;
    for (i = 0; i < ncols; i++) TI[i + 1] += TI[i];
    int *pos = (int *)malloc(sizeof(int) * (size_t)(ncols > 0 ? ncols : 1));
    memcpy(pos, TI, sizeof(int) * (size_t)ncols);
    for (i = 0; i < nrows; i++)
        for (jj = I[i]; jj < I[i + 1]; jj++) {
            int p = pos[J[jj]]++;
            TJ[p] = i;
            if (a) Ta[p] = a[jj];
        }
    free(pos);
}

/* ------------------------------------------------------------------------------------------------
 * C = A*B, Gustavson with an integer marker, two passes (count, then fill); column order within a
 * row is first-touch.  Pattern of hypre_CSRMatrixMultiply / mfem::Mult(SparseMatrix,SparseMatrix),
 * which mfem::RAP(Rt,A,P) (examples/MultigridTest0Form.cpp:454) chains twice; the same marker idiom is
 * in-repo at src/hypreExtension/parcsr-add.c:84-196.
 * Pass 1: CI (size nrowsA+1).  Pass 2 (CJ, Ca non-NULL): fill.
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_csr_matmul_count(int nrowsA, int ncolsB, const int *AI, const int *AJ,
                                 const int *BI, const int *BJ, int *CI)
{
    int ic, ia, ib, nnz = 0;
    int *marker = (int *)malloc(sizeof(int) * (size_t)(ncolsB > 0 ? ncolsB : 1));
    for (ic = 0; ic < ncolsB; ic++) marker[ic] = -1;
    CI[0] = 0;
    for (ic = 0; ic < nrowsA; ic++) {
        for (ia = AI[ic]; ia < AI[ic + 1]; ia++) {
            int k = AJ[ia];
            for (ib = BI[k]; ib < BI[k + 1]; ib++)
                if (marker[BJ[ib]] != ic) { marker[BJ[ib]] = ic; nnz++; }
        }
        CI[ic + 1] = nnz;
    }
    free(marker);
    return nnz;
}

ORC_API void orc_csr_matmul_fill(int nrowsA, int ncolsB, const int *AI, const int *AJ, const double *Aa,
                                 const int *BI, const int *BJ, const double *Ba,
                                 const int *CI, int *CJ, double *Ca)
{
    int ic, ia, ib, pos = 0;
    int *marker = (int *)malloc(sizeof(int) * (size_t)(ncolsB > 0 ? ncolsB : 1));
    for (ic = 0; ic < ncolsB; ic++) marker[ic] = -1;
    for (ic = 0; ic < nrowsA; ic++) {
        int row_begin = CI[ic];
        for (ia = AI[ic]; ia < AI[ic + 1]; ia++) {
            int k = AJ[ia];
            double av = Aa[ia];
            for (ib = BI[k]; ib < BI[k + 1]; ib++) {
                int c = BJ[ib];
                if (marker[c] < row_begin) { marker[c] = pos; CJ[pos] = c; Ca[pos] = av * Ba[ib]; pos++; }
                else Ca[marker[c]] += av * Ba[ib];
            }
        }
    }
    free(marker);
}

/* ------------------------------------------------------------------------------------------------
 * RAP = RT^T * A * P, fused row-wise triple loop of hypre_BoomerAMGBuildCoarseOperator [3P par_rap.c]
 * restricted to the single-rank blocks (RAP_diag from R_diag, A_diag, P_diag); reached through
 * mfem::RAP(A,P) at Hierarchy.cpp:365 and through IgnoreNonLocalRange (MatrixUtils.cpp:878-916), which
 * empties RT's offd so that exactly this loop remains.  R must be the transpose of RT
 * (orc_csr_transpose).  For a square result the diagonal entry is emitted first; other columns are in
 * first-touch order.  Two passes: count (RJ/Ra NULL) then fill.  `nfine` is the number of COLUMNS of A
 * (hypre sizes A_marker by num_cols_diag_A + num_cols_offd_A), so rectangular A -- the off-diagonal blocks of the
 * blocked hierarchy, Hierarchy.cpp:511-513 -- are covered.
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_rap(int nc_rows, int nfine, int nc_cols, int square,
                    const int *RI, const int *RJ, const double *Ra,
                    const int *AI, const int *AJ, const double *Aa,
                    const int *PI, const int *PJ, const double *Pa,
                    int *CI, int *CJ, double *Ca)
{
    int ic, i1, i2, i3, jj1, jj2, jj3, jj_counter = 0;
    int fill = (CJ != NULL);
    int *P_marker = (int *)malloc(sizeof(int) * (size_t)(nc_cols > 0 ? nc_cols : 1));
    int *A_marker = (int *)malloc(sizeof(int) * (size_t)(nfine > 0 ? nfine : 1));
    for (ic = 0; ic < nc_cols; ic++) P_marker[ic] = -1;
    for (i1 = 0; i1 < nfine; i1++) A_marker[i1] = -1;
    if (!fill) CI[0] = 0;
    for (ic = 0; ic < nc_rows; ic++) {
        int jj_row_begin = jj_counter;
        if (square) {
            P_marker[ic] = jj_counter;
            if (fill) { Ca[jj_counter] = 0.0; CJ[jj_counter] = ic; }
            jj_counter++;
        }
        for (jj1 = RI[ic]; jj1 < RI[ic + 1]; jj1++) {
            i1 = RJ[jj1];
            double r_entry = fill ? Ra[jj1] : 0.0;
            for (jj2 = AI[i1]; jj2 < AI[i1 + 1]; jj2++) {
                i2 = AJ[jj2];
                double r_a = fill ? r_entry * Aa[jj2] : 0.0;
                if (A_marker[i2] != ic) {
                    A_marker[i2] = ic;
                    for (jj3 = PI[i2]; jj3 < PI[i2 + 1]; jj3++) {
                        i3 = PJ[jj3];
                        if (P_marker[i3] < jj_row_begin) {
                            P_marker[i3] = jj_counter;
                            if (fill) { Ca[jj_counter] = r_a * Pa[jj3]; CJ[jj_counter] = i3; }
                            jj_counter++;
                        } else if (fill) Ca[P_marker[i3]] += r_a * Pa[jj3];
                    }
                } else if (fill) {
                    for (jj3 = PI[i2]; jj3 < PI[i2 + 1]; jj3++)
                        Ca[P_marker[PJ[jj3]]] += r_a * Pa[jj3];
                }
            }
        }
        if (!fill) CI[ic + 1] = jj_counter;
    }
    free(P_marker); free(A_marker);
    return jj_counter;
}

/* ------------------------------------------------------------------------------------------------
 * C = a*A + b*B : hypre_CSRMatrixAdd2 verbatim semantics (src/hypreExtension/parcsr-add.c:84-196).
 * ---------------------------------------------------------------------------------------------- */
ORC_API int orc_csr_add_count(int nrows, int ncols, const int *AI, const int *AJ,
                              const int *BI, const int *BJ, int *CI)
{
    int ia, ib, ic, nnz = 0;
    int *marker = (int *)malloc(sizeof(int) * (size_t)(ncols > 0 ? ncols : 1));
    for (ia = 0; ia < ncols; ia++) marker[ia] = -1;
    CI[0] = 0;
    for (ic = 0; ic < nrows; ic++) {
        for (ia = AI[ic]; ia < AI[ic + 1]; ia++) { marker[AJ[ia]] = ic; nnz++; }
        for (ib = BI[ic]; ib < BI[ic + 1]; ib++)
            if (marker[BJ[ib]] != ic) { marker[BJ[ib]] = ic; nnz++; }
        CI[ic + 1] = nnz;
    }
    free(marker);
    return nnz;
}

ORC_API void orc_csr_add_fill(int nrows, int ncols, double a, const int *AI, const int *AJ, const double *Aa,
                              double b, const int *BI, const int *BJ, const double *Ba,
                              const int *CI, int *CJ, double *Ca)
{
    int ia, ib, ic, pos = 0;
    int *marker = (int *)malloc(sizeof(int) * (size_t)(ncols > 0 ? ncols : 1));
    for (ia = 0; ia < ncols; ia++) marker[ia] = -1;
    for (ic = 0; ic < nrows; ic++) {
        for (ia = AI[ic]; ia < AI[ic + 1]; ia++) {
            CJ[pos] = AJ[ia]; Ca[pos] = a * Aa[ia]; marker[AJ[ia]] = pos; pos++;
        }
        for (ib = BI[ic]; ib < BI[ic + 1]; ib++) {
            int jcol = BJ[ib];
            if (marker[jcol] < CI[ic]) { CJ[pos] = jcol; Ca[pos] = b * Ba[ib]; marker[jcol] = pos; pos++; }
            else Ca[marker[jcol]] += b * Ba[ib];
        }
    }
    free(marker);
}

/* hypre_CSRMatrixDeleteZeros(A,tol) [3P] as used by deleteZeros.c:16-47: drop entries with |a| <= tol.
 * Returns the new nnz; outputs may alias nothing. */
ORC_API int orc_csr_delete_zeros(int nrows, const int *I, const int *J, const double *a, double tol,
                                 int *OI, int *OJ, double *Oa)
{
    int i, j, pos = 0;
    OI[0] = 0;
    for (i = 0; i < nrows; i++) {
        for (j = I[i]; j < I[i + 1]; j++)
            if (fabs(a[j]) > tol) { OJ[pos] = J[j]; Oa[pos] = a[j]; pos++; }
        OI[i + 1] = pos;
    }
    return pos;
}

/* hypre_CSRDataTransformationSign (hypre_ParCSRDataTransformationSign.c:16-27) */
ORC_API void orc_sign_transform(int nnz, double *a)
{ for (int i = 0; i < nnz; i++) a[i] = (fabs(a[i]) > 1e-9) ? ((a[i] > 0.) ? 1.0 : -1.0) : 0.0; }

/* hypre_CSRMatrixMaxNorm / FrobeniusNorm (hypre_ParCSRMatrixNorms.c:162-195) */
ORC_API double orc_csr_max_norm(int nnz, const double *a)
{ double m = 0, h; for (int i = 0; i < nnz; i++) if ((h = fabs(a[i])) > m) m = h; return m; }
ORC_API double orc_csr_frobenius_norm(int nnz, const double *a)
{ double s = 0; for (int i = 0; i < nnz; i++) s += a[i] * a[i]; return sqrt(s); }
/* local part of Normlinf / Norml1 (:18-127): max over entries of |A| 1 resp. |A|^T 1 */
ORC_API void orc_abs(int nnz, const double *a, double *out)
{ for (int i = 0; i < nnz; i++) out[i] = fabs(a[i]); }

/* ------------------------------------------------------------------------------------------------
 * Agglomerate index maps (must be BIT-EXACT).
 * ---------------------------------------------------------------------------------------------- */
/* MFEMRefinedMeshPartitioner::Partition, MFEM >= 4.1 branch (MFEMRefinedMeshPartitioner.cpp:64-70) */
ORC_API void orc_partition_refined(int nElements, int nParts, int *p)
{
    int cf = nElements / nParts, j = 0;
    for (int i = 0; i < nParts; ++i) for (int k = 0; k < cf; ++k) p[j++] = i;
}
/* older-MFEM branch (:71-91): parents keep their index, children appended pass by pass */
ORC_API void orc_partition_refined_legacy(int nElements, int nParts, int nDim, int *p)
{
    int nsplits = 1 << nDim, cf = nElements / nParts, npass = 0, tmp = cf;
    while (tmp > 1) { tmp /= nsplits; ++npass; }
    for (int i = 0; i < nParts; ++i) p[i] = i;
    int len = nParts;
    for (int i = 0; i < npass; ++i) {
        int *x = p + len;
        for (int k = 0; k < len; ++k) for (int j = 0; j < nsplits - 1; ++j) *x++ = p[k];
        len *= nsplits;
    }
}

/* connectedComponents (src/structures/connectedComponents.cpp:23-85): stack BFS in node order,
 * new id = offset[part] + component.  Returns the number of components. */
ORC_API int orc_connected_components(int n, int *partitioning, const int *I, const int *J)
{
    if (n == 0) return 0;
    int i, j, k, node, num_part = 0;
    for (i = 0; i < n; i++) if (partitioning[i] + 1 > num_part) num_part = partitioning[i] + 1;
    int *component = (int *)malloc(sizeof(int) * (size_t)n);
    int *offset = (int *)calloc((size_t)num_part + 1, sizeof(int));
    int *num_comp = offset + 1;
    int *stack = (int *)malloc(sizeof(int) * (size_t)n);
    for (i = 0; i < n; i++) component[i] = -1;
    int stack_p = 0, stack_top = 0;
    for (node = 0; node < n; node++) {
        if (partitioning[node] < 0) continue;
        if (component[node] >= 0) continue;
        component[node] = num_comp[partitioning[node]]++;
        stack[stack_top++] = node;
        for (; stack_p < stack_top; stack_p++) {
            i = stack[stack_p];
            if (partitioning[i] < 0) continue;
            for (j = I[i]; j < I[i + 1]; j++) {
                k = J[j];
                if (partitioning[k] == partitioning[i] && component[k] < 0) {
                    component[k] = component[i];
                    stack[stack_top++] = k;
                }
            }
        }
    }
    for (i = 0; i < num_part; i++) offset[i + 1] += offset[i];
    /* nodes with a negative id were skipped by the sweep (:50-51, :62-63); the reference's last loop (:79-80) then reads
     * offset_comp[-1] for them, which is undefined -- they keep their id here */
    for (i = 0; i < n; ++i) if (partitioning[i] >= 0) partitioning[i] = offset[partitioning[i]] + component[i];
    int total = offset[num_part];
    free(component); free(offset); free(stack);
    return total;
}

/* transposePartitioning (src/structures/transpose.cpp:27-55): element->part array to part->elements CSR
 * (entries with part -1 skipped).  i_out has nparts+1 entries. */
ORC_API void orc_transpose_partitioning(int n, const int *partitioning, int nparts, int *i_out, int *j_out)
{
    int i;
    for (i = 0; i <= nparts; i++) i_out[i] = 0;
    for (i = 0; i < n; i++) if (partitioning[i] > -1) i_out[partitioning[i] + 1]++;
    for (i = 0; i < nparts; i++) i_out[i + 1] += i_out[i];
    int *cnt = (int *)malloc(sizeof(int) * (size_t)(nparts > 0 ? nparts : 1));
    memcpy(cnt, i_out, sizeof(int) * (size_t)nparts);
    for (i = 0; i < n; i++) { int r = partitioning[i]; if (r > -1) j_out[cnt[r]++] = i; }
    free(cnt);
}

/* ------------------------------------------------------------------------------------------------
 * Serial V-cycle driver: Hierarchy::Mult (preconditioner mode) -> Iterate
 * (src/linalg/solver_ops/ParELAG_Hierarchy.cpp:109-253) with HypreSmootherWrapper smoothers of
 * relax type 1 (l1-Jacobi), 2 (l1-GS) or 16 (Chebyshev), iterative_mode=true inside the cycle
 * (AMGeSolverFactory.cpp:136-137), optional Hiptmair wrapping (ParELAG_HiptmairSmoother.cpp:48-76),
 * and a dense Cholesky coarse solve (stands for the "Coarse solver" factory; a direct solver is a
 * legal choice there: AMGeSolverFactory.cpp:144-163).  Single rank: offd blocks are empty, so every
 * ParCSR call reduces to its diag kernel.  Used by tests and by bench.py's cpu_baseline leg.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int n;
    const int *I, *J; const double *a;
} orc_csr;

typedef struct {
    int type;              /* 1, 2, 4, 6, 16 or 413 */
    int sweeps;
    double weight;         /* "Damping Factor" */
    const double *l1;      /* types 1, 2, 4 */
    int cheby_order;       /* = order-1 (type 16) */
    double coefs[5];
    const double *ds;      /* type 16 */
} orc_smoother;

typedef struct {
    orc_csr A;
    orc_csr P;             /* maps level l+1 -> l (unused on the coarsest level); P.n = rows */
    int ncoarse;
    orc_smoother sm;
    /* Hiptmair (optional) */
    int hiptmair;
    orc_csr D;  int naux;  orc_csr Aaux;  orc_smoother sm_aux;
    double *tmp_resid, *tmp_correct, *resid, *sol, *v, *z, *r, *auxb, *auxx, *pv, *auxv, *auxz, *auxr;
    /* multi-core build only: explicit transposes so that P^T r / D^T r are row-parallel gathers */
    int *PTI, *PTJ, *DTI, *DTJ; double *PTa, *DTa;
} orc_level;

typedef struct {
    int nlevels;
    orc_level *lev;
    int ncoarse; const double *chol;   /* dense lower Cholesky factor, row-major ncoarse x ncoarse */
    const double *spmd_l1[64];         /* orc_hier_spmd_bench with type-2 levels: the l1 norms of the partition it runs on */
} orc_hier;

static void smoother_apply(const orc_csr *A, const orc_smoother *s, const double *f, double *u,
                           double *v, double *z, double *r, int zero_guess)
{
    int n = A->n, i, sw;
    if (zero_guess) memset(u, 0, sizeof(double) * (size_t)n);
    for (sw = 0; sw < s->sweeps; sw++) {
        if (s->type == 1 || s->type == 5) {        /* 5 = lumped Jacobi: mfem::HypreSmoother turns it into type 1 with l1 = A 1 */
            memcpy(v, f, sizeof(double) * (size_t)n);
            orc_csr_matvec(n, A->I, A->J, A->a, -s->weight, u, s->weight, v);
            orc_jacobi_update(n, v, s->l1, u);
        } else if (s->type == 2 || s->type == 4) {
            orc_l1gs_sweep(n, A->I, A->J, A->a, NULL, NULL, NULL, f, s->l1, NULL, u);
        } else if (s->type == 6) {
            orc_gs_sweep(n, A->I, A->J, A->a, NULL, NULL, NULL, f, NULL, u);
        } else if (s->type == 413) {
            orc_diag_scale(n, A->I, A->a, f, u);
        } else { /* 16 */
            int co = s->cheby_order;
            orc_csr_matvec(n, A->I, A->J, A->a, -1.0, u, 0.0, z);
            orc_cheby_r(n, s->ds, f, z, r);
            orc_cheby_start(n, s->coefs[co], r, u, v);          /* v holds orig_u */
            for (i = co - 1; i >= 0; i--) {
                /* tmp = ds .* u ; w = A*tmp ; u = coef*r + ds .* w.  hypre keeps w in a fourth work
                   vector; here row i of w is consumed as soon as it is formed (same arithmetic:
                   matvec(1.0, A, tmp, 0.0, w) gives w_i = 0 + sum in storage order). */
                double *tmp = z;
                orc_vec_mul(n, s->ds, u, tmp);
                ORC_PFOR
                for (int row = 0; row < n; row++) {
                    double t = 0.0;
                    for (int jj = A->I[row]; jj < A->I[row + 1]; jj++) t += A->a[jj] * tmp[A->J[jj]];
                    double td = s->ds[row] * (0.0 + t);
                    u[row] = s->coefs[i] * r[row] + td;
                }
            }
            orc_cheby_finish(n, v, s->ds, u);
        }
    }
}

/* y = B^T x: serial build = hypre's scatter-add MatvecT; multi-core build = gather product with the stored transpose */
static void level_matvecT(int nrows, int ncols, const orc_csr *B, const int *TI, const int *TJ, const double *Ta,
                          const double *x, double *y)
{
    if (TI) { orc_csr_matvec(ncols, TI, TJ, Ta, 1.0, x, 0.0, y); return; }
    orc_csr_matvecT(nrows, ncols, B->I, B->J, B->a, 1.0, x, 0.0, y);
}

static void level_smooth(orc_level *L, const double *f, double *u, int zero_guess)
{
    smoother_apply(&L->A, &L->sm, f, u, L->v, L->z, L->r, zero_guess);
    if (L->hiptmair) {
        int n = L->A.n, i;
        /* R = B - A X  (mg_utils::ComputeResidual, MG_Utils.hpp:405-426: tmp=0; tmp=A x; tmp-=B; tmp*=-1) */
        memset(L->pv, 0, sizeof(double) * (size_t)n);
        orc_csr_matvec(n, L->A.I, L->A.J, L->A.a, 1.0, u, 0.0, L->pv);
        ORC_PFOR for (i = 0; i < n; i++) { L->pv[i] -= f[i]; L->pv[i] *= -1; }
        memset(L->auxb, 0, sizeof(double) * (size_t)L->naux);
        level_matvecT(n, L->naux, &L->D, L->DTI, L->DTJ, L->DTa, L->pv, L->auxb);
        smoother_apply(&L->Aaux, &L->sm_aux, L->auxb, L->auxx, L->auxv, L->auxz, L->auxr, 1);
        orc_csr_matvec(n, L->D.I, L->D.J, L->D.a, 1.0, L->auxx, 0.0, L->pv);
        ORC_PFOR for (i = 0; i < n; i++) u[i] += L->pv[i];
    }
}

static void level_smooth_T(orc_level *L, const double *f, double *u)
{
    /* HiptmairSmoother::MultTranspose (ParELAG_HiptmairSmoother.cpp:79-109), iterative mode:
       auxiliary correction first, then the primary sweep. */
    if (L->hiptmair) {
        int n = L->A.n, i;
        memset(L->pv, 0, sizeof(double) * (size_t)n);
        orc_csr_matvec(n, L->A.I, L->A.J, L->A.a, 1.0, u, 0.0, L->pv);
        ORC_PFOR for (i = 0; i < n; i++) { L->pv[i] -= f[i]; L->pv[i] *= -1; }
        memset(L->auxb, 0, sizeof(double) * (size_t)L->naux);
        level_matvecT(n, L->naux, &L->D, L->DTI, L->DTJ, L->DTa, L->pv, L->auxb);
        smoother_apply(&L->Aaux, &L->sm_aux, L->auxb, L->auxx, L->auxv, L->auxz, L->auxr, 1);
        orc_csr_matvec(n, L->D.I, L->D.J, L->D.a, 1.0, L->auxx, 0.0, L->pv);
        ORC_PFOR for (i = 0; i < n; i++) u[i] += L->pv[i];
    }
    smoother_apply(&L->A, &L->sm, f, u, L->v, L->z, L->r, 0);
}

ORC_API void orc_chol_solve(int n, const double *Lf, const double *b, double *x)
{
    int i, j;
    for (i = 0; i < n; i++) {
        double s = b[i];
        for (j = 0; j < i; j++) s -= Lf[(size_t)i * n + j] * x[j];
        x[i] = s / Lf[(size_t)i * n + i];
    }
    for (i = n - 1; i >= 0; i--) {
        double s = x[i];
        for (j = i + 1; j < n; j++) s -= Lf[(size_t)j * n + i] * x[j];
        x[i] = s / Lf[(size_t)i * n + i];
    }
}

/* in-place dense Cholesky (lower), row-major; returns 0 ok, k+1 if pivot k is not positive */
ORC_API int orc_chol_factor(int n, double *A)
{
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
        if (d <= 0.0) return j + 1;
        d = sqrt(d);
        A[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
            A[(size_t)i * n + j] = s / d;
        }
        for (int i = 0; i < j; i++) A[(size_t)i * n + j] = 0.0;
    }
    return 0;
}

static void iterate(orc_hier *H, const double *RHS, double *SOL, int l, int symmetric_post)
{
    orc_level *F = &H->lev[l];
    if (l == H->nlevels - 1) {
        if (H->chol) orc_chol_solve(H->ncoarse, H->chol, RHS, SOL);
        else { level_smooth(F, RHS, SOL, 0); if (symmetric_post) level_smooth_T(F, RHS, SOL); else level_smooth(F, RHS, SOL, 0); }
        return;
    }
    int n = F->A.n, nc = F->ncoarse, i;
    level_smooth(F, RHS, SOL, 0);                                   /* :180-184 */
    memset(F->tmp_correct, 0, sizeof(double) * (size_t)n);          /* :191-194 */
    orc_csr_matvec(n, F->A.I, F->A.J, F->A.a, 1.0, SOL, 0.0, F->tmp_correct);
    ORC_PFOR for (i = 0; i < n; i++) F->tmp_resid[i] = RHS[i] + (-1.0) * F->tmp_correct[i];
    double *cr = H->lev[l + 1].resid, *cs = H->lev[l + 1].sol;
    level_matvecT(n, nc, &F->P, F->PTI, F->PTJ, F->PTa, F->tmp_resid, cr);        /* :199-203 */
    memset(cs, 0, sizeof(double) * (size_t)nc);                     /* :212 */
    iterate(H, cr, cs, l + 1, symmetric_post);                      /* :222-227, mu = 0 */
    orc_csr_matvec(n, F->P.I, F->P.J, F->P.a, 1.0, cs, 0.0, F->tmp_correct);       /* :231-234 */
    ORC_PFOR for (i = 0; i < n; i++) SOL[i] += F->tmp_correct[i];   /* :238 */
    if (symmetric_post) level_smooth_T(F, RHS, SOL); else level_smooth(F, RHS, SOL, 0);  /* :244-249 */
}

ORC_API orc_hier *orc_hier_create(int nlevels)
{
    orc_hier *H = (orc_hier *)calloc(1, sizeof(orc_hier));
    H->nlevels = nlevels;
    H->lev = (orc_level *)calloc((size_t)nlevels, sizeof(orc_level));
    return H;
}

static double *dalloc(int n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

ORC_API void orc_hier_set_level(orc_hier *H, int l, int n, const int *AI, const int *AJ, const double *Aa,
                                int ncoarse, const int *PI, const int *PJ, const double *Pa,
                                int type, int sweeps, double weight, const double *l1,
                                int order, double max_eig, double min_eig, double fraction, const double *ds)
{
    orc_level *L = &H->lev[l];
    L->A.n = n; L->A.I = AI; L->A.J = AJ; L->A.a = Aa;
    L->P.n = n; L->P.I = PI; L->P.J = PJ; L->P.a = Pa; L->ncoarse = ncoarse;
    L->sm.type = type; L->sm.sweeps = sweeps; L->sm.weight = weight; L->sm.l1 = l1; L->sm.ds = ds;
    if (type == 16) L->sm.cheby_order = orc_cheby_coefs(max_eig, min_eig, fraction, order, L->sm.coefs);
    L->tmp_resid = dalloc(n); L->tmp_correct = dalloc(n); L->resid = dalloc(n); L->sol = dalloc(n);
    L->v = dalloc(n); L->z = dalloc(n); L->r = dalloc(n);
#ifdef _OPENMP
    if (PI && ncoarse > 0) {
        const int nnz = PI[n];
        L->PTI = (int *)malloc(sizeof(int) * ((size_t)ncoarse + 1)); L->PTJ = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
        L->PTa = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
        orc_csr_transpose(n, ncoarse, PI, PJ, Pa, L->PTI, L->PTJ, L->PTa);
    }
#endif
}

ORC_API void orc_hier_set_hiptmair(orc_hier *H, int l, int naux, const int *DI, const int *DJ, const double *Da,
                                   const int *XI, const int *XJ, const double *Xa,
                                   int type, int sweeps, double weight, const double *l1,
                                   int order, double max_eig, double min_eig, double fraction, const double *ds)
{
    orc_level *L = &H->lev[l];
    L->hiptmair = 1; L->naux = naux;
    L->D.n = L->A.n; L->D.I = DI; L->D.J = DJ; L->D.a = Da;
    L->Aaux.n = naux; L->Aaux.I = XI; L->Aaux.J = XJ; L->Aaux.a = Xa;
    L->sm_aux.type = type; L->sm_aux.sweeps = sweeps; L->sm_aux.weight = weight; L->sm_aux.l1 = l1; L->sm_aux.ds = ds;
    if (type == 16) L->sm_aux.cheby_order = orc_cheby_coefs(max_eig, min_eig, fraction, order, L->sm_aux.coefs);
    L->auxb = dalloc(naux); L->auxx = dalloc(naux); L->pv = dalloc(L->A.n);
    L->auxv = dalloc(naux); L->auxz = dalloc(naux); L->auxr = dalloc(naux);
#ifdef _OPENMP
    {
        const int nnz = DI[L->A.n];
        L->DTI = (int *)malloc(sizeof(int) * ((size_t)naux + 1)); L->DTJ = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
        L->DTa = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
        orc_csr_transpose(L->A.n, naux, DI, DJ, Da, L->DTI, L->DTJ, L->DTa);
    }
#endif
}

ORC_API void orc_hier_set_coarse_chol(orc_hier *H, int n, const double *Lfactor)
{ H->ncoarse = n; H->chol = Lfactor; }

/* Hierarchy::Mult, preconditioner mode (:114-118,130-132): sol = 0; Iterate(rhs, sol, 0). */
ORC_API void orc_hier_apply(orc_hier *H, const double *rhs, double *sol, int symmetric_post)
{
    memset(sol, 0, sizeof(double) * (size_t)H->lev[0].A.n);
    iterate(H, rhs, sol, 0, symmetric_post);
}

ORC_API void orc_hier_destroy(orc_hier *H)
{
    for (int l = 0; l < H->nlevels; l++) {
        orc_level *L = &H->lev[l];
        free(L->tmp_resid); free(L->tmp_correct); free(L->resid); free(L->sol); free(L->v); free(L->z); free(L->r);
        free(L->auxb); free(L->auxx); free(L->pv); free(L->auxv); free(L->auxz); free(L->auxr);
        free(L->PTI); free(L->PTJ); free(L->PTa); free(L->DTI); free(L->DTJ); free(L->DTa);
    }
    free(H->lev); free(H);
}

/* ------------------------------------------------------------------------------------------------
 * CPU baseline in the reference's own layout: one process per core, no OpenMP
 * (CMakeLists.txt:130-132 switches OpenMP off; the examples run one MPI rank per core).  W worker
 * processes are forked; each owns a contiguous row block of every level, like hypre's row
 * partitioning; vectors live in memory shared by the workers, so a rank's "halo exchange" is a read
 * of its neighbours' entries once everyone has passed a barrier -- the cheapest message passing a
 * node can offer (it flatters the CPU against an MPI run).  The arithmetic per row is the serial
 * oracle's (l1-Jacobi levels, dense Cholesky on the coarsest level by rank 0); only P^T r is formed
 * from the explicit transpose, as newer hypre does.  Test infrastructure: bench.py's cpu_baseline.
 * ---------------------------------------------------------------------------------------------- */
#include <sched.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

typedef struct { volatile int count; volatile int sense; char pad[56]; } spmd_barrier_t;
typedef struct {
    spmd_barrier_t bar;
    volatile int stop;
    volatile long cycles;
    volatile double elapsed;
} spmd_ctl;
typedef struct { double *sol, *resid, *v, *tc, *tr; } spmd_vecs;

static void *shared_alloc(size_t bytes)
{
    void *p = mmap(NULL, bytes ? bytes : 8, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    return p == MAP_FAILED ? NULL : p;
}

static inline void spmd_wait(spmd_barrier_t *b, int n, int *sense)
{
    *sense = !*sense;
    if (__sync_add_and_fetch(&b->count, 1) == n) { b->count = 0; __sync_synchronize(); b->sense = *sense; }
    else while (b->sense != *sense) __builtin_ia32_pause();
    __sync_synchronize();
}

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

#define SPMD_LO(n) ((int)((long)(n) * rank / W))
#define SPMD_HI(n) ((int)((long)(n) * (rank + 1) / W))
#define SPMD_BARRIER() spmd_wait(&ctl->bar, W, sense)

/* one l1-Jacobi sweep of this rank's rows: v = w f - w A u; barrier; u += v / l1 (smoother_apply, type 1) */
static void spmd_jacobi(const orc_level *L, const double *f, double *u, double *v, int rank, int W, spmd_ctl *ctl, int *sense)
{
    const int n = L->A.n, lo = SPMD_LO(n), hi = SPMD_HI(n);
    const double w = L->sm.weight;
    for (int i = lo; i < hi; i++) {
        double t = 0.0;
        for (int jj = L->A.I[i]; jj < L->A.I[i + 1]; jj++) t += L->A.a[jj] * u[L->A.J[jj]];
        /* memcpy(v, f); matvec(-w, A, u, w, v): v = (w / -w) v; v += t; v *= -w */
        double y = f[i] * (w / -w);
        y += t;
        v[i] = y * -w;
    }
    SPMD_BARRIER();
    for (int i = lo; i < hi; i++) u[i] += v[i] / L->sm.l1[i];
    SPMD_BARRIER();
}

/* hypre's relax type 2 ("L1 Gauss-Seidel", the reference's shipped smoother: Create0FormParameterList.hpp:103-116) on W ranks is HYBRID
 * Gauss-Seidel: each rank sweeps its own rows forward then backward (orc_l1gs_sweep), the other ranks' values frozen at what they were
 * when the sweep started -- Gauss-Seidel inside a rank, Jacobi across ranks --, and the l1 norms count the off-rank entries of a row
 * (hypre_ParCSRComputeL1Norms option 2): WHAT the reference computes depends on how many ranks it runs on.  The rows are cut like the
 * device's virtual ranks (sweep.hip: n / W rows each, the first n mod W blocks one more), so that a W-process run here and
 * elag_hip_set_gs_hybrid_blocks(W) there are the same arithmetic.  l1W: the l1 norms of that partition.  uold: the frozen copy. */
#define SPMD_GS_LO(n, r) ((r) * ((n) / W) + ((r) < (n) % W ? (r) : (n) % W))
static void spmd_l1gs(const orc_level *L, const double *l1W, const double *f, double *u, double *uold, int rank, int W, spmd_ctl *ctl, int *sense)
{
    const int n = L->A.n, lo = SPMD_GS_LO(n, rank), hi = SPMD_GS_LO(n, rank + 1);
    const int *I = L->A.I, *J = L->A.J;
    const double *a = L->A.a;
    for (int i = lo; i < hi; i++) uold[i] = u[i];
    SPMD_BARRIER();
    for (int pass = 0; pass < 2; pass++)
        for (int i = pass ? hi - 1 : lo; pass ? i >= lo : i < hi; i += pass ? -1 : 1) {
            if (l1W[i] != 0.0) {
                double res = f[i];
                for (int jj = I[i]; jj < I[i + 1]; jj++) { const int j = J[jj]; if (j >= lo && j < hi) res -= a[jj] * u[j]; }       /* the rank's own block first ... */
                for (int jj = I[i]; jj < I[i + 1]; jj++) { const int j = J[jj]; if (j < lo || j >= hi) res -= a[jj] * uold[j]; }   /* ... then its ghosts (diag, offd) */
                u[i] += res / l1W[i];
            }
        }
    SPMD_BARRIER();
}

static void spmd_iterate(orc_hier *H, spmd_vecs *V, const double *RHS, double *SOL, int l, int rank, int W, spmd_ctl *ctl, int *sense)
{
    orc_level *F = &H->lev[l];
    if (l == H->nlevels - 1) {
        if (rank == 0) orc_chol_solve(H->ncoarse, H->chol, RHS, SOL);
        SPMD_BARRIER();
        return;
    }
    const int gs = F->sm.type == 2;
    const int n = F->A.n, nc = F->ncoarse, lo = gs ? SPMD_GS_LO(n, rank) : SPMD_LO(n), hi = gs ? SPMD_GS_LO(n, rank + 1) : SPMD_HI(n), clo = SPMD_LO(nc), chi = SPMD_HI(nc);
    if (gs) spmd_l1gs(F, H->spmd_l1[l], RHS, SOL, V[l].v, rank, W, ctl, sense);
    else spmd_jacobi(F, RHS, SOL, V[l].v, rank, W, ctl, sense);
    for (int i = lo; i < hi; i++) {
        double t = 0.0;
        for (int jj = F->A.I[i]; jj < F->A.I[i + 1]; jj++) t += F->A.a[jj] * SOL[F->A.J[jj]];
        V[l].tc[i] = t;
        V[l].tr[i] = RHS[i] + (-1.0) * t;
    }
    SPMD_BARRIER();
    double *cr = V[l + 1].resid, *cs = V[l + 1].sol;
    for (int i = clo; i < chi; i++) {
        double t = 0.0;
        for (int jj = F->PTI[i]; jj < F->PTI[i + 1]; jj++) t += F->PTa[jj] * V[l].tr[F->PTJ[jj]];
        cr[i] = t;
        cs[i] = 0.0;
    }
    SPMD_BARRIER();
    spmd_iterate(H, V, cr, cs, l + 1, rank, W, ctl, sense);
    for (int i = lo; i < hi; i++) {
        double t = 0.0;
        for (int jj = F->P.I[i]; jj < F->P.I[i + 1]; jj++) t += F->P.a[jj] * cs[F->P.J[jj]];
        SOL[i] += t;
    }
    SPMD_BARRIER();
    if (gs) spmd_l1gs(F, H->spmd_l1[l], RHS, SOL, V[l].v, rank, W, ctl, sense);
    else spmd_jacobi(F, RHS, SOL, V[l].v, rank, W, ctl, sense);
}

/* the l1 norms of level l for the partition the next orc_hier_spmd_bench runs on (type-2 levels; the caller keeps the array alive) */
ORC_API void orc_hier_set_spmd_l1(orc_hier *H, int l, const double *l1W)
{
    if (l >= 0 && l < 64) H->spmd_l1[l] = l1W;
}

/* Runs V-cycles (preconditioner mode, rhs fixed) on W forked worker processes for about `seconds`; returns the number of
 * cycles and the elapsed time of the team, and the last solution (for the parity check against the serial cycle).
 * Every level must be an l1-Jacobi level (type 1, one sweep) above a dense Cholesky coarse solve.  Returns 0 on success. */
ORC_API int orc_hier_spmd_bench(orc_hier *H, int W, double seconds, const double *rhs, double *sol_out, long *cycles_out, double *elapsed_out)
{
    const int nl = H->nlevels;
    if (W < 1 || !H->chol) return 1;
    for (int l = 0; l < nl - 1; l++) {
        const int t = H->lev[l].sm.type;
        if ((t != 1 && t != 2) || H->lev[l].sm.sweeps != 1 || H->lev[l].hiptmair || nl > 64) return 2;
        if (t == 2 && !H->spmd_l1[l]) return 2;          /* (hybrid Gauss-Seidel needs the l1 norms of THIS partition: orc_hier_set_spmd_l1) */
    }
    /* explicit transposes of the interpolators (read-only after this point: shared with the children copy-on-write) */
    for (int l = 0; l < nl - 1; l++) {
        orc_level *L = &H->lev[l];
        if (!L->PTI) {
            const int nnz = L->P.I[L->A.n];
            L->PTI = (int *)malloc(sizeof(int) * ((size_t)L->ncoarse + 1)); L->PTJ = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
            L->PTa = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
            orc_csr_transpose(L->A.n, L->ncoarse, L->P.I, L->P.J, L->P.a, L->PTI, L->PTJ, L->PTa);
        }
    }
    spmd_ctl *ctl = (spmd_ctl *)shared_alloc(sizeof(spmd_ctl));
    spmd_vecs *V = (spmd_vecs *)calloc((size_t)nl, sizeof(spmd_vecs));
    if (!ctl || !V) return 3;
    memset((void *)ctl, 0, sizeof(*ctl));
    const int n0 = H->lev[0].A.n;
    double *rhs_s = (double *)shared_alloc(sizeof(double) * (size_t)n0);
    for (int l = 0; l < nl; l++) {
        const size_t n = (size_t)H->lev[l].A.n;
        V[l].sol = (double *)shared_alloc(sizeof(double) * n); V[l].resid = (double *)shared_alloc(sizeof(double) * n);
        V[l].v = (double *)shared_alloc(sizeof(double) * n); V[l].tc = (double *)shared_alloc(sizeof(double) * n);
        V[l].tr = (double *)shared_alloc(sizeof(double) * n);
        if (!V[l].sol || !V[l].resid || !V[l].v || !V[l].tc || !V[l].tr) return 3;
    }
    memcpy(rhs_s, rhs, sizeof(double) * (size_t)n0);
    const long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    pid_t *kids = (pid_t *)calloc((size_t)W, sizeof(pid_t));
    int rank = 0;
    for (int r = 1; r < W; r++) {
        pid_t p = fork();
        if (p < 0) { W = r; break; }          /* (could not start all of them: run with what we have -- every worker reads W after this) */
        if (p == 0) { rank = r; break; }
        kids[r] = p;
    }
    /* NOTE: W as seen by the children forked before a failure would be stale; a failed fork is reported instead of used */
    /* one process per core: the scheduler spreads W <= cores runnable processes over the cores by itself; explicit pinning
     * (ORC_SPMD_PIN=1: the rank-th allowed core) made things worse in containers whose visible cores share physical ones */
    if (getenv("ORC_SPMD_PIN")) {
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            /* pin to the rank-th allowed core (one process per core) */
            int seen = 0;
            for (int c = 0; c < (int)ncpu && c < CPU_SETSIZE; c++)
                if (CPU_ISSET(c, &set)) {
                    if (seen == rank) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(c, &one); sched_setaffinity(0, sizeof(one), &one); break; }
                    seen++;
                }
        }
    }
    int sense_v = 0, *sense = &sense_v;
    long done = 0;
    double t0 = 0.0;
    /* two untimed cycles, then cycles until rank 0 says stop (checked by everyone at the same barrier) */
    for (int it = 0;; it++) {
        if (it == 2) { SPMD_BARRIER(); if (rank == 0) t0 = now_s(); }
        {
            const int lo = SPMD_LO(n0), hi = SPMD_HI(n0);
            for (int i = lo; i < hi; i++) V[0].sol[i] = 0.0;
            SPMD_BARRIER();
        }
        spmd_iterate(H, V, rhs_s, V[0].sol, 0, rank, W, ctl, sense);
        if (it >= 2) done++;
        if (rank == 0 && it >= 2 && (done % 8) == 0 && now_s() - t0 >= seconds) ctl->stop = 1;
        SPMD_BARRIER();
        if (ctl->stop) break;
    }
    if (rank != 0) _exit(0);
    ctl->elapsed = now_s() - t0;
    ctl->cycles = done;
    for (int r = 1; r < W; r++) if (kids[r] > 0) { int st; waitpid(kids[r], &st, 0); }
    memcpy(sol_out, V[0].sol, sizeof(double) * (size_t)n0);
    *cycles_out = ctl->cycles; *elapsed_out = ctl->elapsed;
    for (int l = 0; l < nl; l++) {
        const size_t n = (size_t)H->lev[l].A.n;
        munmap(V[l].sol, sizeof(double) * n); munmap(V[l].resid, sizeof(double) * n); munmap(V[l].v, sizeof(double) * n);
        munmap(V[l].tc, sizeof(double) * n); munmap(V[l].tr, sizeof(double) * n);
    }
    munmap(rhs_s, sizeof(double) * (size_t)n0); munmap((void *)ctl, sizeof(*ctl));
    free(V); free(kids);
    return 0;
}
