// Smoothers: hypre_ParCSRRelax types 0/1 (Jacobi, l1-Jacobi) and 16 (Chebyshev), HiptmairSmoother, and the
// dense coarse solver.  Internally smoothers work on a ping-pong pair of device vectors so that the
// Jacobi update never needs an extra copy pass.
#pragma once
#include <functional>

#include "parcsr.hpp"

struct elag_hip_smoother_s {
    elag::ParCSR *A = nullptr;
    virtual ~elag_hip_smoother_s() {}
    // sol <- S(rhs, sol).  On entry the iterate is in `cur`; on exit the result is in `cur` (the two
    // pointers may have been swapped).  zero_guess: ignore the content of cur (iterative_mode == false).
    virtual void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) = 0;
    // number of cur/alt swaps apply() performs (lets the caller choose where the result ends up)
    virtual int flips(bool zero_guess) const = 0;
    int n() const { return A->nrows(); }
};

namespace elag {

struct RelaxSmoother : elag_hip_smoother_s {
    int type = 1, sweeps = 1, order = 2, cheby_order = 1;
    double weight = 1.0, omega = 1.0, fraction = 0.3;
    double max_eig = 0.0, min_eig = 0.0;
    double coefs[5] = {0, 0, 0, 0, 0};
    double *dinv = nullptr;               // weight / l1   (types 0, 1)
    double *ds = nullptr;                 // 1/sqrt(a_ii)  (type 16)
    double *r = nullptr, *za = nullptr, *zb = nullptr;
    ~RelaxSmoother() override;
    void setup();
    void estimate_eigs();                 // hypre_ParCSRMaxEigEstimateCG(A, scale=1, 10)
    void compute_coefs();
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool zero_guess) const override;
};

struct HiptmairSmoother : elag_hip_smoother_s {
    ParCSR *D = nullptr;
    elag_hip_smoother_s *primary = nullptr, *aux = nullptr;
    double *pv = nullptr, *auxb = nullptr, *auxx = nullptr, *auxalt = nullptr;
    ~HiptmairSmoother() override;
    void setup();
    void aux_correct(const double *resid, double *x);
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool zero_guess) const override { return primary->flips(zero_guess); }
};

}  // namespace elag

struct elag_hip_coarse_s {
    elag::ParCSR *A = nullptr;
    int kind = 0, n = 0;
    double *Minv = nullptr;  // dense inverse, row-major n x n (device)
    // replicated mode: n = global size, this rank applies rows [my0, my0 + nloc)
    bool replicated = false;
    int nloc = 0, my0 = 0;
    std::vector<int> counts, displs;
    double *full_rhs = nullptr;
    // sub-hierarchy mode (elag_hip_coarse_create_from_hierarchy): apply() = [all-reduce rhs] + replicated cycle
    std::function<void(const double *, double *)> custom_apply;
    ~elag_hip_coarse_s() { elag::dev_free(Minv); elag::dev_free(full_rhs); }
    void setup();
    void apply(const double *rhs, double *sol);
};
