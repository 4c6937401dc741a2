// Smoother implementations + ABI.  Algorithms: SURVEY App. A.2/A.3 (hypre_ParCSRRelax types 0/1/16,
// hypre_ParCSRComputeL1Norms, hypre_ParCSRMaxEigEstimateCG), HiptmairSmoother.cpp:48-109.
#include "smoother.hpp"

#include <algorithm>
#include <cmath>

#include "comm.hpp"
#include "staging.hpp"

namespace elag {

RelaxSmoother::~RelaxSmoother() { dev_free(dinv); dev_free(ds); dev_free(r); dev_free(za); dev_free(zb); dev_free(ds2); dev_free(cds2); dev_free(cds); }

static int read_int(int *d)
{
    return fetch_int(d);
}

static double read_scalar(const double *d)
{
    double v = 0.0;
    fetch_doubles(d, 1, &v);
    return v;
}

void RelaxSmoother::setup()
{
    const int nn = n();
    Temp<int> bad(1);
    bool check_l1 = false;
    zero_async(bad, sizeof(int));
    if (type == ELAG_HIP_SMOOTHER_L1_JACOBI || type == ELAG_HIP_SMOOTHER_JACOBI || type == ELAG_HIP_SMOOTHER_LUMPED_JACOBI) {
        Temp<double> l1((size_t)nn);
        if (type == ELAG_HIP_SMOOTHER_LUMPED_JACOBI) {
            // mfem::HypreSmoother::SetOperator, type 5: l1_norms = A * 1 (row sums over diag and offd), relaxation of type 1 from there on
            Temp<double> ones((size_t)std::max(A->ncols(), 1));
            k::vec_fill(A->ncols(), 1.0, ones);
            k::SpmvArgs a;
            a.x = ones; a.alpha = 1.0; a.out = l1;
            par_spmv(*A, a);
        } else if (type == ELAG_HIP_SMOOTHER_L1_JACOBI) {
            // mfem::HypreSmoother::SetOperator: l1_norms = hypre_ParCSRComputeL1Norms(A, 1)
            k::l1_norms(A->diag, &A->offd, 1, l1, bad);
            check_l1 = true;                                   // (read with the drain at the end: one round trip per smoother, not two)
        } else {
            k::extract_diagonal(A->diag, l1);
        }
        dinv = dev_alloc<double>((size_t)nn);
        k::vec_reciprocal_scale(nn, weight, l1, dinv);
        if (sweeps > 1 || true) { /* nothing else: the alt vector is supplied by the caller */ }
    } else if (type == ELAG_HIP_SMOOTHER_CHEBYSHEV) {
        Temp<double> d((size_t)nn);
        k::extract_diagonal(A->diag, d);
        ds = dev_alloc<double>((size_t)nn);
        k::vec_rsqrt(nn, d, ds, bad);
        if (read_int(bad)) fail(ELAG_HIP_ERR_NUMERIC, "Chebyshev: matrix has a non-positive diagonal entry");
        r = dev_alloc<double>((size_t)nn);
        za = dev_alloc<double>((size_t)nn);
        zb = dev_alloc<double>((size_t)nn);
        estimate_eigs();
        compute_coefs();
    } else if (type == ELAG_HIP_SMOOTHER_TAUBIN || type == ELAG_HIP_SMOOTHER_FIR) {
        // mfem::HypreSmoother::SetOperator, types 1001 / 1002: poly_scale = 0, hypre_ParCSRMaxEigEstimateCG(A, 0, 10); "the Taubin and
        // FIR polynomials are defined on [0, 2]": max_eig /= 2; FIR: window x Chebyshev coefficients
        ELAG_REQUIRE(order >= 1, "FIR / Taubin: \"Cheby Poly Order\" must be >= 1");
        r = dev_alloc<double>((size_t)nn);
        za = dev_alloc<double>((size_t)nn);
        zb = dev_alloc<double>((size_t)nn);
        estimate_eigs();
        max_eig /= 2.0;
        if (!(max_eig > 0.0)) fail(ELAG_HIP_ERR_NUMERIC, "Taubin / FIR: the eigenvalue estimate is not positive (%g)", max_eig);
        compute_fir();
    } else {
        fail(ELAG_HIP_ERR_ARG,
             "relax type %d is not available on the device: the table of ParELAG_HypreSmootherFactory.cpp:92-109 is served for "
             "\"Jacobi\" (0), \"L1 Jacobi\" (1), \"L1 Gauss-Seidel\" (2), \"Kaczmarz\" (3), \"L1 Gauss-Seidel Truncated\" (4), \"Lumped Jacobi\" (5), "
             "\"Gauss-Seidel\" (6), \"Chebyshev\" (16), \"Hypre Jacobi\" (413), \"Taubin\" (1001) and \"FIR\" (1002)", type);
    }
    if (check_l1) {
        if (read_int(bad)) fail(ELAG_HIP_ERR_NUMERIC, "l1-Jacobi: matrix has a zero l1 row norm");
    } else {
        ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    }
}

void RelaxSmoother::compute_fir()
{
    fir.assign((size_t)order + 1, 0.0);
    const double theta = std::acos(1.0 - 0.5 * fraction * max_eig);          // pass band k_pb = poly_fraction * max_eig
    for (int i = 0; i <= order; i++) {
        const double t = (i * M_PI) / (order + 1);
        const double win = window[0] + window[1] * std::cos(t) + window[2] * std::cos(2 * t);
        const double ch = i == 0 ? theta / M_PI : 2.0 * std::sin(i * theta) / (i * M_PI);
        fir[(size_t)i] = win * ch;
    }
}

void RelaxSmoother::compute_coefs()
{
    int ord = std::min(std::max(order, 1), 4);
    cheby_order = ord - 1;
    const double ub = max_eig * 1.1;
    const double lb = (ub - min_eig) * fraction + min_eig;
    const double th = (ub + lb) / 2, de = (ub - lb) / 2;
    double den;
    switch (cheby_order) {
    case 0: coefs[0] = 1.0 / th; break;
    case 1: den = de * de - 2 * th * th; coefs[0] = -4 * th / den; coefs[1] = 2 / den; break;
    case 2: den = 3 * (de * de) * th - 4 * (th * th * th);
            coefs[0] = (3 * de * de - 12 * th * th) / den; coefs[1] = 12 * th / den; coefs[2] = -4 / den; break;
    default: den = std::pow(de, 4) - 8 * de * de * th * th + 8 * std::pow(th, 4);
             coefs[0] = (32 * std::pow(th, 3) - 16 * de * de * th) / den;
             coefs[1] = (8 * de * de - 48 * th * th) / den; coefs[2] = 32 * th / den; coefs[3] = -8 / den; break;
    }
    if (type == ELAG_HIP_SMOOTHER_CHEBYSHEV && cheby_order == 1 && ds) {
        const int nn = n();
        if (!ds2) { ds2 = dev_alloc<double>((size_t)nn); cds2 = dev_alloc<double>((size_t)nn); cds = dev_alloc<double>((size_t)nn); }
        k::vec_scale_mul(nn, 1.0, ds, ds, ds2);
        k::vec_scale_mul(nn, coefs[1], ds, ds, cds2);
        k::vec_axpby(nn, coefs[1], ds, 0.0, nullptr, cds);
    }
}

// symmetric tridiagonal eigenvalues (implicit QL with Wilkinson shift); d[0..n-1], e[1..n-1]
static void tridiag_eigs(int n, std::vector<double> &d, std::vector<double> &e)
{
    for (int i = 1; i < n; i++) e[(size_t)i - 1] = e[(size_t)i];
    e[(size_t)n - 1] = 0.0;
    for (int l = 0; l < n; l++) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; m++) {
                const double dd = std::fabs(d[(size_t)m]) + std::fabs(d[(size_t)m + 1]);
                if (std::fabs(e[(size_t)m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (iter++ == 60) fail(ELAG_HIP_ERR_NUMERIC, "tridiagonal eigen-solve did not converge");
                double g = (d[(size_t)l + 1] - d[(size_t)l]) / (2.0 * e[(size_t)l]);
                double rr = std::hypot(g, 1.0);
                g = d[(size_t)m] - d[(size_t)l] + e[(size_t)l] / (g + std::copysign(rr, g));
                double s = 1.0, c = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; i--) {
                    double f = s * e[(size_t)i], b = c * e[(size_t)i];
                    rr = std::hypot(f, g);
                    e[(size_t)i + 1] = rr;
                    if (rr == 0.0) { d[(size_t)i + 1] -= p; e[(size_t)m] = 0.0; break; }
                    s = f / rr; c = g / rr;
                    g = d[(size_t)i + 1] - p;
                    rr = (d[(size_t)i] - g) * s + 2.0 * c * b;
                    p = s * rr;
                    d[(size_t)i + 1] = g + p;
                    g = c * rr - b;
                }
                if (rr == 0.0 && i >= l) continue;
                d[(size_t)l] -= p; e[(size_t)l] = g; e[(size_t)m] = 0.0;
            }
        } while (m != l);
    }
    std::sort(d.begin(), d.begin() + n);
}

void RelaxSmoother::estimate_eigs()
{
    // hypre_ParCSRMaxEigEstimateCG(A, scale = 1, max_iter = 10): CG-Lanczos on D^-1/2 A D^-1/2 from
    // hypre_ParVectorSetRandomValues(r, 1) (local seed = 1 * (rank + 1)).
    Context &c = ctx();
    const int nn = n();
    int max_iter = 10;
    if (A->global_rows < max_iter) max_iter = (int)A->global_rows;
    if (max_iter == 0) { max_eig = min_eig = 0.0; return; }
    // a matrix whose local rows are all its rows is serial (or replicated on every rank): serial semantics, so that
    // replicated smoothers are identical on all ranks
    const bool distributed = A->needs_exchange() || A->global_rows != A->nrows();
    Temp<double> rv((size_t)nn), p((size_t)nn), s((size_t)nn);
    k::vec_hypre_random(nn, distributed ? 1 * (c.rank + 1) : 1, rv);
    std::vector<double> tri((size_t)max_iter + 1, 0.0), off((size_t)max_iter + 1, 0.0);
    double gamma = 1.0, beta = 1.0;
    int i = 0;
    while (i < max_iter) {
        const double gamma_old = gamma;
        k::vec_dot(nn, rv, rv, c.d_scalar);
        gamma = read_scalar(c.d_scalar);
        if (distributed) comm_allreduce_scalar(&gamma, 0);
        if (i == 0) {
            beta = 1.0;
            dev_copy(p, rv, sizeof(double) * (size_t)nn, c.stream);
        } else {
            beta = gamma / gamma_old;
            k::vec_axpby(nn, 1.0, rv, beta, p, p);
        }
        k::SpmvArgs a;  // s = ds .* A (ds .* p)        (ds == nullptr: scale = 0, s = A p)
        a.x = p; a.cs = ds; a.alpha = 1.0; a.s = ds; a.out = s;
        par_spmv(*A, a);
        k::vec_dot(nn, s, p, c.d_scalar);
        double sdotp = read_scalar(c.d_scalar);
        if (distributed) comm_allreduce_scalar(&sdotp, 0);
        const double alpha = gamma / sdotp;
        const double alphainv = 1.0 / alpha;
        tri[(size_t)i + 1] = alphainv;
        tri[(size_t)i] *= beta;
        tri[(size_t)i] += alphainv;
        off[(size_t)i + 1] = alphainv;
        off[(size_t)i] *= std::sqrt(beta);
        k::vec_axpby(nn, -alpha, s, 1.0, rv, rv);
        i++;
    }
    tridiag_eigs(i, tri, off);
    max_eig = tri[(size_t)i - 1];
    min_eig = tri[0];
}

int RelaxSmoother::flips(bool zero_guess) const
{
    if (type == ELAG_HIP_SMOOTHER_CHEBYSHEV || type == ELAG_HIP_SMOOTHER_TAUBIN || type == ELAG_HIP_SMOOTHER_FIR) return 0;
    return zero_guess ? sweeps - 1 : sweeps;
}

void RelaxSmoother::apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool /*transpose*/)
{
    const int nn = n();
    if (type == ELAG_HIP_SMOOTHER_TAUBIN) {
        // mfem ParCSRRelax_Taubin: taubin_iter steps  r = f - A u;  u += c r / max_eig,  c = lambda, mu, lambda, ...
        if (zero_guess) k::vec_fill(nn, 0.0, cur);
        double *u = cur, *w = r;
        for (int sw = 0; sw < sweeps; sw++)
            for (int i = 0; i < taubin_iter; i++) {
                const double c = ((i % 2) == 0 ? taubin_lambda : taubin_mu) / max_eig;
                k::SpmvArgs a;
                a.x = u; a.alpha = -c; a.f = f; a.gamma = c; a.yin = u; a.beta = 1.0; a.out = w;
                par_spmv(*A, a);
                std::swap(u, w);
            }
        if (u != cur) dev_copy(cur, u, sizeof(double) * (size_t)nn, ctx().stream);
        return;
    }
    if (type == ELAG_HIP_SMOOTHER_FIR) {
        // mfem ParCSRRelax_FIR: x1 = x0 + (f - A x0) / m;  x3 = c0 x0 + c1 x1;  x2 = 2 x1 - x0 + 2 (f - A x1) / m;  x3 += c_k x2;  u = x3
        if (zero_guess) k::vec_fill(nn, 0.0, cur);
        const double m = max_eig;
        for (int sw = 0; sw < sweeps; sw++) {
            double *x0 = cur, *x1 = za, *x2 = zb, *x3 = r;
            {
                k::SpmvArgs a;
                a.x = x0; a.alpha = -1.0 / m; a.f = f; a.gamma = 1.0 / m; a.yin = x0; a.beta = 1.0; a.out = x1;
                par_spmv(*A, a);
            }
            k::vec_axpby(nn, fir[0], x0, fir[1], x1, x3);
            for (int kk = 2; kk <= order; kk++) {
                k::SpmvArgs a;                                    // x2 = 2 x1 + 2 (f - A x1) / m
                a.x = x1; a.alpha = -2.0 / m; a.f = f; a.gamma = 2.0 / m; a.yin = x1; a.beta = 2.0; a.out = x2;
                par_spmv(*A, a);
                k::vec_axpby(nn, 1.0, x2, -1.0, x0, x2);          // x2 -= x0
                k::vec_axpby(nn, fir[(size_t)kk], x2, 1.0, x3, x3);
                double *old0 = x0;                                // (the old x0 -- `cur` itself the first time -- is free from here on)
                x0 = x1; x1 = x2; x2 = old0;
            }
            dev_copy(cur, x3, sizeof(double) * (size_t)nn, ctx().stream);
        }
        return;
    }
    if (type != ELAG_HIP_SMOOTHER_CHEBYSHEV) {
        for (int sw = 0; sw < sweeps; sw++) {
            if (zero_guess && sw == 0) {
                // u = 0 + w (f - A 0) / l1
                k::vec_scale_mul(nn, 1.0, f, dinv, cur);
            } else {
                // hypre_ParCSRRelax type 1: v = w (f - A u); u += v / l1   -- one fused pass, out of place
                k::SpmvArgs a;
                a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.s = dinv; a.yin = cur; a.beta = 1.0; a.out = alt;
                par_spmv(*A, a);
                std::swap(cur, alt);
            }
        }
        return;
    }
    // hypre_ParCSRRelax_Cheby(scale = 1, variant = 0)
    const int co = cheby_order;
    for (int sw = 0; sw < sweeps; sw++) {
        const bool zero = zero_guess && sw == 0;
        if (co == 1 && ds2) {
            // order 2: the sweep's vector passes ride on its products.  From the zero guess
            //   u = ds (c0 ds f + ds A (ds c1 ds f)) = ds^2 (c0 f + A (c1 ds^2 f))                                  -- one product;
            // otherwise r = ds (f - A u), then u += ds (c0 r + ds A (c1 ds r))                                       -- two.
            if (zero) {
                k::SpmvArgs a;
                a.x = f; a.cs = cds2; a.alpha = 1.0; a.f = f; a.gamma = coefs[0]; a.s = ds2; a.out = cur;
                par_spmv(*A, a);
            } else {
                k::SpmvArgs a;  // r = ds .* (f - A u)
                a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.s = ds; a.out = r;
                par_spmv(*A, a);
                k::SpmvArgs b;
                b.x = r; b.cs = cds; b.alpha = 1.0; b.s2 = ds; b.f = r; b.gamma = coefs[0]; b.s = ds; b.yin = cur; b.beta = 1.0; b.out = cur;
                par_spmv(*A, b);
            }
            continue;
        }
        if (zero) {
            k::vec_scale_mul(nn, 1.0, f, ds, r);  // r = ds .* f
        } else {
            k::SpmvArgs a;  // r = ds .* (f - A u)
            a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.s = ds; a.out = r;
            par_spmv(*A, a);
        }
        if (co == 0) {
            k::vec_add_scaled_mul(nn, zero ? nullptr : cur, ds, nullptr, coefs[0], r, cur);
            continue;
        }
        // z = c_{co-1} r + ds .* A (ds .* (c_co r))
        double *z = za, *zo = zb;
        {
            k::SpmvArgs a;
            a.x = r; a.cs = ds; a.alpha = coefs[co]; a.s = ds; a.yin = r; a.beta = coefs[co - 1]; a.out = z;
            par_spmv(*A, a);
        }
        for (int i = co - 2; i >= 0; i--) {
            k::SpmvArgs a;  // z' = c_i r + ds .* A (ds .* z)
            a.x = z; a.cs = ds; a.alpha = 1.0; a.s = ds; a.yin = r; a.beta = coefs[i]; a.out = zo;
            par_spmv(*A, a);
            std::swap(z, zo);
        }
        k::vec_add_scaled_mul(nn, zero ? nullptr : cur, ds, z, 0.0, nullptr, cur);  // u = orig_u + ds .* z
    }
}

// ---- Hiptmair -----------------------------------------------------------------------------------
HiptmairSmoother::~HiptmairSmoother() { dev_free(pv); dev_free(auxb); dev_free(auxx); dev_free(auxalt); }

void HiptmairSmoother::setup()
{
    ELAG_REQUIRE(D->nrows() == A->nrows(), "Hiptmair: D has %d rows, A has %d", D->nrows(), A->nrows());
    ELAG_REQUIRE(aux->n() == D->ncols(), "Hiptmair: auxiliary operator size %d != columns of D %d", aux->n(), D->ncols());
    pv = dev_alloc<double>((size_t)A->nrows());
    auxb = dev_alloc<double>((size_t)D->ncols());
    auxx = dev_alloc<double>((size_t)D->ncols());
    auxalt = dev_alloc<double>((size_t)D->ncols());
    D->ensure_transpose();
    // the derivative of a lowest-order sequence is an incidence matrix (entries +-1, eliminated essential columns 0): both it
    // and its stored transpose go to the 4-byte-per-entry sign form when that is what they hold (the smoother keeps D fixed)
    if (!D->needs_exchange() && std::getenv("ELAG_HIP_NO_SIGN_FORM") == nullptr) {
        k::prepare_sign_form(D->diag);
        if (D->diagT) k::prepare_sign_form(*D->diagT);
    }
    k::prepare_spmv(A->diag); k::prepare_spmv(D->diag);
    if (D->diagT) k::prepare_spmv(*D->diagT);
    if (aux->A) k::prepare_spmv(aux->A->diag);
    // A W for the sweep from the zero guess on a bandwidth-bound level (smoother.hpp)
    AW.reset();
    const double *w = (primary && primary->A == A) ? primary->diag_scaling() : nullptr;
    const char *e_aw = std::getenv("ELAG_HIP_HIPTMAIR_AW");                  // (0: the product that gathers f and W per entry; A/B runs)
    if (w && !(e_aw && !atoi(e_aw)) && !A->needs_exchange() && A->offd.ncols == 0 && A->diag.nnz >= (int64_t(1) << 20)) {
        AW.reset(new DevCSR());
        k::csr_copy(A->diag, *AW);
        k::csr_scale_cols(*AW, w);
        k::prepare_spmv(*AW);
        if (AW->xc_state == 1 || AW->lanes_per_row != 0) AW.reset();           // (the option is the plain stream kernel's; with an x-cache form the fused gather is cheap anyway)
    }
}

void HiptmairSmoother::aux_correct(const double *resid, double *x)
{
    // cB = D^T defect ; relax on A_aux (iterative_mode == false) ; X += D * correction   (HiptmairSmoother.cpp:61-75)
    double *cx = auxx, *ca = auxalt;
    const double *aux_w = (aux->A && aux->n() == D->ncols()) ? aux->diag_scaling() : nullptr;
    if (aux_w && !D->needs_exchange() && D->diagT && k::sign_twohop(D->diag, *D->diagT, aux_w, resid, x, x)) return;      // D^T, the scaling and D in one launch
    if (aux_w && !D->needs_exchange()) {
        // one diagonal sweep from the zero guess is a row scaling: folded into the epilogue of the D^T product
        par_spmvT(*D, 1.0, resid, 0.0, cx, aux_w);
    } else {
        par_spmvT(*D, 1.0, resid, 0.0, auxb);
        if (aux->flips(true) & 1) std::swap(cx, ca);
        aux->apply(auxb, cx, ca, true, false);
    }
    k::SpmvArgs a;
    a.x = cx; a.alpha = 1.0; a.yin = x; a.beta = 1.0; a.out = x;
    par_spmv(*D, a);
}

void HiptmairSmoother::finish_after_primary(const double *f, double *x)
{
    k::SpmvArgs a;  // R = B - A X
    a.x = x; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.out = pv;
    par_spmv(*A, a);
    aux_correct(pv, x);
}

void HiptmairSmoother::apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose)
{
    if (!transpose) {
        const double *w = (primary->A == A) ? primary->diag_scaling() : nullptr;
        if (zero_guess && w && AW) {
            k::SpmvArgs a;   // R = B - (A W) B, X = W B written beside it
            a.x = f; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.out = pv; a.out2 = cur; a.o2s = w;
            k::spmv_fused(*AW, nullptr, a);
        } else if (zero_guess && w) {
            // X = W B and R = B - A X in one product: the sweep's result is the (column-scaled) operand of the SpMV
            k::SpmvArgs a;
            a.x = f; a.cs = w; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.out = pv; a.out2 = cur;
            par_spmv(*A, a);
        } else {
            primary->apply(f, cur, alt, zero_guess, false);
            k::SpmvArgs a;  // R = B - A X
            a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.out = pv;
            par_spmv(*A, a);
        }
        aux_correct(pv, cur);
    } else {
        // MultTranspose (:79-109): auxiliary correction first, then the primary sweep in iterative mode
        if (zero_guess) {
            k::vec_fill(n(), 0.0, cur);
            aux_correct(f, cur);
        } else {
            k::SpmvArgs a;
            a.x = cur; a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.out = pv;
            par_spmv(*A, a);
            aux_correct(pv, cur);
        }
        primary->apply(f, cur, alt, false, true);
    }
}

}  // namespace elag

// ---- dense coarse solver --------------------------------------------------------------------------
void elag_hip_coarse_s::setup()
{
    using namespace elag;
    ELAG_REQUIRE(kind == 0 || kind == 1, "coarse solver kind %d is not implemented (0 = dense Cholesky, 1 = dense LU)", kind);
    ELAG_REQUIRE(!A->needs_exchange() && (A->global_rows == A->nrows() || replicated),
                 "dense coarse solve expects a rank-local operator (use elag_hip_coarse_create_replicated across ranks)");
    n = A->nrows();
    ELAG_REQUIRE(n == A->ncols(), "coarse operator must be square");
    ELAG_REQUIRE(n <= 4096, "dense coarse solve limited to 4096 dofs (got %d)", n);
    if (kind == 1) {
        // general (indefinite) operator: Gauss-Jordan with partial pivoting on the device (dense.hip)
        Minv = dev_alloc<double>((size_t)n * n);
        job.stream = async_stream;
        k::dense_inverse_general(A->diag, Minv, async_stream ? &job : nullptr);
        return;
    }
    // beyond a few hundred dofs the host's scalar Cholesky + n triangular solve pairs take longer than everything else of a setup (358 ms at
    // n = 1 015): the device inverts those (two launches per pivot, ~10 ms); small operators keep the host path
    static int dev_from = -1;
    if (dev_from < 0) { const char *e = getenv("ELAG_HIP_COARSE_DEVICE_FROM"); dev_from = e ? atoi(e) : 256; }
    // ... and up to 141 dofs the matrix fits one workgroup's LDS: one launch (dense.hip; ELAG_HIP_COARSE_LDS=0: the host path)
    static int lds_inv = -1;
    if (lds_inv < 0) { const char *e = getenv("ELAG_HIP_COARSE_LDS"); lds_inv = e ? atoi(e) : 1; }
    if (n >= dev_from || (lds_inv && n >= 8 && n <= 141)) {
        Minv = dev_alloc<double>((size_t)n * n);
        job.stream = async_stream;
        k::dense_inverse_spd(A->diag, Minv, async_stream ? &job : nullptr);
        return;
    }
    std::vector<int> I = host_from_dev(A->diag.I, (size_t)n + 1);
    std::vector<int> J = host_from_dev(A->diag.J, (size_t)A->diag.nnz);
    std::vector<double> a = host_from_dev(A->diag.a, (size_t)A->diag.nnz);
    std::vector<double> L((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++)
        for (int k = I[(size_t)i]; k < I[(size_t)i + 1]; k++) L[(size_t)i * n + J[(size_t)k]] += a[(size_t)k];
    // symmetrise the (numerically symmetric) input, then Cholesky L L^T in place (lower)
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) {
            const double v = 0.5 * (L[(size_t)i * n + j] + L[(size_t)j * n + i]);
            L[(size_t)i * n + j] = v; L[(size_t)j * n + i] = v;
        }
    for (int j = 0; j < n; j++) {
        double d = L[(size_t)j * n + j];
        for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) fail(ELAG_HIP_ERR_NUMERIC, "coarse operator is not positive definite (pivot %d = %g)", j, d);
        d = std::sqrt(d);
        L[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = L[(size_t)i * n + j];
            for (int k = 0; k < j; k++) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / d;
        }
    }
    // inverse: solve L L^T X = I column by column
    std::vector<double> Minv_h((size_t)n * n), x((size_t)n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = (c < i ? c : i); k < i; k++) s -= L[(size_t)i * n + k] * x[(size_t)k];
            x[(size_t)i] = (i < c) ? 0.0 : s / L[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double s = x[(size_t)i];
            for (int k = i + 1; k < n; k++) s -= L[(size_t)k * n + i] * x[(size_t)k];
            x[(size_t)i] = s / L[(size_t)i * n + i];
        }
        for (int i = 0; i < n; i++) Minv_h[(size_t)i * n + c] = x[(size_t)i];
    }
    // exact symmetry of the stored inverse keeps the cycle symmetric to rounding
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++) {
            const double v = 0.5 * (Minv_h[(size_t)i * n + j] + Minv_h[(size_t)j * n + i]);
            Minv_h[(size_t)i * n + j] = v; Minv_h[(size_t)j * n + i] = v;
        }
    Minv = dev_from_host(Minv_h.data(), Minv_h.size());
    ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
}

void elag_hip_coarse_s::apply(const double *rhs, double *sol)
{
    using namespace elag;
    if (custom_apply) { custom_apply(rhs, sol); return; }
    finish();                                                            // (an inverse still running beside a build: nobody reads Minv before it is there)
    if (!replicated) { k::dense_gemv(n, n, Minv, rhs, sol); return; }
    // RCCL traffic always goes through the comm stream (one communicator, one stream): fork/join with events
    Context &c = ctx();
    ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
    comm_allgatherv(rhs, full_rhs, counts.data(), displs.data(), c.comm_stream);
    ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
    ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
    k::dense_gemv(nloc, n, Minv + (size_t)my0 * n, full_rhs, sol);
}

using namespace elag;

extern "C" {

int elag_hip_smoother_create(elag_hip_smoother_t *out, elag_hip_parcsr_t A, int relax_type, int sweeps, double damping,
                             double omega, int cheby_order, double cheby_fraction)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && A, "null argument");
        ELAG_REQUIRE(A->nrows() == A->ncols(), "smoother needs a square operator (%d x %d)", A->nrows(), A->ncols());
        ELAG_REQUIRE(sweeps >= 1, "Sweeps must be >= 1");
        if (relax_type == ELAG_HIP_SMOOTHER_L1_GS || relax_type == ELAG_HIP_SMOOTHER_L1_GS_TRUNCATED || relax_type == ELAG_HIP_SMOOTHER_GS ||
            relax_type == ELAG_HIP_SMOOTHER_KACZMARZ) {
            std::unique_ptr<SweepSmoother> G(new SweepSmoother());
            G->A = A; G->type = relax_type; G->sweeps = sweeps; G->weight = damping; G->omega = omega;
            G->setup();
            *out = G.release();
            return;
        }
        if (relax_type == ELAG_HIP_SMOOTHER_HYPRE_JACOBI) {
            std::unique_ptr<DiagScaleSmoother> G(new DiagScaleSmoother());
            G->A = A;
            *out = G.release();
            return;
        }
        std::unique_ptr<RelaxSmoother> S(new RelaxSmoother());
        S->A = A; S->type = relax_type; S->sweeps = sweeps; S->weight = damping; S->omega = omega;
        S->order = cheby_order; S->fraction = cheby_fraction;
        S->setup();
        *out = S.release();
    });
}

int elag_hip_smoother_get_eig_estimate(elag_hip_smoother_t S, double *max_eig, double *min_eig)
{
    return guarded([&] {
        auto *R = dynamic_cast<RelaxSmoother *>(S);
        ELAG_REQUIRE(R && (R->type == ELAG_HIP_SMOOTHER_CHEBYSHEV || R->type == ELAG_HIP_SMOOTHER_TAUBIN || R->type == ELAG_HIP_SMOOTHER_FIR),
                     "not a Chebyshev, Taubin or FIR smoother");
        if (max_eig) *max_eig = R->max_eig;
        if (min_eig) *min_eig = R->min_eig;
    });
}

int elag_hip_set_gs_hybrid_blocks(int nblocks)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(nblocks >= 0, "elag_hip_set_gs_hybrid_blocks: nblocks must be >= 0 (got %d)", nblocks);
        ctx().gs_blocks = nblocks;
    });
}

int elag_hip_smoother_set_eig_estimate(elag_hip_smoother_t S, double max_eig, double min_eig)
{
    return guarded([&] {
        auto *R = dynamic_cast<RelaxSmoother *>(S);
        ELAG_REQUIRE(R && (R->type == ELAG_HIP_SMOOTHER_CHEBYSHEV || R->type == ELAG_HIP_SMOOTHER_TAUBIN || R->type == ELAG_HIP_SMOOTHER_FIR),
                     "not a Chebyshev, Taubin or FIR smoother");
        R->max_eig = max_eig; R->min_eig = min_eig;
        if (R->type == ELAG_HIP_SMOOTHER_CHEBYSHEV) R->compute_coefs();
        else R->compute_fir();                                 // (max_eig is taken as given: the halved estimate)
    });
}

int elag_hip_hiptmair_create(elag_hip_smoother_t *out, elag_hip_parcsr_t A, elag_hip_parcsr_t D,
                             elag_hip_smoother_t primary, elag_hip_smoother_t auxiliary)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && A && D && primary && auxiliary, "null argument");
        std::unique_ptr<HiptmairSmoother> S(new HiptmairSmoother());
        S->A = A; S->D = D; S->primary = primary; S->aux = auxiliary;
        S->setup();
        *out = S.release();
    });
}

int elag_hip_smoother_apply(elag_hip_smoother_t S, const double *rhs, double *sol, int zero_initial_guess, int transpose, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(S && rhs && sol, "null argument");
        check_loc(loc);
        const size_t n = (size_t)S->n();
        VecIn f(rhs, n, loc);
        VecInOut u(sol, n, loc, !zero_initial_guess);
        Temp<double> altbuf(n);
        double *cur = u.d, *alt = altbuf;
        S->apply(f.d, cur, alt, zero_initial_guess != 0, transpose != 0);
        if (cur != u.d)
            dev_copy(u.d, cur, n * sizeof(double), ctx().stream);
        u.finish();
    });
}

int elag_hip_smoother_destroy(elag_hip_smoother_t S)
{
    return guarded([&] {
        if (S) { (void)hipStreamSynchronize(ctx().stream); delete S; }
    });
}

}  // extern "C"

namespace elag {
elag_hip_coarse_s *coarse_create_beside(ParCSR *A, int kind)
{
    std::unique_ptr<elag_hip_coarse_s> C(new elag_hip_coarse_s());
    C->A = A; C->kind = kind;
    const char *e = getenv("ELAG_HIP_COARSE_ASYNC");                     // (read at every build: a test compares the two)
    Context &c = ctx();
    if (!(e && !atoi(e)) && c.nranks == 1 && c.comm_kind != 2 && c.comm_stream && !comm_active()) C->async_stream = c.comm_stream;
    C->setup();
    return C.release();
}
}  // namespace elag

extern "C" {

int elag_hip_coarse_create(elag_hip_coarse_t *out, elag_hip_parcsr_t A, int kind, int iters)
{
    return guarded([&] {
        require_init();
        (void)iters;
        ELAG_REQUIRE(out && A, "null argument");
        std::unique_ptr<elag_hip_coarse_s> C(new elag_hip_coarse_s());
        C->A = A; C->kind = kind;
        C->setup();
        *out = C.release();
    });
}

int elag_hip_coarse_create_replicated(elag_hip_coarse_t *out, elag_hip_parcsr_t A_global, const int64_t *row_starts, int kind)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && A_global && row_starts, "null argument");
        const int nr = ctx().nranks, me = ctx().rank;
        ELAG_REQUIRE(row_starts[0] == 0 && row_starts[nr] == A_global->nrows(), "row_starts must cover the %d coarse rows", A_global->nrows());
        std::unique_ptr<elag_hip_coarse_s> C(new elag_hip_coarse_s());
        C->A = A_global; C->kind = kind; C->replicated = true;
        C->counts.resize((size_t)nr); C->displs.resize((size_t)nr);
        for (int r = 0; r < nr; r++) {
            ELAG_REQUIRE(row_starts[r + 1] >= row_starts[r], "row_starts must not decrease");
            C->counts[(size_t)r] = (int)(row_starts[r + 1] - row_starts[r]);
            C->displs[(size_t)r] = (int)row_starts[r];
        }
        C->setup();
        C->nloc = C->counts[(size_t)me]; C->my0 = C->displs[(size_t)me];
        C->full_rhs = dev_alloc<double>((size_t)C->n);
        *out = C.release();
    });
}

int elag_hip_coarse_apply(elag_hip_coarse_t C, const double *rhs, double *sol, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(C && rhs && sol, "null argument");
        check_loc(loc);
        const size_t nv = (size_t)(C->replicated ? C->nloc : C->n);
        ELAG_REQUIRE(!C->custom_apply || loc == ELAG_HIP_DEVICE, "a hierarchy-backed coarse solver takes device vectors");
        VecIn f(rhs, nv, loc);
        VecInOut u(sol, nv, loc, false);
        C->apply(f.d, u.d);
        u.finish();
    });
}

int elag_hip_coarse_destroy(elag_hip_coarse_t C)
{
    return guarded([&] {
        if (C) { (void)hipStreamSynchronize(ctx().stream); delete C; }
    });
}

}  // extern "C"
