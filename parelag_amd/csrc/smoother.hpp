// Smoothers: hypre_ParCSRRelax types 0/1 (Jacobi, l1-Jacobi) and 16 (Chebyshev), HiptmairSmoother, and the
// dense coarse solver.  Internally smoothers work on a ping-pong pair of device vectors so that the
// Jacobi update never needs an extra copy pass.
#pragma once
#include <memory>
#include <functional>
#include <vector>

#include "parcsr.hpp"

struct elag_hip_smoother_s {
    elag::ParCSR *A = nullptr;
    virtual ~elag_hip_smoother_s() {}
    // sol <- S(rhs, sol).  On entry the iterate is in `cur`; on exit the result is in `cur` (the two
    // pointers may have been swapped).  zero_guess: ignore the content of cur (iterative_mode == false).
    virtual void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) = 0;
    // number of cur/alt swaps apply() performs (lets the caller choose where the result ends up)
    virtual int flips(bool zero_guess) const = 0;
    // non-null iff apply() is exactly  sol <- sol + W (rhs - A sol)  with W = diag(returned vector), one sweep: lets the
    // hierarchy fold the sweep into its fused level kernels
    virtual const double *diag_scaling() const { return nullptr; }
    // identifies smoothers that are the same operator (same W): type/weight for the relaxations
    virtual bool same_scaling_as(const elag_hip_smoother_s *o) const { return o == this; }
    // roughly how many dependent device steps (~2 us each) one apply() is: a launch for most smoothers, a dependency level for the
    // sweeps -- what the hierarchy prices a sub-cycle with before it builds a dense tail out of n applications of it
    virtual double chain_cost() const { return 2.0; }
    // true iff apply() does nothing but enqueue kernels on the compute stream -- no read-back of a count or a flag, no allocation, no halo
    // exchange or collective: the only solvers a Krylov step that replays as a graph may contain.  A whitelist: a solver that does not
    // say so is issued as plain launches (krylov.hip: capturable()).
    virtual bool graph_capturable() const { return false; }
    // A solver applied from the zero guess is linear in its right-hand side, and S(-f) = -S(f) to the last bit (round to nearest is symmetric): a
    // caller that can hand over -f, or take the result with a sign, for nothing -- the sign of a product's epilogue -- asks for the solver
    // WITHOUT its own factor and, where offered, for the result added in place, instead of paying a vector pass for either (blocked.hip: LDU).
    //   result_scale(): the factor apply() multiplies its result with (1: none)
    //   plain_forms(): bit 0 = apply_plain(f, out, false) is offered, bit 1 = apply_plain(f, out, true)
    //   apply_plain(f, out, add): out (+)= S(f) from the zero guess, the factor left out
    virtual double result_scale() const { return 1.0; }
    virtual int plain_forms() const { return 0; }
    virtual void apply_plain(const double *, double *, bool) { elag::fail(ELAG_HIP_ERR_STATE, "internal: apply_plain() of a solver that does not offer it"); }
    int n() const { return A->nrows(); }
};

namespace elag {

// The coarsest operator's dense solver with its inverse computed beside the rest of the build (elag_hip_coarse_s::async_stream: the second
// stream, idle on one rank; ELAG_HIP_COARSE_ASYNC=0, several ranks: as elag_hip_coarse_create).  The caller calls finish() before Minv is read.
elag_hip_coarse_s *coarse_create_beside(ParCSR *A, int kind);

struct RelaxSmoother : elag_hip_smoother_s {
    int type = 1, sweeps = 1, order = 2, cheby_order = 1;
    double weight = 1.0, omega = 1.0, fraction = 0.3;
    double max_eig = 0.0, min_eig = 0.0;
    double coefs[5] = {0, 0, 0, 0, 0};
    double *dinv = nullptr;               // weight / l1   (types 0, 1)
    double *ds = nullptr;                 // 1/sqrt(a_ii)  (type 16)
    double *r = nullptr, *za = nullptr, *zb = nullptr;
    double *ds2 = nullptr, *cds2 = nullptr, *cds = nullptr;   // order 2: ds.^2, c1 ds.^2, c1 ds (the sweep's vector passes folded into its products)
    // types 1001 / 1002 (mfem::HypreSmoother's Taubin and FIR polynomials; max_eig holds the halved estimate)
    double taubin_lambda = 0.5, taubin_mu = -0.5;
    int taubin_iter = 40;
    double window[3] = {0.42, 0.50, 0.08};
    std::vector<double> fir;
    void compute_fir();                   // HypreSmoother::SetFIRCoefficients(max_eig)
    ~RelaxSmoother() override;
    void setup();
    void estimate_eigs();                 // hypre_ParCSRMaxEigEstimateCG(A, scale = 1 (Chebyshev) or 0 (Taubin, FIR), 10)
    void compute_coefs();
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool zero_guess) const override;
    const double *diag_scaling() const override { return (dinv && sweeps == 1) ? dinv : nullptr; }
    bool graph_capturable() const override { return !is_distributed(*A); }
    bool same_scaling_as(const elag_hip_smoother_s *o) const override
    {
        const RelaxSmoother *r = dynamic_cast<const RelaxSmoother *>(o);
        return r && r->A == A && r->type == type && r->weight == weight && r->sweeps == sweeps;
    }
};

// hypre relax types 2 / 4 (l1 Gauss-Seidel, plain / truncated l1 norms), 6 (hybrid symmetric Gauss-Seidel) and 3 (Kaczmarz):
// level-scheduled forward + backward sweeps in place (sweep.hip)
struct SweepSmoother : elag_hip_smoother_s {
    struct Segment { int first, count, max_rows, row_begin, row_end; };   // steps [first, first + count); max_rows > 0: one big level = schedule slots [row_begin, row_end) (multi-workgroup launch)
    int type = 2, sweeps = 1, num_levels = 0, n_steps = 0;
    double weight = 1.0, omega = 1.0;
    int *d_rows = nullptr, *d_lev_ptr = nullptr;      // schedule slots (row, first entry, length, 0) by (level, row); step offsets into it
    double *d_l1 = nullptr;                           // types 2 / 4
    int *d_sJ = nullptr;                              // the first 16 entries of every row in schedule order (columns, -1 past the end; values)
    double *d_sa = nullptr;
    std::vector<Segment> segments;
    ~SweepSmoother() override;
    void setup();
    // The dataflow sweep (sweep.hip, sweep_dataflow_kernel): ONE launch per symmetric sweep in which every row waits for the rows it depends on
    // and for nothing else -- no launch and no barrier per dependency level.  df_slots: (row | -1, first entry in df_J / df_a, 16-entry chunks, 0)
    // per schedule position, the levels padded to whole wave tiles of 4; df_J: column | flags per entry (rows padded to 16); df_gran: two arrays
    // of 2 n tagged 8-byte words (the row's value as two {epoch, half} granules), used in turn; df_state: done counter, array in use, abort word.
    bool dataflow = false;
    int *d_df_slots = nullptr;
    unsigned *d_df_J = nullptr;
    double *d_df_a = nullptr;
    unsigned long long *d_df_gran = nullptr;
    int *d_df_state = nullptr;
    int df_npos = 0, df_grid = 0;
    bool df_xcd = false;                              // the launch confines itself to one XCD (plain stores through that XCD's L2)
    void setup_dataflow(const std::vector<int> &I, const std::vector<int> &J, const std::vector<int> &level, const std::vector<int> &ptr, const std::vector<int> &rows);
    void launch_dataflow(const double *f, double *u);
    // setup() = the common part (virtual ranks, l1 norms), then the schedule: for large operators by kernels (setup_device: dependency levels by a
    // dataflow kernel in natural row order, rows dealt to their levels, entries and flags -- nothing but a few counts crosses to the host), otherwise, or
    // when a kernel path does not apply, the host's pass over the rows (setup_host), which also builds the launch-per-level plan.  host_ready: that plan
    // exists (built on demand for what needs it: apply_dense, a sweep with the dataflow form switched off).
    bool host_ready = false;
    std::vector<int> h_block_of;
    bool setup_device();
    void setup_host(bool allow_dataflow);
    void finish_dataflow(int nlev);
    bool lds_u = false;                               // the operator's u fits the chain workgroup's LDS: every level is chained (sweep.hip)
    // "virtual ranks" (elag_hip_set_gs_hybrid_blocks > 1, rank-local operators): the rows are cut into that many contiguous blocks and the
    // sweep is hypre's hybrid Gauss-Seidel as the reference runs it on that many MPI ranks -- Gauss-Seidel inside a block, the other
    // blocks' values frozen as they were when the symmetric sweep began.  As = the operator split like a ParCSR matrix on those ranks
    // (diag: the block-diagonal part, offd: the rest, its columns numbered through d_self_ghost); M = what the sweep runs on
    int blocks = 1;
    std::unique_ptr<ParCSR> As;
    int *d_self_ghost = nullptr;
    int *d_blk_first = nullptr, *d_blk_count = nullptr;    // per block: its first step and its number of steps (a workgroup per block walks them)
    ParCSR *M = nullptr;
    void launch_segment(const Segment &s, int dir, const double *f, double *u, bool sym = false);
    bool sor = false;                                 // hypre's weighted branches (relax_weight != 1; omega != 1 for type 6): a launch per dependency level
    double *d_v = nullptr;                            // ... and the vector the forward pass started from
    void sweep_sor(const double *f, double *u);
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool) const override { return 0; }
    bool graph_capturable() const override { return !is_distributed(*A); }   // (all scratch is allocated by setup())
    int longest_chain = 0;                            // steps of the longest per-block chain (virtual ranks with a workgroup per block)
    // the sweeps applied to m right-hand sides at once (X, F: n x m row-major; F == nullptr: the identity, m == n): what the hierarchy builds
    // the dense operator of its smallest levels with -- m columns per dependency level instead of m sweeps.  false: not available (Kaczmarz)
    bool apply_dense(const double *F, double *X, int m, bool zero_guess);
    std::vector<int> h_lev_ptr;                       // rows by dependency level (whatever the launch plan does with them): offsets ...
    int *d_lev_rows = nullptr;                        // ... into this list
    // (the dataflow sweep: a dependency level is one hand-off between workgroups, ~1 us against the ~2-3 us of a launch or a chain step)
    double chain_cost() const override
    {
        if (dataflow) return 2.0 * sweeps * 0.45 * (double)num_levels + 4.0;
        return 2.0 * sweeps * (double)(sor ? num_levels : longest_chain > 0 ? longest_chain : n_steps) + 2.0;
    }
    bool same_scaling_as(const elag_hip_smoother_s *o) const override
    {
        const SweepSmoother *r = dynamic_cast<const SweepSmoother *>(o);
        return r && r->A == A && r->type == type && r->sweeps == sweeps && r->weight == weight && r->omega == omega;
    }
};

// "Hypre Jacobi" = 413 -> mfem::HypreDiagScale (ParELAG_HypreSmootherFactory.cpp:28-31): sol = rhs ./ diag(A), whatever sol held
struct DiagScaleSmoother : elag_hip_smoother_s {
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool) const override { return 0; }
    bool graph_capturable() const override { return true; }
};

struct HiptmairSmoother : elag_hip_smoother_s {
    ParCSR *D = nullptr;
    elag_hip_smoother_s *primary = nullptr, *aux = nullptr;
    double *pv = nullptr, *auxb = nullptr, *auxx = nullptr, *auxalt = nullptr;
    // A W (the columns of A scaled by the primary l1-Jacobi weights) of a bandwidth-bound level: the sweep from the zero guess is then
    // pv = f - (A W) f, x = W f in ONE plain pass (gathering f and W per entry cost 16.9 against 11.1 us on the 39 MB H(div) x3 operator)
    std::unique_ptr<DevCSR> AW;
    ~HiptmairSmoother() override;
    void setup();
    void aux_correct(const double *resid, double *x);
    // the second half of a (non-transposed) sweep once the primary relaxation has been applied to x elsewhere: R = B - A X, X += D aux(D^T R)
    void finish_after_primary(const double *f, double *x);
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool zero_guess) const override { return primary->flips(zero_guess); }
    double chain_cost() const override { return primary->chain_cost() + aux->chain_cost() + 4.0; }
    bool graph_capturable() const override { return !is_distributed(*A) && !is_distributed(*D) && primary->graph_capturable() && aux->graph_capturable(); }
};

}  // namespace elag

struct elag_hip_hierarchy_s;
namespace elag {
// hands an object's destructor to the hierarchy (run, in reverse order, when the hierarchy is destroyed)
void hierarchy_adopt(elag_hip_hierarchy_s *H, std::function<void()> destroy);
// runs f as a named phase of the hierarchy's set-up (wall time up to a drain of the compute stream, added to the phase of that name: elag_hip_hierarchy_setup_breakdown)
void hierarchy_phase(elag_hip_hierarchy_s *H, const char *name, const std::function<void()> &f);
// Hierarchy::Mult on device vectors without the hierarchy's own graph replay (callers that capture their own graph)
void hierarchy_apply_raw(elag_hip_hierarchy_s *H, const double *rhs, double *sol, bool precond);
bool hierarchy_is_local(const elag_hip_hierarchy_s *H);   // no halo exchange / collective anywhere in the cycle
bool hierarchy_prefers_plain_launches(const elag_hip_hierarchy_s *H);   // Finalize measured the cycle faster as plain launches than as a replayed graph
bool hierarchy_internal_numbering(const elag_hip_hierarchy_s *H, const ParCSR *A_user, ParCSR **A_internal, const int **d_order0);
void hierarchy_apply_internal(elag_hip_hierarchy_s *H, const double *rhs_perm, double *sol_perm);
int hierarchy_height(const elag_hip_hierarchy_s *H);
}  // namespace elag

struct elag_hip_coarse_s {
    elag::ParCSR *A = nullptr;
    int kind = 0, n = 0;   // kind 0: Cholesky (SPD), 1: LU with partial pivoting (indefinite, e.g. saddle point)
    double *Minv = nullptr;  // dense inverse, row-major n x n (device)
    // replicated mode: n = global size, this rank applies rows [my0, my0 + nloc)
    bool replicated = false;
    int nloc = 0, my0 = 0;
    std::vector<int> counts, displs;
    double *full_rhs = nullptr;
    // sub-hierarchy mode (elag_hip_coarse_create_from_hierarchy): apply() = [all-reduce rhs] + replicated cycle
    std::function<void(const double *, double *)> custom_apply;
    elag_hip_hierarchy_s *sub = nullptr;          // ... that hierarchy (the owning hierarchy's distributed flattened cycle reads its level maps)
    int sub_allreduce = 0;
    // The dense inverse is a latency chain (one workgroup's 141 pivots: 0.22 ms; the blocked LU of a 1 471-dof saddle point: 5.9 ms) that the rest
    // of a hierarchy's build does not wait for until its dense tail is planned: with `async_stream` set before setup() it runs there, beside the
    // build's products on the compute stream, and finish() -- before anybody reads Minv -- waits for it and looks at its flag (kernels.hpp: DenseJob).
    hipStream_t async_stream = nullptr;
    elag::k::DenseJob job;
    void finish()
    {
        if (!job.finish) return;
        std::function<void()> f;
        f.swap(job.finish);
        f();
    }
    ~elag_hip_coarse_s()
    {
        try { finish(); } catch (...) {}
        elag::dev_free(Minv); elag::dev_free(full_rhs);
    }
    void setup();
    void apply(const double *rhs, double *sol);
};
