// parelag::Hierarchy on the device (src/linalg/solver_ops/ParELAG_Hierarchy.cpp:109-267) plus the
// hierarchy builder (buildHierarchyFromDeRhamSequence :282-383 + AMGeSolverFactory.cpp:26-169) and the
// measurement hooks used by bench.py.
#include <algorithm>
#include <cmath>

#include "comm.hpp"
#include "smoother.hpp"
#include "staging.hpp"

#include <chrono>
#include <cstring>

using namespace elag;

namespace {

struct Level {
    ParCSR *A = nullptr;
    ParCSR *P = nullptr;  // maps this level to the next finer one (null on level 0)
    elag_hip_smoother_s *pre = nullptr, *post = nullptr;
    double *resid = nullptr, *sol = nullptr, *alt = nullptr, *tmp_resid = nullptr;
    // launch-bound levels with one l1-Jacobi sweep before and after (see plan_fused): the level costs two launches
    //   down:  r_c  = G f,                      G  = P^T (I - A W),     W = diag(weight / l1)
    //   up:    x    = W (2 f - (A W f + S e)),  S  = A P - W^-1 P
    // which is algebraically the reference's  x = W f; r = f - A x; r_c = P^T r; x += P e; x += W (f - A x).
    std::shared_ptr<DevCSR> G, S;
    const double *fused_dinv = nullptr;
    // A P of this level when elag_hip_hierarchy_build formed it on the way to the Galerkin product and the level is a candidate for
    // the bandwidth-bound form of plan_fused (S is made from it there, in place, and this pointer is gone afterwards)
    std::shared_ptr<DevCSR> AP_kept;
    // A W (columns scaled by the pre-smoother's diagonal weights) of a bandwidth-bound level whose operator runs on the plain stream kernel (no
    // x-cache form): the sweep from the zero guess and its residual are then ONE plain pass, r = f - (A W) f with x = W f written beside it --
    // instead of a vector pass for W f and the product (the stream kernel's own fused form gathers f and W per entry: 16.9 against 11.1 us on
    // a 39 MB operator).  One more copy of the level's matrix.
    std::shared_ptr<DevCSR> AW;
    // the same l1-Jacobi level when its operators live on several ranks (see plan_fused): S = A P - W^-1 P as a distributed operator.  With
    // r = f - A W f from the way down, the way up is  x = W (f + r - S e):  no product with A, no halo exchange for x -- a level costs
    // one exchange for W f, the reverse exchange of P^T r and one for e (none where the coarse level is replicated) instead of four
    std::shared_ptr<ParCSR> Sd;
    const double *dist_w = nullptr;
    // launch-bound levels whose smoother is Hiptmair(l1-Jacobi, l1-Jacobi), one sweep each (see plan_fused): six launches
    // instead of ten, with  Gaux = D Waux D^T  (the auxiliary-space correction as one operator),  RA = P^T A  and  PTneg = -P^T
    // (restricted residual in one product pair),  S = A P - W^-1 P  (prolongation + primary post-sweep in one product pair)
    std::shared_ptr<DevCSR> Gaux, RA, PTneg;
    const double *hipt_w = nullptr;
    // ... and on the smallest of them (where the denser composites still cost one launch each) four launches instead of six:
    //   Kh = W + Gaux (I - A W)  (the whole pre-sweep from the zero guess, x = Kh f),   Mh = I - Gaux A  (x'' = Mh x' + Gaux f)
    std::shared_ptr<DevCSR> Kh, Mh;
    // launch-bound levels smoothed by one Chebyshev sweep of order <= 2 before and after (see plan_fused): the sweep is the linear map
    // u += Sc (f - A u),  Sc = Ds (c0 I + c1 Ds A Ds) Ds  (A's pattern), so the level is two products like the diagonal case:
    //   down:  r_c = G f,           G  = P^T - (P^T A) Sc
    //   up:    x   = U1 f + U2 e,   U1 = Sc (2 I - A Sc),   U2 = P - Sc (A P)
    std::shared_ptr<DevCSR> U1, U2;
};

struct GraphEntry {
    const double *rhs;
    double *sol;
    int mode;
    hipGraphExec_t exec;
};

}  // namespace

// What a Finalize may invest in a faster cycle is priced against the cycles the caller expects to run on the hierarchy: ELAG_HIP_EXPECTED_CYCLES
// (or ELAG_HIP_DENSE_TAIL_CYCLES, the name the first such decision gave it), read at every Finalize; default: many.
static double expected_cycles()
{
    for (const char *name : {"ELAG_HIP_EXPECTED_CYCLES", "ELAG_HIP_DENSE_TAIL_CYCLES"})
        if (const char *e = getenv(name))
            if (atof(e) > 0) return atof(e);
    return 1.0e9;
}

struct elag_hip_hierarchy_s {
    std::vector<Level> lev;
    elag_hip_coarse_s *coarse = nullptr;
    bool finalized = false;
    bool use_graph = false;
    bool all_local = true;
    bool dist_pieces = false;           // some operator is a rank's piece of a distributed one: Finalize's decisions are taken by all ranks together
    std::vector<GraphEntry> graphs;
    // objects created by elag_hip_hierarchy_build (owned)
    std::vector<ParCSR *> owned_mats;
    std::vector<elag_hip_smoother_s *> owned_smoothers;
    elag_hip_coarse_s *owned_coarse = nullptr;
    std::vector<std::function<void()>> adopted;   // destructors of objects built for this hierarchy elsewhere (blocked.hip)
    double *r0 = nullptr, *e0 = nullptr;  // solver-mode residual / correction on level 0
    // dense coarse-cycle operator: every level from dense_level down is replaced by ONE dense matrix B (the sub-cycle
    // is a linear map; B's columns are its images of the unit vectors, computed once in finalize), applied as a GEMV
    int dense_level = -1;
    double *Bdense = nullptr;
    // resident tail: the levels from resident_top down, when every one of them is a fused launch-bound level (G / S or the Hiptmair
    // composition) ending in a dense product, run as ONE kernel with grid-wide barriers between the products (kernels_spmv.hip)
    int resident_top = -1;
    k::ResidentSeq res_seq;
    // wall time of the setup phases (Finalize's, and elag_hip_hierarchy_build's before it), each closed by a drain of the compute stream:
    // what the reference reports as "Build Hierarchy" (ParELAG_AMGeSolverFactory.cpp:48-49,130-135), by phase
    std::vector<std::pair<const char *, double>> setup_ms;
    template <class F>
    void phase(const char *name, F &&f)
    {
        const auto t0 = std::chrono::steady_clock::now();
        f();
        stream_wait(ctx().stream);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (auto &p : setup_ms)
            if (!strcmp(p.first, name)) { p.second += ms; return; }
        setup_ms.emplace_back(name, ms);
    }
    // flattened tail: the fused launch-bound levels flat_top .. flat_bottom-1 above a dense product on level flat_bottom run as THREE
    // launches however many levels they are (plan_flat): everything that depends on the same inputs shares a launch
    int flat_top = -1, flat_bottom = -1;
    k::FlatSegs flat_s1, flat_s2;
    k::FlatSum flat_s3;
    std::vector<std::shared_ptr<DevCSR>> flat_ops;
    // distributed flattened cycle (plan_flat_dist): a row-partitioned l1-Jacobi level above a replicated sub-hierarchy as
    //   [pack + exchange of f] -> stage 1 -> [ONE all-reduce of the stacked coarse right-hand sides] -> stage 2 -> stage 3
    bool fd_active = false;
    k::FlatSegs fd_s1, fd_s2;
    k::FlatSum fd_s3;
    std::vector<std::shared_ptr<DevCSR>> fd_ops;
    double *fd_stack = nullptr, *fd_fext = nullptr;
    int fd_stack_n = 0;
    // graph replay or plain stream launches, by measurement at Finalize (tune_launch_mode): a replayed hipGraph costs a fixed 7-9 us per
    // launch on top of its nodes, which a cycle of a handful of kernels does not earn back (cube456 x3: 29 us replayed, 21.5 us launched)
    int tune_nodes = 0;
    double tune_graph_us = 0.0, tune_eager_us = 0.0;
    // internal renumbering (elag_hip_amge_options::reorder): order[l][new] = old
    bool reordered = false;
    ParCSR *A_user = nullptr;             // level-0 operator in the caller's numbering (solver-mode residual)
    std::vector<std::vector<int>> order;  // host copies, one per level (coarsest: made at build time; the others on request)
    std::vector<int *> d_order;           // device copies, one per level (d_order[0] == d_order0)
    int *d_order0 = nullptr;
    double *p_rhs = nullptr, *p_sol = nullptr;

    ~elag_hip_hierarchy_s()
    {
        for (auto &g : graphs) (void)hipGraphExecDestroy(g.exec);
        for (auto it = adopted.rbegin(); it != adopted.rend(); ++it) (*it)();
        for (auto &l : lev) { dev_free(l.resid); dev_free(l.sol); dev_free(l.alt); dev_free(l.tmp_resid); }
        k::resident_free(res_seq);
        for (size_t l = 1; l < d_order.size(); l++) dev_free(d_order[l]);
        dev_free(r0); dev_free(e0); dev_free(d_order0); dev_free(p_rhs); dev_free(p_sol); dev_free(Bdense);
        dev_free(fd_stack); dev_free(fd_fext);
        for (auto *s : owned_smoothers) delete s;
        delete owned_coarse;
        for (auto *m : owned_mats) delete m;
    }

    void finalize(bool graph)
    {
        AllocScope alloc_scope(!comm_active());
        const int nl = (int)lev.size();
        for (int l = 0; l < nl; l++) {
            Level &L = lev[(size_t)l];
            ELAG_REQUIRE(L.A, "Hierarchy::Finalize: level %d has no operator \"A\"", l);
            ELAG_REQUIRE(L.A->nrows() == L.A->ncols(), "level %d operator is not square", l);
            if (l > 0) {
                ELAG_REQUIRE(L.P, "Hierarchy::Finalize: level %d has no prolongator \"P\"", l);
                ELAG_REQUIRE(L.P->nrows() == lev[(size_t)l - 1].A->nrows() && L.P->ncols() == L.A->nrows(),
                             "P on level %d is %dx%d but fine/coarse sizes are %d/%d", l, L.P->nrows(), L.P->ncols(),
                             lev[(size_t)l - 1].A->nrows(), L.A->nrows());
                L.P->ensure_transpose();
            }
            const size_t n = (size_t)L.A->nrows();
            dev_free(L.resid); dev_free(L.sol); dev_free(L.alt); dev_free(L.tmp_resid);
            L.resid = dev_alloc<double>(n); L.sol = dev_alloc<double>(n); L.alt = dev_alloc<double>(n);
            L.tmp_resid = dev_alloc<double>(n);
        }
        dev_free(r0); dev_free(e0);
        r0 = dev_alloc<double>((size_t)lev[0].A->nrows());
        e0 = dev_alloc<double>((size_t)lev[0].A->nrows());
        if (reordered) {
            dev_free(p_rhs); dev_free(p_sol);
            p_rhs = dev_alloc<double>((size_t)lev[0].A->nrows());
            p_sol = dev_alloc<double>((size_t)lev[0].A->nrows());
        }
        for (auto &g : graphs) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
        resident_top = -1;
        k::resident_free(res_seq);
        bool local_only = true;   // every operator rank-local (no halo): graphs / fusion are safe even across ranks
        for (auto &L : lev) {
            if (L.A && L.A->needs_exchange()) local_only = false;
            if (L.P && L.P->needs_exchange()) local_only = false;
        }
        if (coarse && (coarse->replicated || coarse->custom_apply)) local_only = false;
        // A hierarchy whose operators are pieces of distributed ones is finalised by all of its ranks together, and what follows (plan_fused's
        // distributed products) is collective: the ranks must take the same branch even when one of them happens to have no off-rank coupling
        // (a user-assembled partition with an isolated sub-domain) -- agree on the AND of the ranks' views.
        dist_pieces = false;
        for (auto &L : lev) {
            if (L.A && is_distributed(*L.A)) dist_pieces = true;
            if (L.P && is_distributed(*L.P)) dist_pieces = true;
        }
        if (dist_pieces && comm_active()) {
            double not_local = local_only ? 0.0 : 1.0;
            comm_allreduce_scalar(&not_local, 1);
            local_only = not_local == 0.0;
        }
        all_local = local_only;
        use_graph = graph && local_only;
        phase("fused_level_operators", [&] { plan_fused(); });
        // SpMV forms that are built on first use must exist before the cycle is captured into a graph
        phase("spmv_forms", [&] {
            // (the x-cache window plans -- sampled for every large operator, built where the columns are reused -- and the stored A W are investments
            //  too: 1.5 ms of a 5.3 ms set-up at cube456 x5 for ~45 us a cycle; with fewer than 64 cycles expected the operators keep the stream kernel)
            const bool invest = expected_cycles() >= 64.0;
            auto prepare = [&](const DevCSR &M) { if (invest) k::prepare_spmv(M); else k::decline_spmv_forms(M); };
            for (size_t l = 0; l < lev.size(); l++) {
                Level &L = lev[l];
                prepare(L.A->diag);
                // (a bandwidth-bound level that goes up through S never multiplies by P itself)
                const bool p_unused = l > 0 && lev[l - 1].S && !lev[l - 1].G && !lev[l - 1].Gaux;
                if (L.P) { if (!p_unused) prepare(L.P->diag); if (L.P->diagT) prepare(*L.P->diagT); }
                if (L.G) prepare(*L.G);
                if (L.S) prepare(*L.S);
                if (L.U1) prepare(*L.U1);
                if (L.Kh) { prepare(*L.Kh); prepare(*L.Mh); }
                if (L.Gaux) prepare(*L.Gaux);
                if (L.RA) prepare(*L.RA);
            }
            if (A_user) prepare(A_user->diag);
            const char *e_aw = getenv("ELAG_HIP_LEVEL_AW");                  // (0: off; read at every Finalize)
            for (size_t l = 0; l + 1 < lev.size(); l++) {
                Level &L = lev[l];
                L.AW.reset();
                if ((e_aw && !atoi(e_aw)) || !invest) continue;
                const double *w = (L.pre && L.pre->A == L.A) ? L.pre->diag_scaling() : nullptr;
                if (!w || L.G || L.Gaux || L.U1 || L.Sd || L.A->needs_exchange() || L.A->offd.ncols != 0) continue;
                if (L.A->diag.nnz < (int64_t(1) << 20) || L.A->diag.xc_state == 1 || L.A->diag.lanes_per_row != 0) continue;
                if (dynamic_cast<HiptmairSmoother *>(L.pre)) continue;         // (that smoother keeps its own)
                L.AW.reset(new DevCSR());
                k::csr_copy(L.A->diag, *L.AW);
                k::csr_scale_cols(*L.AW, w);
                k::prepare_spmv(*L.AW);
                if (L.AW->xc_state == 1 || L.AW->lanes_per_row != 0) L.AW.reset();
            }
        });
        // (the coarsest operator's inverse may still be running on the second stream: the tail is the first to read it)
        if (coarse && coarse->job.finish) phase("coarse_solver_wait", [&] { coarse->finish(); });
        phase("dense_tail", [&] { plan_dense_tail(); });
        phase("flat_cycle_operators", [&] { plan_flat(); plan_resident(); plan_flat_dist(); });
        finalized = true;
        tune_graph_us = tune_eager_us = 0.0; tune_nodes = 0;
        if (use_graph) phase("launch_mode", [&] { tune_launch_mode(); });
    }

    // Replayed graph or plain stream launches?  A replayed hipGraph costs the DEVICE a fixed 5-9 us per launch on this stack (the gap in front of
    // the first node of every replay, profiles/r02_cycle_trace_k3_graph_vs_plain.txt) and saves the HOST ~2.5 us per kernel.  Every cycle of
    // up to 23 kernels that was timed both ways (round 2: H1 x3 / x4 / x5, Chebyshev, Darcy, H(div), H(curl), SPE10 size) ran faster as plain
    // launches, by 1-25 %; the Gauss-Seidel cycles -- hundreds to thousands of 2-4 us kernels -- are host-bound without a graph.  The rule
    // is therefore the kernel count: plain launches up to kPlainLaunchNodes kernels per cycle, a graph beyond.  It is deterministic (the
    // round-2 choice by timing at every Finalize differed from run to run on a busy host, and so did everything downstream of it).
    // ELAG_HIP_GRAPH_MODE=graph | eager forces either; =measure times one cycle both ways (host clock around a drained queue, best of
    // 3 x 40) and keeps plain launches when they are >= 3 % faster.
    static constexpr int kPlainLaunchNodes = 32;
    void tune_launch_mode()
    {
        const char *mode = getenv("ELAG_HIP_GRAPH_MODE");
        if (mode && !strcmp(mode, "graph")) return;
        if (mode && !strcmp(mode, "eager")) { use_graph = false; return; }
        const bool measure = mode && !strcmp(mode, "measure");
        const int n = lev[0].A->nrows();
        if (n == 0) return;
        hipStream_t st = ctx().stream;
        Temp<double> rhs((size_t)n), sol((size_t)n);
        k::vec_fill(n, 1.0, rhs);
        hipGraph_t graph = nullptr;
        ELAG_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        try { run(rhs, sol, true); } catch (...) { (void)hipStreamEndCapture(st, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
        ELAG_HIP_CHECK(hipStreamEndCapture(st, &graph));
        size_t nodes = 0;
        ELAG_HIP_CHECK(hipGraphGetNodes(graph, nullptr, &nodes));
        tune_nodes = (int)nodes;
        if (!measure) {
            (void)hipGraphDestroy(graph);
            if (nodes <= (size_t)kPlainLaunchNodes) use_graph = false;
            return;
        }
        if (nodes > 64) { (void)hipGraphDestroy(graph); return; }
        hipGraphExec_t exec = nullptr;
        ELAG_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        auto clock_us = [&](auto once) {
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                for (int i = 0; i < 5; i++) once();
                stream_wait(st);
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < 40; i++) once();
                stream_wait(st);
                best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 40.0);
            }
            return best;
        };
        tune_graph_us = clock_us([&] { ELAG_HIP_CHECK(hipGraphLaunch(exec, st)); });
        tune_eager_us = clock_us([&] { run(rhs, sol, true); });
        (void)hipGraphExecDestroy(exec);
        if (tune_eager_us < 0.97 * tune_graph_us) use_graph = false;
        if (getenv("ELAG_HIP_GRAPH_VERBOSE"))
            fprintf(stderr, "[elag_hip] cycle of %d kernels: %.1f us replayed as a graph, %.1f us as plain launches -> %s\n", tune_nodes, tune_graph_us,
                    tune_eager_us, use_graph ? "graph" : "plain launches");
    }

    void plan_dense_tail()
    {
        dense_level = -1;
        dev_free(Bdense);
        static int max_n = -1;
        if (max_n < 0) { const char *e = getenv("ELAG_HIP_DENSE_TAIL_N"); max_n = e ? atoi(e) : 1024; }
        const int nl = (int)lev.size();
        if (!all_local || nl < 3 || max_n <= 0) return;
        int L = -1;
        for (int l = 1; l < nl - 1; l++)   // the coarsest level alone is already a single dense product
            if (lev[(size_t)l].A->nrows() <= max_n && lev[(size_t)l].A->offd.ncols == 0) { L = l; break; }
        // Gauss-Seidel levels are dependency chains (a level of the cube456 x3 hierarchy with 5 739 dofs: two sweeps of 2 x 51 steps, ~0.3 ms)
        // where a dense product streams n^2 doubles (0.05 ms at that size): the tail may start higher up, wherever the product is the cheaper
        // of the two and the operator fits ELAG_HIP_DENSE_TAIL_N_SWEEPS (8 192: 0.5 GB)
        {
            static int max_ns = -1;
            if (max_ns < 0) { const char *e = getenv("ELAG_HIP_DENSE_TAIL_N_SWEEPS"); max_ns = e ? atoi(e) : 8192; }
            double below = 2.0;                                   // dependent device steps of the sub-cycle from level l down (~1.5 us each)
            int best = -1;
            for (int l = nl - 2; l >= 1; l--) {
                const Level &F = lev[(size_t)l];
                if (!dynamic_cast<SweepSmoother *>(F.pre) || !dynamic_cast<SweepSmoother *>(F.post) || F.A->offd.ncols != 0) break;
                below += F.pre->chain_cost() + F.post->chain_cost() + 4.0;
                const double nn = (double)F.A->nrows();
                if (nn > max_ns) break;
                const double dense_us = nn * nn * 8.0 / 5.0e6, chains_us = below * 1.5;      // (bytes / (5 TB/s) against steps x 1.5 us)
                if (!(dense_us < chains_us)) continue;
                // ... and the product has to be BUILT: four half-sweeps over the n x n block, nnz(A) x n entries of 8 bytes each at ~3 TB/s (6.5 ms for
                // the 5 739-dof level of cube456 x3, against 0.2-0.5 ms a cycle saved).  ELAG_HIP_EXPECTED_CYCLES (ELAG_HIP_DENSE_TAIL_CYCLES, its first
                // name) = the number of cycles the caller expects to run on this hierarchy (default: many -- a tail wherever it is the cheaper cycle);
                // with a small number the tail starts only where building it pays within that many cycles -- one PCG solve of the example is 6-16 cycles.
                const double horizon = expected_cycles();                // (read at every Finalize)
                const double build_us = 4.0 * (double)F.A->diag.nnz * nn * 8.0 / 3.0e6;
                if (build_us <= horizon * (chains_us - dense_us)) best = l;
            }
            if (best >= 1 && (L < 0 || best < L)) {
                if (dense_tail_by_sweeps(best)) { dense_level = best; return; }
            }
        }
        if (L < 0) return;
        const int n = lev[(size_t)L].A->nrows();
        if (n == 0) return;
        hipStream_t st = ctx().stream;
        if (dense_tail_by_recursion(L)) { dense_level = L; return; }
        if (dense_tail_by_sweeps(L)) { dense_level = L; return; }
        // general levels: column j of B = sub-cycle applied to e_j -- n applications of the sub-cycle, which must not cost more than the tail
        // will ever give back: with Gauss-Seidel sweeps (a dependency level per ~2 us step) 840 applications were 266 ms of a 284 ms
        // Finalize at cube456 x3 for 0.25 ms per cycle, against 15 ms for the example's whole PCG solve.  Priced in dependent device steps
        // (smoother.hpp, chain_cost); ELAG_HIP_DENSE_TAIL_STEPS (default 60 000, ~0.1 s) is the most the loop may take.
        {
            static long long budget = -1;
            if (budget < 0) { const char *e = getenv("ELAG_HIP_DENSE_TAIL_STEPS"); budget = e ? atoll(e) : 60000; }
            double steps = 2.0;                                  // (the coarse solve)
            for (int l = L; l < nl - 1; l++) {
                const Level &F = lev[(size_t)l];
                steps += (F.pre ? F.pre->chain_cost() : 0.0) + (F.post ? F.post->chain_cost() : 0.0) + 4.0;
            }
            if ((double)n * steps > (double)budget) return;
        }
        // stored transposed first (one contiguous row per application)
        double *Bt = dev_alloc<double>((size_t)n * n);
        Temp<double> ej((size_t)n);
        zero_async(ej, sizeof(double) * (size_t)n);
        for (int j = 0; j < n; j++) {
            k::vec_fill(1, 1.0, ej.p + j);           // (a kernel, not a copy from pageable host memory: that one is synchronous)
            iterate(L, ej, Bt + (size_t)j * n, lev[(size_t)L].alt);
            k::vec_fill(1, 0.0, ej.p + j);
        }
        Bdense = dev_alloc<double>((size_t)n * n);
        k::dense_transpose(n, Bt, Bdense);
        ELAG_HIP_CHECK(hipStreamSynchronize(st));
        dev_free(Bt);
        dense_level = L;
    }

    // Gauss-Seidel levels: the sweeps are linear too, and a sweep on n right-hand sides at once is n columns per dependency level instead of n
    // sweeps -- with one level above a dense coarse inverse,
    //   X1 = S(I; 0),   X2 = X1 + P B_below P^T (I - A X1),   B = S(I; X2)       (S(F; X0): the symmetric sweep on F from X0)
    // level by level from the dense coarse inverse up -- a few hundred small launches where 840 applications of the sub-cycle were 266 ms
    // of a 284 ms Finalize (cube456 x3).
    bool dense_tail_by_sweeps(int L)
    {
        const int nl = (int)lev.size();
        const char *en = getenv("ELAG_HIP_DENSE_TAIL_SWEEPS");            // (read at every Finalize)
        if ((en && !atoi(en)) || !coarse || coarse->custom_apply || coarse->replicated || !coarse->Minv) return false;
        if (coarse->n != lev[(size_t)nl - 1].A->nrows()) return false;
        for (int l = L; l < nl - 1; l++) {
            Level &F = lev[(size_t)l];
            Level &C = lev[(size_t)l + 1];
            SweepSmoother *pre = dynamic_cast<SweepSmoother *>(F.pre), *post = dynamic_cast<SweepSmoother *>(F.post);
            if (!pre || !post || pre->A != F.A || post->A != F.A || F.A->needs_exchange() || C.P->needs_exchange()) return false;
            if (pre->type == ELAG_HIP_SMOOTHER_KACZMARZ || post->type == ELAG_HIP_SMOOTHER_KACZMARZ) return false;
            if (C.P->nrows() != F.A->nrows() || C.P->ncols() != C.A->nrows()) return false;
        }
        const double *Bc = coarse->Minv;         // the dense operator of the sub-cycle below the level at hand
        double *own = nullptr;
        int nc = coarse->n;
        for (int l = nl - 2; l >= L; l--) {
            Level &F = lev[(size_t)l];
            Level &C = lev[(size_t)l + 1];
            SweepSmoother *pre = static_cast<SweepSmoother *>(F.pre), *post = static_cast<SweepSmoother *>(F.post);
            const int n = F.A->nrows();
            double *B = dev_alloc<double>((size_t)n * n);
            bool ok = pre->apply_dense(nullptr, B, n, true);                           // X1 = S(I; 0)
            if (ok) {
                Temp<double> R((size_t)n * n), T1((size_t)nc * n), T2((size_t)nc * n);
                k::csr_times_dense(F.A->diag, n, -1.0, B, 0.0, true, R);               // R = I - A X1
                C.P->ensure_transpose();
                k::csr_times_dense(*C.P->diagT, n, 1.0, R, 0.0, false, T1);            // T1 = P^T R
                k::dense_mm(nc, nc, n, Bc, T1, T2);                                    // T2 = B_below T1
                k::csr_times_dense(C.P->diag, n, 1.0, T2, 1.0, false, B);              // X2 = X1 + P T2
                // (no drain: the scratch is reused in stream order)
                ok = post->apply_dense(nullptr, B, n, false);                          // B = S(I; X2)
            }
            ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
            dev_free(own);
            if (!ok) { dev_free(B); return false; }
            own = B; Bc = B; nc = n;
        }
        Bdense = own;
        return true;
    }

    // The dense operator of the sub-cycle from level L down, level by level from the bottom, when every level on the way is one of the
    // fused forms with linear level maps  f_c = G f,  x = U f + V x_c  (l1-Jacobi: U = W (2 I - A W), V = -W S; Chebyshev: U1, U2) above a
    // dense inverse:   B_last = Minv,   B_l = U_l + V_l (B_{l+1} G_l)   -- one transpose and three small launches per level instead of n_L
    // applications of the sub-cycle to unit vectors (cube456 x3, n_L = 840: ~20 ms of a 24 ms Finalize).
    bool dense_tail_by_recursion(int L)
    {
        const char *en = getenv("ELAG_HIP_DENSE_TAIL_RECURSION");          // (read at every Finalize: a test switches it)
        const int nl = (int)lev.size();
        if ((en && !atoi(en)) || !coarse || coarse->custom_apply || coarse->replicated || !coarse->Minv || coarse->n != lev[(size_t)nl - 1].A->nrows()) return false;
        for (int l = L; l < nl - 1; l++) {
            const Level &F = lev[(size_t)l];
            if (F.Gaux || !F.G || !(F.U1 || (F.S && F.fused_dinv))) return false;
        }
        const double *Bc = coarse->Minv;
        double *own = nullptr;            // B of the level below (owned unless it is Minv)
        int nc = coarse->n;
        for (int l = nl - 2; l >= L; l--) {
            Level &F = lev[(size_t)l];
            const int n = F.A->nrows();
            DevCSR GT;
            k::csr_transpose(*F.G, GT);
            Temp<double> X((size_t)nc * n);
            k::dense_times_csr_t(nc, n, Bc, GT, X);
            double *B = dev_alloc<double>((size_t)n * n);
            if (F.U1) k::dense_level_op(n, nc, *F.U2, nullptr, X, F.U1.get(), nullptr, nullptr, B);
            else k::dense_level_op(n, nc, *F.S, F.fused_dinv, X, nullptr, &F.A->diag, F.fused_dinv, B);
            ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));      // (GT and X go out of scope)
            dev_free(own);
            own = B; Bc = B; nc = n;
        }
        Bdense = own;
        return true;
    }

    // the products iterate(l, ...) would launch from level l down and up again, as a list; false when a level on the way is not one of
    // the fused launch-bound forms (or a product would not take the row-group kernel)
    bool record(int l, const double *rhs, double *sol, double *alt, std::vector<k::ResidentOp> &ops)
    {
        Level &F = lev[(size_t)l];
        const int nl = (int)lev.size();
        k::ResidentOp op;
        if (l == dense_level) { k::resident_op_dense(F.A->nrows(), F.A->nrows(), Bdense, rhs, sol, op); ops.push_back(op); return true; }
        if (l == nl - 1) {
            if (!coarse || coarse->custom_apply || coarse->replicated || !coarse->Minv) return false;
            k::resident_op_dense(coarse->n, coarse->n, coarse->Minv, rhs, sol, op); ops.push_back(op);
            return true;
        }
        Level &C = lev[(size_t)l + 1];
        auto add = [&](const DevCSR &A, const DevCSR *offd, const k::SpmvArgs &a) {
            if (!k::resident_op_spmv(A, offd, a, op)) return false;
            ops.push_back(op);
            return true;
        };
        if (F.U1) {
            k::SpmvArgs d;
            d.x = rhs; d.alpha = 1.0; d.out = C.resid;
            if (!add(*F.G, nullptr, d)) return false;
            if (!record(l + 1, C.resid, C.sol, C.alt, ops)) return false;
            k::SpmvArgs u;
            u.x = rhs; u.xg = C.sol; u.alpha = 1.0; u.out = sol;
            return add(*F.U1, F.U2.get(), u);
        }
        if (F.G) {                                   // (the argument lists are iterate()'s)
            k::SpmvArgs d;
            d.x = rhs; d.alpha = 1.0; d.out = C.resid;
            if (!add(*F.G, nullptr, d)) return false;
            if (!record(l + 1, C.resid, C.sol, C.alt, ops)) return false;
            k::SpmvArgs u;
            u.x = rhs; u.cs = F.fused_dinv; u.xg = C.sol; u.alpha = -1.0; u.f = rhs; u.gamma = 2.0; u.s = F.fused_dinv; u.out = sol;
            return add(F.A->diag, F.S.get(), u);
        }
        if (F.Gaux && F.Kh) {
            double *cur = alt, *oth = sol;
            const double *w = F.hipt_w;
            k::SpmvArgs k1;
            k1.x = rhs; k1.alpha = 1.0; k1.out = cur;
            k::SpmvArgs a3;
            a3.x = cur; a3.xg = rhs; a3.alpha = -1.0; a3.out = C.resid;
            if (!add(*F.Kh, nullptr, k1) || !add(*F.RA, F.PTneg.get(), a3)) return false;
            if (!record(l + 1, C.resid, C.sol, C.alt, ops)) return false;
            k::SpmvArgs a5;
            a5.x = cur; a5.xg = C.sol; a5.alpha = -1.0; a5.f = rhs; a5.gamma = 1.0; a5.s = w; a5.yin = cur; a5.beta = 1.0; a5.out = F.tmp_resid;
            k::SpmvArgs m;
            m.x = F.tmp_resid; m.xg = rhs; m.alpha = 1.0; m.out = oth;
            return add(F.A->diag, F.S.get(), a5) && add(*F.Mh, F.Gaux.get(), m);
        }
        if (F.Gaux) {
            double *cur = alt, *oth = sol;
            const double *w = F.hipt_w;
            k::SpmvArgs a1;
            a1.x = rhs; a1.cs = w; a1.alpha = -1.0; a1.f = rhs; a1.gamma = 1.0; a1.out = F.tmp_resid; a1.out2 = cur;
            k::SpmvArgs a2;
            a2.x = F.tmp_resid; a2.alpha = 1.0; a2.yin = cur; a2.beta = 1.0; a2.out = cur;
            k::SpmvArgs a3;
            a3.x = cur; a3.xg = rhs; a3.alpha = -1.0; a3.out = C.resid;
            if (!add(F.A->diag, nullptr, a1) || !add(*F.Gaux, nullptr, a2) || !add(*F.RA, F.PTneg.get(), a3)) return false;
            if (!record(l + 1, C.resid, C.sol, C.alt, ops)) return false;
            k::SpmvArgs a5;
            a5.x = cur; a5.xg = C.sol; a5.alpha = -1.0; a5.f = rhs; a5.gamma = 1.0; a5.s = w; a5.yin = cur; a5.beta = 1.0; a5.out = oth;
            k::SpmvArgs a6;
            a6.x = oth; a6.alpha = -1.0; a6.f = rhs; a6.gamma = 1.0; a6.out = F.tmp_resid;
            k::SpmvArgs a7;
            a7.x = F.tmp_resid; a7.alpha = 1.0; a7.yin = oth; a7.beta = 1.0; a7.out = oth;
            return add(F.A->diag, F.S.get(), a5) && add(F.A->diag, nullptr, a6) && add(*F.Gaux, nullptr, a7);
        }
        return false;
    }

    void plan_resident()
    {
        // OFF unless ELAG_HIP_RESIDENT=1: measured on MI355X (cube456 x3, 5 products per cycle) the grid-wide barrier costs ~20 us -- an
        // agent-scope release / acquire has to write back and invalidate the eight XCDs' L2s, and 128-512 arrivals on one address are
        // serialised at the memory side -- against ~4.5 us for a kernel boundary inside a replayed graph: 0.107 ms per cycle at the best
        // grid (128 workgroups) against 0.029 ms for the five separate launches (CHANGELOG.md, round 2).  Kept for the record and for
        // hardware where the trade is different; results are bit-identical either way (tests/test_gpu_parity.py).
        const char *en = getenv("ELAG_HIP_RESIDENT");          // (read at every Finalize: a test switches it)
        if (!(en && atoi(en)) || !all_local) return;
        const int nl = (int)lev.size();
        // the highest level from which everything below is recordable; level 0 takes the caller's vectors (given at launch time)
        for (int top = 0; top < nl - 1; top++) {
            std::vector<k::ResidentOp> ops;
            const double *rhs = top == 0 ? k::resident_rhs() : lev[(size_t)top].resid;
            double *sol = top == 0 ? k::resident_sol() : lev[(size_t)top].sol;
            if (!record(top, rhs, sol, lev[(size_t)top].alt, ops) || ops.size() < 3) continue;
            k::resident_build(ops, res_seq);
            resident_top = top;
            if (getenv("ELAG_HIP_RESIDENT_VERBOSE"))
                fprintf(stderr, "[elag_hip] resident tail from level %d: %d products in one launch of %u workgroups\n", top, (int)ops.size(), res_seq.grid);
            return;
        }
    }

    // The levels of a launch-bound cycle are a chain of dependent launches, each about 4 us on MI355X whatever its size (two per fused
    // level, one for the dense tail).  With linear level maps
    //     f_{l+1} = G_l f_l,     x_l = U_l f_l + V_l x_{l+1}        (l1-Jacobi levels: U = W (2 I - A W), V = -W S; Chebyshev levels: U1, U2)
    // and a dense x_D = B f_D at the bottom, the same cycle is
    //     stage 1:  h = U_top f,   f_l = (G_{l-1} ... G_top) f            for l = top+1 .. D      (all read f:      ONE launch)
    //     stage 2:  t_l = U_l f_l  for l = top+1 .. D-1,   x_D = B f_D                                (all read stage 1: ONE launch)
    //     stage 3:  x_top = h + sum_l (V_top ... V_{l-1}) t_l + (V_top ... V_{D-1}) x_D              (ONE launch)
    // with the products of G and of V formed once here by the device SpGEMM: three dependent launches instead of 2 (D - top) + 1, for
    // more bytes (cube456 x3: 5 launches -> 3, 1.9 M -> 2.7 M stored entries, all cache resident).  Same linear map, other rounding
    // (parity with the oracle's level-by-level cycle stays ~1e-15).  ELAG_HIP_FLAT_CYCLE=0 keeps the level-by-level launches.
    void plan_flat()
    {
        flat_top = flat_bottom = -1;
        flat_ops.clear();
        const char *en = getenv("ELAG_HIP_FLAT_CYCLE");          // (read at every Finalize: tests switch it)
        const char *res = getenv("ELAG_HIP_RESIDENT");
        if ((en && !atoi(en)) || (res && atoi(res)) || !all_local) return;
        const int nl = (int)lev.size();
        int D = -1, nd = 0;
        const double *Md = nullptr;
        if (dense_level >= 0) { D = dense_level; Md = Bdense; nd = lev[(size_t)D].A->nrows(); }
        else if (coarse && !coarse->custom_apply && !coarse->replicated && coarse->Minv) { D = nl - 1; Md = coarse->Minv; nd = coarse->n; }
        if (D < 2 || !Md || nd == 0) return;
        auto fused = [&](int l) {
            const Level &F = lev[(size_t)l];
            if (F.Gaux || !F.G) return false;
            return (bool)F.U1 || (F.S && F.fused_dinv);
        };
        int top_min = D;
        while (top_min > 0 && D - (top_min - 1) <= 4 && fused(top_min - 1)) top_min--;
        // The composites have to be BUILT (0.7-0.9 ms of products at cube456 x3, where they save two launches = ~5 us a cycle: 140 cycles, ten
        // solves of the example, before they have paid): with the number of cycles the caller expects to run on this hierarchy
        // (expected_cycles(): ELAG_HIP_EXPECTED_CYCLES, default many) below that, the level-by-level launches stay.
        if (D - top_min >= 2 && expected_cycles() * 2.5 * (2.0 * (D - top_min) + 1.0 - 3.0) < 800.0) return;
        for (int top = top_min; D - top >= 2; top++) {
            const int m = D - top;
            try {
                std::vector<std::shared_ptr<DevCSR>> U((size_t)m), V((size_t)m), Gc((size_t)m), Vc((size_t)m);
                for (int i = 0; i < m; i++) {
                    Level &F = lev[(size_t)(top + i)];
                    if (F.U1) { U[(size_t)i] = F.U1; V[(size_t)i] = F.U2; continue; }
                    const int n = F.A->nrows();
                    const double *w = F.fused_dinv;
                    // U = 2 W - W A W: on a copy of A in one launch where every row stores its diagonal (the general route otherwise)
                    U[(size_t)i].reset(new DevCSR());
                    k::csr_copy(F.A->diag, *U[(size_t)i]);
                    k::CsrCombine uop;
                    uop.c_scale = -1.0; uop.c_row = w; uop.c_col = w; uop.diag = w; uop.d_scale = 2.0;
                    if (!k::csr_combine_inplace(*U[(size_t)i], uop)) {
                        DevCSR T, Wd;
                        k::csr_copy(F.A->diag, T);
                        k::csr_scale_rows(T, w);
                        k::csr_scale_cols(T, w);
                        k::csr_diagonal(n, w, 1.0, Wd);
                        k::csr_add(2.0, Wd, -1.0, T, *U[(size_t)i]);
                    }
                    Temp<double> negw((size_t)n);
                    k::vec_scale_mul(n, -1.0, w, nullptr, negw);
                    V[(size_t)i].reset(new DevCSR());
                    k::csr_copy(*F.S, *V[(size_t)i]);
                    k::csr_scale_rows(*V[(size_t)i], negw);
                }
                Gc[0] = lev[(size_t)top].G;
                Vc[0] = V[0];
                long long extra = 0;
                const long long budget = std::max<long long>(1ll << 20, 4 * lev[(size_t)top].A->diag.nnz);
                // (The rows of G_l ... G_top are long -- cube456 x3: 840 rows of ~512 entries from 1 300-6 000 products each, the SpGEMM's
                // one-row-per-workgroup path, 1.4 ms for its two launches.  Forming the product transposed on the short-row path and transposing
                // it back was tried: 0.2 ms SLOWER, the transpose's per-row column sort is the same long-row work.)
                for (int i = 1; i < m; i++) {
                    Gc[(size_t)i].reset(new DevCSR());
                    k::csr_spgemm(*lev[(size_t)(top + i)].G, *Gc[(size_t)i - 1], *Gc[(size_t)i], false);
                    Vc[(size_t)i].reset(new DevCSR());
                    k::csr_spgemm(*Vc[(size_t)i - 1], *V[(size_t)i], *Vc[(size_t)i], false);
                    extra += Gc[(size_t)i]->nnz + Vc[(size_t)i]->nnz;
                    if (extra > budget) fail(ELAG_HIP_ERR_STATE, "flattened cycle: composites outgrow their budget");
                }
                const double *rhs = k::resident_rhs();
                double *sol = k::resident_sol();
                double *h = lev[(size_t)top].tmp_resid;
                k::FlatSegs s1, s2;
                k::FlatSum s3;
                // (the segments with the longest rows first: their workgroups take longest and should not start last)
                for (int i = m - 1; i >= 0; i--) k::flat_segs_add_csr(s1, *Gc[(size_t)i], rhs, lev[(size_t)(top + i + 1)].resid);
                k::flat_segs_add_csr(s1, *U[0], rhs, h);
                k::flat_segs_add_dense(s2, nd, nd, Md, lev[(size_t)D].resid, lev[(size_t)D].sol);
                for (int i = m - 1; i >= 1; i--) k::flat_segs_add_csr(s2, *U[(size_t)i], lev[(size_t)(top + i)].resid, lev[(size_t)(top + i)].alt);
                for (int i = 0; i < m; i++) k::flat_sum_add(s3, *Vc[(size_t)i], i + 1 < m ? lev[(size_t)(top + i + 1)].alt : lev[(size_t)D].sol);
                k::flat_sum_finish(s3, h, sol);
                flat_s1 = s1; flat_s2 = s2; flat_s3 = s3;
                for (auto &v : {U, V, Gc, Vc})
                    for (auto &q : v) flat_ops.push_back(q);
                flat_top = top; flat_bottom = D;
                ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
                if (getenv("ELAG_HIP_FLAT_VERBOSE"))
                    fprintf(stderr, "[elag_hip] flattened cycle: levels %d..%d in three launches (%u + %u + %u workgroups, %lld composite entries)\n", top, D,
                            s1.grid, s2.grid, s3.grid, extra);
                return;
            } catch (const Error &) {
                // a product's rows outgrow what the SpGEMM handles, or the composites their budget: flatten fewer levels
                flat_ops.clear();
            }
        }
    }

    // the linear maps of a fused level,  x_l = U f_l + V x_{l+1}:  l1-Jacobi  U = 2 W - W A W, V = -W S;  Chebyshev  U1, U2  (see plan_flat)
    static bool level_maps(Level &F, std::shared_ptr<DevCSR> &U, std::shared_ptr<DevCSR> &V)
    {
        if (F.U1) { U = F.U1; V = F.U2; return true; }
        if (!(F.S && F.fused_dinv && F.G) || F.Gaux) return false;
        const int n = F.A->nrows();
        const double *w = F.fused_dinv;
        U.reset(new DevCSR());
        k::csr_copy(F.A->diag, *U);
        k::CsrCombine uop;
        uop.c_scale = -1.0; uop.c_row = w; uop.c_col = w; uop.diag = w; uop.d_scale = 2.0;
        if (!k::csr_combine_inplace(*U, uop)) {
            DevCSR T, Wd;
            k::csr_copy(F.A->diag, T);
            k::csr_scale_rows(T, w);
            k::csr_scale_cols(T, w);
            k::csr_diagonal(n, w, 1.0, Wd);
            k::csr_add(2.0, Wd, -1.0, T, *U);
        }
        Temp<double> negw((size_t)n);
        k::vec_scale_mul(n, -1.0, w, nullptr, negw);
        V.reset(new DevCSR());
        k::csr_copy(*F.S, *V);
        k::csr_scale_rows(*V, negw);
        // (no drain: the scratch is reused in stream order)
        return true;
    }

    // The flattened cycle for the layout of the multi-rank runs: level 0 row-partitioned (l1-Jacobi, one sweep before and after), the levels below
    // complete on every rank as a hierarchy of their own behind one all-reduce (distributed.py, elag_hip_coarse_create_from_hierarchy).
    // Level by level that is 7 launches -- A, P^T | G, dense, up | S -- before any communication (one rank through it: 31.5 k V-cycles/s against
    // 60 k for the one-rank path).  With the level maps  f_{l+1} = G_l f_l,  x_l = U_l f_l + V_l x_{l+1}  it becomes
    //     pack + exchange of f                                                      (ONE neighbour exchange, as before)
    //     stage 1:  h = U_0 f,   partial f_l = (G_{l-1} ... G_0)|my columns f     for l = 1 .. D, stacked          (one launch)
    //     all-reduce of the stack                                                   (ONE all-reduce, as before: of every f_l instead of f_1)
    //     stage 2:  t_l = U_l f_l,   x_D = B f_D                                   on every rank                    (one launch)
    //     stage 3:  x = h + sum_l (V_0 ... V_{l-1})|my rows t_l                                                     (one launch)
    // G_0 = P^T (I - A W) and U_0 = 2 W - W A W act on [own | ghost] entries of f (A's diag and offd blocks side by side, W's ghost entries
    // fetched once here); V_0 = -W S is the rows this rank owns of the distributed S = A P - W^-1 P that plan_fused formed (global coarse
    // columns, no ghosts: the coarse level is replicated).  Same linear map as the level-by-level cycle, other rounding.  Taken only when every
    // rank can (agreed by an all-reduce); ELAG_HIP_FLAT_DIST=0 keeps the level-by-level launches.
    void plan_flat_dist()
    {
        fd_active = false;
        fd_ops.clear();
        dev_free(fd_stack); dev_free(fd_fext);
        fd_stack_n = 0;
        const char *en = getenv("ELAG_HIP_FLAT_DIST");            // (read at every Finalize: tests switch it)
        if (lev.size() != 2 || !coarse || !coarse->sub || !coarse->sub_allreduce) return;
        Level &F = lev[0];
        Level &C = lev[1];
        elag_hip_hierarchy_s *S = coarse->sub;
        const int nlS = (int)S->lev.size();
        int D = -1, nd = 0;
        const double *Md = nullptr;
        if (S->dense_level >= 0) { D = S->dense_level; Md = S->Bdense; nd = S->lev[(size_t)D].A->nrows(); }
        else if (S->coarse && !S->coarse->custom_apply && !S->coarse->replicated && S->coarse->Minv) { D = nlS - 1; Md = S->coarse->Minv; nd = S->coarse->n; }
        bool can = !(en && !atoi(en)) && F.Sd && F.dist_w && F.Sd->offd.ncols == 0 && S->finalized && S->all_local && !S->reordered && D >= 0 && Md && nd > 0 &&
                   D + 1 <= k::kFlatTerms && D + 2 <= k::kFlatSegs && !C.P->needs_exchange() && C.P->ncols() == S->lev[0].A->nrows() &&
                   F.A->diag.nnz + F.A->offd.nnz < (int64_t(1) << 20);
        for (int i = 0; can && i < D; i++) {
            const Level &L = S->lev[(size_t)i];
            can = L.G && !L.Gaux && (L.U1 || (L.S && L.fused_dinv));
        }
        // every rank or none (the stacked all-reduce has another length than the level-by-level one)
        if (dist_pieces && comm_active()) {
            double no = can ? 0.0 : 1.0;
            comm_allreduce_scalar(&no, 1);
            can = no == 0.0;
        }
        if (!can) {
            if (getenv("ELAG_HIP_FLAT_VERBOSE"))
                fprintf(stderr, "[elag_hip] distributed flattened cycle not applicable (Sd %d, sub finalised %d local %d reordered %d, dense level %d of %d, local entries %lld)\n", F.Sd ? 1 : 0,
                        (int)S->finalized, (int)S->all_local, (int)S->reordered, D, nlS, (long long)(F.A->diag.nnz + F.A->offd.nnz));
            return;
        }
        double failed = 0.0;
        try {
            const int n = F.A->nrows(), ng = F.A->offd.ncols, next = n + ng, m = D + 1;
            const double *w = F.dist_w;
            // W on [own | ghost] entries
            Temp<double> wext((size_t)std::max(next, 1));
            dev_copy(wext.p, w, sizeof(double) * (size_t)n, ctx().stream);
            par_halo_exchange(*F.A, w);                                                          // (collective: every rank is here)
            if (ng) dev_copy(wext.p + n, F.A->d_ghost, sizeof(double) * (size_t)ng, ctx().stream);
            std::vector<std::shared_ptr<DevCSR>> U((size_t)m), V((size_t)m), Gc((size_t)m), Vc((size_t)m);
            DevCSR Aloc;                                                                         // [diag | offd], n x (n + ng)
            csr_hstack(F.A->diag, nullptr, 0, ng ? &F.A->offd : nullptr, nullptr, n, next, false, Aloc);
            C.P->ensure_transpose();
            {   // G_0 = [P^T | 0] - (P^T [diag | offd]) W
                DevCSR PText;
                k::csr_copy(*C.P->diagT, PText);
                PText.ncols = next;
                std::shared_ptr<DevCSR> RA(new DevCSR());
                k::csr_spgemm(*C.P->diagT, Aloc, *RA);
                k::CsrCombine g;
                g.c_scale = -1.0; g.c_col = wext; g.B = &PText; g.b_scale = 1.0;
                if (k::csr_combine_inplace(*RA, g)) {
                    RA->choose_spmv_shape();
                    Gc[0] = RA;
                } else {
                    DevCSR RA2;
                    k::csr_spgemm(*C.P->diagT, Aloc, RA2);
                    k::csr_scale_cols(RA2, wext);
                    Gc[0].reset(new DevCSR());
                    k::csr_add(1.0, PText, -1.0, RA2, *Gc[0]);
                }
            }
            {   // U_0 = 2 W - W [diag | offd] W
                U[0].reset(new DevCSR());
                k::csr_copy(Aloc, *U[0]);
                k::CsrCombine uop;
                uop.c_scale = -1.0; uop.c_row = w; uop.c_col = wext; uop.diag = w; uop.d_scale = 2.0;
                if (!k::csr_combine_inplace(*U[0], uop)) fail(ELAG_HIP_ERR_STATE, "distributed flattened cycle: a row of the fine operator stores no diagonal entry");
            }
            {   // V_0 = -W S (this rank's rows, global coarse columns)
                Temp<double> negw((size_t)n);
                k::vec_scale_mul(n, -1.0, w, nullptr, negw);
                V[0].reset(new DevCSR());
                k::csr_copy(F.Sd->diag, *V[0]);
                k::csr_scale_rows(*V[0], negw);
                // (no drain: the scratch is reused in stream order)
            }
            for (int i = 1; i < m; i++)
                if (!level_maps(S->lev[(size_t)i - 1], U[(size_t)i], V[(size_t)i])) fail(ELAG_HIP_ERR_STATE, "distributed flattened cycle: level without linear maps");
            Vc[0] = V[0];
            long long extra = 0;
            const long long budget = std::max<long long>(1ll << 21, 8 * (F.A->diag.nnz + F.A->offd.nnz));
            for (int i = 1; i < m; i++) {
                Gc[(size_t)i].reset(new DevCSR());
                k::csr_spgemm(*S->lev[(size_t)i - 1].G, *Gc[(size_t)i - 1], *Gc[(size_t)i], false);
                Vc[(size_t)i].reset(new DevCSR());
                k::csr_spgemm(*Vc[(size_t)i - 1], *V[(size_t)i], *Vc[(size_t)i], false);
                extra += Gc[(size_t)i]->nnz + Vc[(size_t)i]->nnz;
                if (extra > budget) fail(ELAG_HIP_ERR_STATE, "distributed flattened cycle: composites outgrow their budget");
            }
            // the stacked coarse right-hand sides f_1 .. f_D (one all-reduce) -- f_l has the height of sub level l - 1
            std::vector<int> off((size_t)m + 1, 0);
            for (int i = 1; i <= m; i++) off[(size_t)i] = off[(size_t)i - 1] + S->lev[(size_t)i - 1].A->nrows();
            fd_stack_n = off[(size_t)m];
            fd_stack = dev_alloc<double>((size_t)fd_stack_n);
            if (F.A->needs_exchange()) fd_fext = dev_alloc<double>((size_t)next);
            const double *rhs = k::resident_rhs();
            double *sol = k::resident_sol();
            double *h = F.tmp_resid;
            k::FlatSegs s1, s2;
            k::FlatSum s3;
            for (int i = m - 1; i >= 0; i--) k::flat_segs_add_csr(s1, *Gc[(size_t)i], rhs, fd_stack + off[(size_t)i]);
            k::flat_segs_add_csr(s1, *U[0], rhs, h);
            k::flat_segs_add_dense(s2, nd, nd, Md, fd_stack + off[(size_t)m - 1], S->lev[(size_t)D].sol);
            for (int i = m - 1; i >= 1; i--) k::flat_segs_add_csr(s2, *U[(size_t)i], fd_stack + off[(size_t)i - 1], S->lev[(size_t)i - 1].alt);
            for (int i = 0; i < m; i++) k::flat_sum_add(s3, *Vc[(size_t)i], i + 1 < m ? S->lev[(size_t)i].alt : S->lev[(size_t)D].sol);
            k::flat_sum_finish(s3, h, sol);
            fd_s1 = s1; fd_s2 = s2; fd_s3 = s3;
            for (auto &v : {U, V, Gc, Vc})
                for (auto &q : v) fd_ops.push_back(q);
            ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
            if (getenv("ELAG_HIP_FLAT_VERBOSE"))
                fprintf(stderr, "[elag_hip] distributed flattened cycle: level 0 (%d rows + %d ghosts) over %d replicated levels in three launches (%u + %u + %u workgroups)\n", n, ng,
                        m, s1.grid, s2.grid, s3.grid);
        } catch (const Error &e) {
            failed = 1.0;
            if (getenv("ELAG_HIP_FLAT_VERBOSE")) fprintf(stderr, "[elag_hip] distributed flattened cycle not built: %s\n", e.what());
        }
        if (dist_pieces && comm_active()) comm_allreduce_scalar(&failed, 1);
        if (failed != 0.0) { fd_ops.clear(); dev_free(fd_stack); dev_free(fd_fext); fd_stack_n = 0; return; }
        fd_active = true;
    }

    void plan_fused()
    {
        // (read at every Finalize: tests switch them)
        const char *e_dis = getenv("ELAG_HIP_NO_LEVEL_FUSION"), *e_nnz = getenv("ELAG_HIP_FUSE_NNZ"), *e_big = getenv("ELAG_HIP_BIG_LEVEL_S");
        const int disabled = (e_dis && atoi(e_dis)) ? 1 : 0;
        const long long max_nnz = e_nnz ? atoll(e_nnz) : (1ll << 20);
        const int big_s = e_big ? atoi(e_big) : 1;
        const int nl = (int)lev.size();
        for (int l = 0; l + 1 < nl; l++) {
            Level &F = lev[(size_t)l];
            Level &C = lev[(size_t)l + 1];
            F.G.reset(); F.S.reset(); F.fused_dinv = nullptr;
            F.Sd.reset(); F.dist_w = nullptr;
            std::shared_ptr<DevCSR> kept;                 // (used by the bandwidth-bound l1-Jacobi branch or dropped here)
            kept.swap(F.AP_kept);
            F.Gaux.reset(); F.RA.reset(); F.PTneg.reset(); F.hipt_w = nullptr;
            F.U1.reset(); F.U2.reset(); F.Kh.reset(); F.Mh.reset();
            // ---- Chebyshev of order <= 2 on a launch-bound level below the finest ----
            if (RelaxSmoother *cs = (!disabled && all_local && l >= 1 && F.pre && F.post) ? dynamic_cast<RelaxSmoother *>(F.pre) : nullptr) {
                RelaxSmoother *cp = dynamic_cast<RelaxSmoother *>(F.post);
                if (cs->type == ELAG_HIP_SMOOTHER_CHEBYSHEV && cp && cp->type == ELAG_HIP_SMOOTHER_CHEBYSHEV && cs->A == F.A && cp->A == F.A &&
                    cs->sweeps == 1 && cp->sweeps == 1 && cs->cheby_order <= 1 && cp->cheby_order == cs->cheby_order &&
                    cs->coefs[0] == cp->coefs[0] && cs->coefs[1] == cp->coefs[1] && F.A->offd.ncols == 0 && C.P->offd.ncols == 0 &&
                    F.A->diag.nnz < max_nnz / 4) {
                    const int n = F.A->nrows();
                    C.P->ensure_transpose();
                    try {
                    // Sc = c0 Ds^2 + c1 Ds^2 A Ds^2
                    std::vector<double> hds = host_from_dev((const double *)cs->ds, (size_t)n);
                    std::vector<int> hi((size_t)n + 1), hj((size_t)n);
                    std::vector<double> ha((size_t)n);
                    for (int i = 0; i < n; i++) { hi[(size_t)i] = i; hj[(size_t)i] = i; ha[(size_t)i] = hds[(size_t)i] * hds[(size_t)i]; }
                    hi[(size_t)n] = n;
                    DevCSR Ds2, Sc;
                    Ds2.upload(n, n, hi.data(), hj.data(), ha.data());
                    Temp<double> ds2((size_t)n);
                    stage_h2d(ds2.p, ha.data(), sizeof(double) * (size_t)n);
                    if (cs->cheby_order == 1) {
                        DevCSR T;
                        k::csr_copy(F.A->diag, T);
                        k::csr_scale_rows(T, ds2);
                        k::csr_scale_cols(T, ds2);
                        k::csr_add(cs->coefs[0], Ds2, cs->coefs[1], T, Sc);
                    } else {
                        k::csr_add(cs->coefs[0], Ds2, 0.0, Ds2, Sc);
                    }
                    for (int i = 0; i < n; i++) ha[(size_t)i] = 1.0;
                    DevCSR Id;
                    Id.upload(n, n, hi.data(), hj.data(), ha.data());
                    // G = P^T - (P^T A) Sc
                    {
                        DevCSR RA, RAS;
                        k::csr_spgemm(*C.P->diagT, F.A->diag, RA, false);
                        k::csr_spgemm(RA, Sc, RAS, false);
                        F.G.reset(new DevCSR());
                        k::csr_add(1.0, *C.P->diagT, -1.0, RAS, *F.G);
                    }
                    // U1 = Sc (2 I - A Sc),  U2 = P - Sc (A P)
                    {
                        DevCSR AS, T2;
                        k::csr_spgemm(F.A->diag, Sc, AS, false);
                        k::csr_add(2.0, Id, -1.0, AS, T2);
                        F.U1.reset(new DevCSR());
                        k::csr_spgemm(Sc, T2, *F.U1);
                        DevCSR AP, SAP;
                        k::csr_spgemm(F.A->diag, C.P->diag, AP, false);
                        k::csr_spgemm(Sc, AP, SAP, false);
                        F.U2.reset(new DevCSR());
                        k::csr_add(1.0, C.P->diag, -1.0, SAP, *F.U2);
                    }
                    // (no drain: the scratch is reused in stream order)
                    continue;
                    } catch (const Error &) {
                        // a row of one of the products expands beyond what the SpGEMM handles (operators with long rows: AMGe coarse
                        // levels): this level keeps the separate sweeps
                        F.G.reset(); F.U1.reset(); F.U2.reset();
                    }
                }
            }
            // ---- Hiptmair(l1-Jacobi, l1-Jacobi) on a launch-bound level ----
            if (HiptmairSmoother *hs = (!disabled && all_local && F.pre && F.pre == F.post) ? dynamic_cast<HiptmairSmoother *>(F.pre) : nullptr) {
                const double *w = (hs->primary && hs->primary->A == F.A) ? hs->primary->diag_scaling() : nullptr;
                const double *wa = (hs->aux && hs->aux->A && hs->aux->n() == hs->D->ncols()) ? hs->aux->diag_scaling() : nullptr;
                if (hs->A == F.A && w && F.A->offd.ncols == 0 && C.P->offd.ncols == 0 && F.A->diag.nnz >= max_nnz && big_s) {
                    // bandwidth-bound Hiptmair level: the primary post-sweep goes up through S like the l1-Jacobi level below (the residual of
                    // the way down, f - A x after the whole pre-sweep, is what the restriction used); the auxiliary correction follows as it is
                    try {
                        std::shared_ptr<DevCSR> AP(new DevCSR());
                        k::csr_spgemm(F.A->diag, C.P->diag, *AP, false);
                        k::CsrCombine sop;
                        sop.B = &C.P->diag; sop.b_scale = -1.0; sop.b_row = w; sop.b_row_divide = 1;
                        if (AP->nnz <= F.A->diag.nnz + C.P->diag.nnz && k::csr_combine_inplace(*AP, sop)) {
                            AP->choose_spmv_shape();
                            F.S = AP;
                            F.hipt_w = w;
                        }
                    } catch (const Error &) {
                        F.S.reset(); F.hipt_w = nullptr;
                    }
                    continue;
                }
                if (hs->A == F.A && w && wa && !hs->D->needs_exchange() && hs->D->offd.ncols == 0 && F.A->offd.ncols == 0 && C.P->offd.ncols == 0 &&
                    F.A->diag.nnz < max_nnz) {
                    const int n = F.A->nrows();
                    hs->D->ensure_transpose();
                    C.P->ensure_transpose();
                    {   // Gaux = D (Waux D^T)
                        DevCSR WDT;
                        k::csr_copy(*hs->D->diagT, WDT);
                        k::csr_scale_rows(WDT, wa);
                        F.Gaux.reset(new DevCSR());
                        k::csr_spgemm(hs->D->diag, WDT, *F.Gaux);
                    }
                    F.RA.reset(new DevCSR());
                    k::csr_spgemm(*C.P->diagT, F.A->diag, *F.RA);
                    F.PTneg.reset(new DevCSR());
                    k::csr_add(-1.0, *C.P->diagT, 0.0, *C.P->diagT, *F.PTneg);
                    {   // S = A P - W^-1 P
                        DevCSR AP, WP;
                        k::csr_spgemm(F.A->diag, C.P->diag, AP);
                        k::csr_copy(C.P->diag, WP);
                        Temp<double> winv((size_t)n);
                        k::vec_reciprocal_scale(n, 1.0, w, winv);
                        k::csr_scale_rows(WP, winv);
                        F.S.reset(new DevCSR());
                        k::csr_add(1.0, AP, -1.0, WP, *F.S);
                    }
                    F.hipt_w = w;
                    static long long compose_nnz = -1;
                    if (compose_nnz < 0) { const char *e = getenv("ELAG_HIP_HIPTMAIR_COMPOSE_NNZ"); compose_nnz = e ? atoll(e) : (1ll << 17); }
                    if (F.A->diag.nnz < compose_nnz) {
                        try {
                            std::vector<double> hw = host_from_dev(w, (size_t)n);
                            std::vector<int> hi((size_t)n + 1), hj((size_t)n);
                            std::vector<double> one((size_t)n, 1.0);
                            for (int i = 0; i < n; i++) { hi[(size_t)i] = i; hj[(size_t)i] = i; }
                            hi[(size_t)n] = n;
                            DevCSR Wd, Id, GA, GAW, T;
                            Wd.upload(n, n, hi.data(), hj.data(), hw.data());
                            Id.upload(n, n, hi.data(), hj.data(), one.data());
                            k::csr_spgemm(*F.Gaux, F.A->diag, GA, false);
                            k::csr_copy(GA, GAW);
                            k::csr_scale_cols(GAW, w);
                            k::csr_add(1.0, Wd, 1.0, *F.Gaux, T);
                            F.Kh.reset(new DevCSR());
                            k::csr_add(1.0, T, -1.0, GAW, *F.Kh);
                            F.Mh.reset(new DevCSR());
                            k::csr_add(1.0, Id, -1.0, GA, *F.Mh);
                            // (no drain: the scratch is reused in stream order)
                        } catch (const Error &) {
                            F.Kh.reset(); F.Mh.reset();
                        }
                    }
                }
                continue;
            }
            const double *dinv = F.pre ? F.pre->diag_scaling() : nullptr;
            bool diag_level = !(disabled || !dinv || !F.post || !F.post->diag_scaling() || F.pre->A != F.A || F.post->A != F.A ||
                                !F.pre->same_scaling_as(F.post));
            if (!all_local) {
                // ---- l1-Jacobi level of a hierarchy that lives on several ranks ----
                const char *nf = getenv("ELAG_HIP_NO_DIST_LEVEL_FUSION");          // (read at every Finalize: a test switches it)
                if (nf && atoi(nf)) continue;                 // (one rank through this layout -- bench.py --force-dist -- takes the same branch: the products below need no peer)
                // the products below are collective: every rank enters them or none does (a rank's own smoother may differ), and a rank that
                // fails inside (memory, a row beyond the SpGEMM's limits) must not leave the others waiting in the next collective
                if (dist_pieces) {
                    double no = diag_level ? 0.0 : 1.0;
                    comm_allreduce_scalar(&no, 1);
                    diag_level = no == 0.0;
                }
                if (!diag_level) continue;
                const int n = F.A->nrows();
                std::shared_ptr<ParCSR> Sd;
                double failed = 0.0;
                std::string why;
                try {
                    std::unique_ptr<ParCSR> AP(distributed_matmul(*F.A, *C.P));
                    std::unique_ptr<ParCSR> WP(parcsr_clone(*C.P));
                    {
                        Temp<double> winv((size_t)n);
                        k::vec_reciprocal_scale(n, 1.0, dinv, winv);
                        k::csr_scale_rows(WP->diag, winv);
                        if (WP->offd.nnz) k::csr_scale_rows(WP->offd, winv);
                        // (no drain: the scratch is reused in stream order)
                    }
                    Sd.reset(distributed_add(1.0, *AP, -1.0, *WP));
                } catch (const Error &e) {
                    // (an error between two collectives of the products desynchronises the ranks for good: only what fails before or after
                    // them is recoverable, and that is what the agreement below covers)
                    failed = 1.0; why = e.what();
                }
                if (dist_pieces) comm_allreduce_scalar(&failed, 1);
                if (failed != 0.0) {
                    if (getenv("ELAG_HIP_FLAT_VERBOSE")) fprintf(stderr, "[elag_hip] level %d keeps the separate sweeps on every rank (%s)\n", l, why.empty() ? "another rank failed" : why.c_str());
                    continue;
                }
                F.Sd = Sd;
                F.dist_w = dinv;
                continue;
            }
            if (!diag_level) continue;
            if (F.A->offd.ncols > 0 || C.P->offd.ncols > 0) continue;
            if (F.A->diag.nnz >= max_nnz) {
                // ---- bandwidth-bound l1-Jacobi level: only the way up changes.  With r = f - A W f kept from the way down,
                //   x + P e + W (f - A (x + P e)) = x + W (r - S e),  S = A P - W^-1 P:
                // one pass over S instead of a pass over P (read and rewrite x) and a second pass over A -- whenever S is the smaller
                // object (nested P1: 9 entries per row against 15 + 2); the way down stays the reference's sequence, G would not pay there
                // (it replaces one pass over A by a pass over an operator with S's pattern transposed, and r would be gone)
                if (!big_s) continue;
                try {
                    std::shared_ptr<DevCSR> AP = kept;
                    if (!AP || AP->nrows != F.A->nrows() || AP->ncols != C.P->ncols()) {
                        AP.reset(new DevCSR());
                        k::csr_spgemm(F.A->diag, C.P->diag, *AP, false);
                    }
                    if (getenv("ELAG_HIP_FLAT_VERBOSE"))
                        fprintf(stderr, "[elag_hip] level %d: nnz(A) %lld + nnz(P) %lld against nnz(A P) %lld\n", l, (long long)F.A->diag.nnz, (long long)C.P->diag.nnz, (long long)AP->nnz);
                    if (AP->nnz > F.A->diag.nnz + C.P->diag.nnz) continue;
                    k::CsrCombine sop;
                    sop.B = &C.P->diag; sop.b_scale = -1.0; sop.b_row = dinv; sop.b_row_divide = 1;
                    if (!k::csr_combine_inplace(*AP, sop)) continue;
                    AP->choose_spmv_shape();
                    F.S = AP;
                    F.fused_dinv = dinv;
                } catch (const Error &) {
                    F.S.reset(); F.fused_dinv = nullptr;
                }
                continue;
            }
            const int n = F.A->nrows();
            C.P->ensure_transpose();
            // G = R - (R A) W  and  S = A P - W^-1 P.  With a full diagonal in A the pattern of R A contains R's and the pattern of A P contains
            // P's, so each is ONE in-place launch on the product (csr_combine_inplace); the general route -- scale, copy, sparse add through the
            // ESC engine, ~0.35 ms more per operator on the small levels -- stays for operators where an entry has no place
            {
                std::unique_ptr<DevCSR> RA(new DevCSR());
                k::csr_spgemm(*C.P->diagT, F.A->diag, *RA);
                k::CsrCombine g;
                g.c_scale = -1.0; g.c_col = dinv; g.B = C.P->diagT.get(); g.b_scale = 1.0;
                if (k::csr_combine_inplace(*RA, g)) {
                    RA->choose_spmv_shape();
                    F.G.reset(RA.release());
                } else {
                    DevCSR RA2;
                    k::csr_spgemm(*C.P->diagT, F.A->diag, RA2);
                    k::csr_scale_cols(RA2, dinv);
                    F.G.reset(new DevCSR());
                    k::csr_add(1.0, *C.P->diagT, -1.0, RA2, *F.G);
                }
            }
            {
                std::unique_ptr<DevCSR> AP(new DevCSR());
                k::csr_spgemm(F.A->diag, C.P->diag, *AP);
                k::CsrCombine sop;
                sop.B = &C.P->diag; sop.b_scale = -1.0; sop.b_row = dinv; sop.b_row_divide = 1;
                if (k::csr_combine_inplace(*AP, sop)) {
                    AP->choose_spmv_shape();
                    F.S.reset(AP.release());
                } else {
                    DevCSR AP2, WP;
                    k::csr_spgemm(F.A->diag, C.P->diag, AP2);
                    k::csr_copy(C.P->diag, WP);
                    Temp<double> winv((size_t)n);
                    k::vec_reciprocal_scale(n, 1.0, dinv, winv);
                    k::csr_scale_rows(WP, winv);
                    F.S.reset(new DevCSR());
                    k::csr_add(1.0, AP2, -1.0, WP, *F.S);
                }
            }
            F.fused_dinv = dinv;
        }
        ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
    }

    // Hierarchy::Iterate(RHS, SOL, l) with mu = 0; SOL is zero on entry by construction (:130, :212), which
    // the pre-smoother exploits (zero_guess).  The result is left in `sol`.
    void iterate(int l, const double *rhs, double *sol, double *alt)
    {
        Level &F = lev[(size_t)l];
        const int nl = (int)lev.size();
        if (l == flat_top) {
            k::flat_segs_launch(flat_s1, rhs, sol);
            k::flat_segs_launch(flat_s2, rhs, sol);
            k::flat_sum_launch(flat_s3, rhs, sol);
            return;
        }
        if (l == 0 && fd_active) { iterate_flat_dist(rhs, sol); return; }
        if (l == resident_top) { k::resident_launch(res_seq, rhs, sol); return; }
        if (l == dense_level) { k::dense_gemv(F.A->nrows(), F.A->nrows(), Bdense, rhs, sol); return; }
        if (l == nl - 1) {
            if (coarse) { coarse->apply(rhs, sol); return; }
            // no coarse solver: pre- then post-smoother (:160-172)
            int nf = (F.pre ? F.pre->flips(true) : 0) + (F.post ? F.post->flips(F.pre == nullptr) : 0);
            double *cur = (nf & 1) ? alt : sol, *oth = (nf & 1) ? sol : alt;
            if (!F.pre && !F.post) { k::vec_fill(F.A->nrows(), 0.0, sol); return; }
            if (F.pre) F.pre->apply(rhs, cur, oth, true, false);
            if (F.post) F.post->apply(rhs, cur, oth, F.pre == nullptr, false);
            return;
        }
        Level &C = lev[(size_t)l + 1];
        const int n = F.A->nrows();
        if (F.U1) {
            k::SpmvArgs d;  // restricted residual of the zero-guess Chebyshev sweep in one product
            d.x = rhs; d.alpha = 1.0; d.out = C.resid;
            k::spmv_fused(*F.G, nullptr, d);
            iterate(l + 1, C.resid, C.sol, C.alt);
            k::SpmvArgs u;  // coarse correction and the post-sweep: x = U1 f + U2 e
            u.x = rhs; u.xg = C.sol; u.alpha = 1.0; u.out = sol;
            k::spmv_fused(*F.U1, F.U2.get(), u);
            return;
        }
        if (F.G) {
            k::SpmvArgs d;  // restricted residual of the zero-guess Jacobi sweep in one product
            d.x = rhs; d.alpha = 1.0; d.out = C.resid;
            k::spmv_fused(*F.G, nullptr, d);
            iterate(l + 1, C.resid, C.sol, C.alt);
            k::SpmvArgs u;  // coarse correction and post-smoothing in one product pair
            u.x = rhs; u.cs = F.fused_dinv; u.xg = C.sol; u.alpha = -1.0; u.f = rhs; u.gamma = 2.0; u.s = F.fused_dinv; u.out = sol;
            k::spmv_fused(F.A->diag, F.S.get(), u);
            return;
        }
        if (F.Gaux) {
            // Hiptmair(l1-Jacobi, l1-Jacobi) pre / post on a launch-bound level, algebraically the reference's sequence
            //   x = W f; pv = f - A x; x += D Waux D^T pv | r_c = P^T (f - A x) | x += P e; x += W (f - A x); pv = f - A x; x += D Waux D^T pv
            double *cur = alt, *oth = sol;
            const double *w = F.hipt_w;
            if (F.Kh) {
                k::SpmvArgs k1;   // x = Kh f: the primary sweep from the zero guess and the auxiliary-space correction in one product
                k1.x = rhs; k1.alpha = 1.0; k1.out = cur;
                k::spmv_fused(*F.Kh, nullptr, k1);
            } else {
                k::SpmvArgs a1;   // x = W f (the operand, written as out2) and pv = f - A x in one product
                a1.x = rhs; a1.cs = w; a1.alpha = -1.0; a1.f = rhs; a1.gamma = 1.0; a1.out = F.tmp_resid; a1.out2 = cur;
                k::spmv_fused(F.A->diag, nullptr, a1);
                k::SpmvArgs a2;   // x += Gaux pv
                a2.x = F.tmp_resid; a2.alpha = 1.0; a2.yin = cur; a2.beta = 1.0; a2.out = cur;
                k::spmv_fused(*F.Gaux, nullptr, a2);
            }
            k::SpmvArgs a3;   // r_c = P^T f - (P^T A) x
            a3.x = cur; a3.xg = rhs; a3.alpha = -1.0; a3.out = C.resid;
            k::spmv_fused(*F.RA, F.PTneg.get(), a3);
            iterate(l + 1, C.resid, C.sol, C.alt);
            k::SpmvArgs a5;   // x' = x + W (f - A x - S e),  S = A P - W^-1 P
            a5.x = cur; a5.xg = C.sol; a5.alpha = -1.0; a5.f = rhs; a5.gamma = 1.0; a5.s = w; a5.yin = cur; a5.beta = 1.0;
            a5.out = F.Mh ? F.tmp_resid : oth;
            k::spmv_fused(F.A->diag, F.S.get(), a5);
            if (F.Mh) {
                k::SpmvArgs m;    // x'' = x' + Gaux (f - A x') = Mh x' + Gaux f
                m.x = F.tmp_resid; m.xg = rhs; m.alpha = 1.0; m.out = oth;
                k::spmv_fused(*F.Mh, F.Gaux.get(), m);
                return;
            }
            k::SpmvArgs a6;   // pv = f - A x'
            a6.x = oth; a6.alpha = -1.0; a6.f = rhs; a6.gamma = 1.0; a6.out = F.tmp_resid;
            k::spmv_fused(F.A->diag, nullptr, a6);
            k::SpmvArgs a7;   // x' += Gaux pv
            a7.x = F.tmp_resid; a7.alpha = 1.0; a7.yin = oth; a7.beta = 1.0; a7.out = oth;
            k::spmv_fused(*F.Gaux, nullptr, a7);
            return;
        }
        if (F.Sd) {
            const double *w = F.dist_w;
            if (F.A->diag.nnz < (int64_t(1) << 20)) {
                k::SpmvArgs a;   // r = f - A (W f): the sweep from the zero guess gathered on the fly, ONE halo exchange (of W f); W f kept for the way up
                a.x = rhs; a.cs = w; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = F.tmp_resid; a.out2 = alt;
                par_spmv(*F.A, a);
            } else {
                k::vec_scale_mul(n, 1.0, rhs, w, alt);
                k::SpmvArgs a;
                a.x = alt; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = F.tmp_resid;
                par_spmv(*F.A, a);
            }
            par_spmvT(*C.P, 1.0, F.tmp_resid, 0.0, C.resid);
            iterate(l + 1, C.resid, C.sol, C.alt);
            k::SpmvArgs u;       // x = W f + W (r - S e)  (= W (f + r - S e): the vector pass that formed f + r is gone, W f comes from the way down)
            u.x = C.sol; u.alpha = -1.0; u.f = F.tmp_resid; u.gamma = 1.0; u.s = w; u.yin = alt; u.beta = 1.0; u.out = sol;
            par_spmv(*F.Sd, u);
            return;
        }
        const int nf = (F.pre ? F.pre->flips(true) : 0) + (F.post ? F.post->flips(false) : 0);
        double *cur = (nf & 1) ? alt : sol, *oth = (nf & 1) ? sol : alt;
        // pre-smoothing (:180-184) and residual (:191-194)
        const double *pre_dinv = (F.pre && F.pre->A == F.A) ? F.pre->diag_scaling() : nullptr;
        // (worth it while the level is launch-bound -- the stream kernel's fused form gathers two vectors per entry -- and wherever the
        //  operator runs in its x-cache form, which gathers once per DISTINCT column of a window: there the separate pass that formed W f,
        //  13.8 us of a 386 us PCG iteration at cube456 x5, is one more gather per cached column)
        if (pre_dinv && (F.A->diag.nnz < (int64_t(1) << 20) || (F.A->diag.xc_state == 1 && !F.A->needs_exchange()))) {
            // one kernel: SOL = dinv .* RHS (Jacobi sweep from the zero guess) gathered on the fly as the SpMV
            // operand, tmp_resid = RHS - A SOL
            k::SpmvArgs a;
            a.x = rhs; a.cs = pre_dinv; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = F.tmp_resid; a.out2 = cur;
            par_spmv(*F.A, a);
        } else if (pre_dinv && F.AW) {
            k::SpmvArgs a;  // r = f - (A W) f, x = W f beside it
            a.x = rhs; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = F.tmp_resid; a.out2 = cur; a.o2s = pre_dinv;
            k::spmv_fused(*F.AW, nullptr, a);
        } else {
            if (F.pre) F.pre->apply(rhs, cur, oth, true, false);
            else k::vec_fill(n, 0.0, cur);
            k::SpmvArgs a;  // fused A*SOL and RHS - tmp
            a.x = cur; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = F.tmp_resid;
            par_spmv(*F.A, a);
        }
        // restriction P^T r (:199-203)
        par_spmvT(*C.P, 1.0, F.tmp_resid, 0.0, C.resid);
        // coarse-grid correction (:212-227)
        iterate(l + 1, C.resid, C.sol, C.alt);
        if (F.S && !F.G && !F.Gaux) {
            // bandwidth-bound l1-Jacobi level (plan_fused): prolongation and post-sweep in one pass over S, x' = x + W (r - S e);
            // Hiptmair: that is the primary relaxation of the post-sweep, the residual and the auxiliary-space correction follow
            k::SpmvArgs u;
            u.x = C.sol; u.alpha = -1.0; u.f = F.tmp_resid; u.gamma = 1.0; u.s = F.hipt_w ? F.hipt_w : F.fused_dinv; u.yin = cur; u.beta = 1.0; u.out = sol;
            k::spmv_fused(*F.S, nullptr, u);
            if (F.hipt_w) static_cast<HiptmairSmoother *>(F.post)->finish_after_primary(rhs, sol);
            return;
        }
        // SOL += P * coarse_sol (:231-238), fused
        {
            k::SpmvArgs a;
            a.x = C.sol; a.alpha = 1.0; a.yin = cur; a.beta = 1.0; a.out = cur;
            par_spmv(*C.P, a);
        }
        // post-smoothing (:244-249)
        if (F.post) F.post->apply(rhs, cur, oth, false, false);
        if (cur != sol) fail(ELAG_HIP_ERR_STATE, "internal: ping-pong parity mismatch on level %d", l);
    }

    void iterate_flat_dist(const double *rhs, double *sol)
    {
        Context &c = ctx();
        ParCSR &A = *lev[0].A;
        const double *x1 = rhs;
        if (A.needs_exchange()) {
            // f on [own | ghost] entries: the pack kernel also copies the own part, the exchange delivers the ghosts behind it
            const int n = A.nrows(), st = A.send_total();
            k::vec_pack_copy(n, st, A.d_send_elmts, rhs, A.d_send_buf, fd_fext);
            // a comm package that lists only this rank itself (periodic coupling; how one rank exercises the halo path) is a copy on the compute
            // stream: no second stream to order against, four event operations less per cycle on the host
            bool remote = false;
            for (int p : A.send_procs) remote = remote || p != c.rank;
            for (int p : A.recv_procs) remote = remote || p != c.rank;
            hipStream_t xs = remote ? c.comm_stream : c.stream;
            if (remote) {
                ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
                ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
            }
            comm_neighbor_exchange(A.d_send_buf, (int)A.send_procs.size(), A.send_procs.data(), A.send_starts.data(), fd_fext + n,
                                   (int)A.recv_procs.size(), A.recv_procs.data(), A.recv_starts.data(), sizeof(double), xs);
            if (remote) {
                ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
                ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
            }
            x1 = fd_fext;
        }
        k::flat_segs_launch(fd_s1, x1, sol);
        if (comm_active()) {
            ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
            ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
            comm_allreduce_device(fd_stack, fd_stack_n, 0, c.comm_stream);
            ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
            ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
        }
        k::flat_segs_launch(fd_s2, x1, sol);
        k::flat_sum_launch(fd_s3, x1, sol);
    }

    void run(const double *rhs, double *sol, bool precond)
    {
        Level &L0 = lev[0];
        const int n = L0.A->nrows();
        if (!reordered) {
            if (precond) {
                iterate(0, rhs, sol, L0.alt);
            } else {
                k::SpmvArgs a;  // r = rhs - A sol (:119-128)
                a.x = sol; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = r0;
                par_spmv(*L0.A, a);
                iterate(0, r0, e0, L0.alt);
                k::vec_axpby(n, 1.0, sol, 1.0, e0, sol);  // sol += correction (:134-135)
            }
            return;
        }
        // internal numbering: permute on the way in and out
        const double *src = rhs;
        if (!precond) {
            k::SpmvArgs a;
            a.x = sol; a.alpha = -1.0; a.f = rhs; a.gamma = 1.0; a.out = r0;
            par_spmv(*A_user, a);
            src = r0;
        }
        if (iterate0_permuted(src, sol, !precond)) return;
        k::vec_gather(n, d_order0, src, p_rhs);
        iterate(0, p_rhs, p_sol, L0.alt);
        k::vec_scatter(n, d_order0, p_sol, sol, precond ? 0 : 1);
    }

    // sol += cycle(rhs) from the zero guess, where that costs no pass of its own: a renumbered hierarchy leaves its numbering by a kernel
    // that can add as well as store
    bool adds_in_place() const { return reordered; }
    void run_add(const double *rhs, double *sol)
    {
        ELAG_REQUIRE(reordered, "internal: run_add() on a hierarchy in the caller's numbering");
        if (iterate0_permuted(rhs, sol, true)) return;
        const int n = lev[0].A->nrows();
        k::vec_gather(n, d_order0, rhs, p_rhs);
        iterate(0, p_rhs, p_sol, lev[0].alt);
        k::vec_scatter(n, d_order0, p_sol, sol, 1);
    }

    // Level 0 of a renumbered hierarchy with the permutations folded into its first and last kernel: the caller's right-hand side
    // is gathered by the kernel that starts the zero-guess Jacobi sweep, and the kernel that finishes the post-smoothing writes
    // (or adds to) the caller's numbering directly -- two passes over level-0 vectors less per cycle.  Only for the layout this is
    // worth it on (HBM-bound level 0 with one l1-Jacobi / Jacobi sweep before and after); false = not applicable.
    bool iterate0_permuted(const double *src_nat, double *sol_nat, bool add)
    {
        if (lev.size() < 2 || dense_level == 0) return false;
        Level &F = lev[0];
        Level &C = lev[1];
        if (F.G || F.Gaux || !F.pre || !F.post || F.pre->A != F.A || F.post->A != F.A) return false;
        const double *pre_dinv = F.pre->diag_scaling(), *post_dinv = F.post->diag_scaling();
        if (!pre_dinv || !post_dinv || F.A->needs_exchange() || F.A->diag.nnz < (int64_t(1) << 20)) return false;
        const int n = F.A->nrows();
        double *cur = F.alt;           // (never F.sol / F.resid: host callers' vectors are staged there, i.e. they may BE sol_nat / src_nat)
        // p_rhs = src[order], x = W p_rhs in one pass
        k::vec_gather_scale(n, d_order0, src_nat, pre_dinv, p_rhs, cur);
        {
            k::SpmvArgs a;  // r = f - A x
            a.x = cur; a.alpha = -1.0; a.f = p_rhs; a.gamma = 1.0; a.out = F.tmp_resid;
            par_spmv(*F.A, a);
        }
        par_spmvT(*C.P, 1.0, F.tmp_resid, 0.0, C.resid);
        iterate(1, C.resid, C.sol, C.alt);
        if (F.S) {
            k::SpmvArgs a;  // x' = x + W (r - S e) (plan_fused: prolongation and post-sweep in one pass over S), written to the caller's numbering
            a.x = C.sol; a.alpha = -1.0; a.f = F.tmp_resid; a.gamma = 1.0; a.s = F.fused_dinv; a.yin = cur; a.beta = 1.0;
            a.out = sol_nat; a.out_perm = d_order0; a.out_add = add ? sol_nat : nullptr;
            k::spmv_fused(*F.S, nullptr, a);
            return true;
        }
        {
            k::SpmvArgs a;  // x += P e
            a.x = C.sol; a.alpha = 1.0; a.yin = cur; a.beta = 1.0; a.out = cur;
            par_spmv(*C.P, a);
        }
        {
            k::SpmvArgs a;  // x' = x + W (f - A x), written to the caller's numbering: sol[order[i]] (+)= x'_i
            a.x = cur; a.alpha = -1.0; a.f = p_rhs; a.gamma = 1.0; a.s = post_dinv; a.yin = cur; a.beta = 1.0;
            a.out = sol_nat; a.out_perm = d_order0; a.out_add = add ? sol_nat : nullptr;
            par_spmv(*F.A, a);
        }
        return true;
    }

    void apply_device(const double *rhs, double *sol, bool precond)
    {
        if (!use_graph) { run(rhs, sol, precond); return; }
        hipStream_t st = ctx().stream;
        for (auto &g : graphs)
            if (g.rhs == rhs && g.sol == sol && g.mode == (int)precond) {
                ELAG_HIP_CHECK(hipGraphLaunch(g.exec, st));
                return;
            }
        // capture the whole cycle once per (rhs, sol) pair: the small levels are launch-bound
        hipGraph_t graph = nullptr;
        ELAG_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        try {
            run(rhs, sol, precond);
        } catch (...) {
            (void)hipStreamEndCapture(st, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            throw;
        }
        ELAG_HIP_CHECK(hipStreamEndCapture(st, &graph));
        hipGraphExec_t exec = nullptr;
        ELAG_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (graphs.size() >= 8) { (void)hipGraphExecDestroy(graphs.front().exec); graphs.erase(graphs.begin()); }
        graphs.push_back(GraphEntry{rhs, sol, (int)precond, exec});
        ELAG_HIP_CHECK(hipGraphLaunch(exec, st));
    }
};

namespace elag {
void hierarchy_adopt(elag_hip_hierarchy_s *H, std::function<void()> destroy) { H->adopted.push_back(std::move(destroy)); }
void hierarchy_phase(elag_hip_hierarchy_s *H, const char *name, const std::function<void()> &f) { H->phase(name, f); }
void hierarchy_apply_raw(elag_hip_hierarchy_s *H, const double *rhs, double *sol, bool precond)
{
    ELAG_REQUIRE(H->finalized, "Hierarchy::Mult before Finalize()");
    H->run(rhs, sol, precond);
}
bool hierarchy_is_local(const elag_hip_hierarchy_s *H) { return H->all_local; }
bool hierarchy_prefers_plain_launches(const elag_hip_hierarchy_s *H) { return H->tune_nodes > 0 && !H->use_graph; }
// A solver that iterates with a reordered hierarchy as its preconditioner can live in the hierarchy's numbering altogether:
// the level-0 operator as the cycle runs it (same matrix as `A_user`, rows / columns permuted) and the new -> old map.
bool hierarchy_internal_numbering(const elag_hip_hierarchy_s *H, const ParCSR *A_user, ParCSR **A_internal, const int **d_order0)
{
    if (!H->reordered || H->A_user != A_user || !H->d_order0) return false;
    *A_internal = H->lev[0].A;
    *d_order0 = H->d_order0;
    return true;
}
// preconditioner-mode cycle on vectors that already are in the internal numbering (no gather / scatter around it)
void hierarchy_apply_internal(elag_hip_hierarchy_s *H, const double *rhs_perm, double *sol_perm)
{
    ELAG_REQUIRE(H->finalized, "Hierarchy::Mult before Finalize()");
    H->iterate(0, rhs_perm, sol_perm, H->lev[0].alt);
}

// A hierarchy standing where the reference puts any mfem::Solver: the "A00 Inverse" / "S Inverse" entries of the
// Block2x2*SolverFactory lists name AMGe hierarchies (example_parameterlists/spe10_example_parameters.xml:114-127,
// "A00_1 Inverse" = "AMGe-GS").  scale multiplies the result: -1 applies  -(B D^-1 B^T)^-1  with a hierarchy built on the
// positive matrix  B D^-1 B^T  (the Schur complement of the saddle point is its negative).
struct HierarchySolver : elag_hip_smoother_s {
    elag_hip_hierarchy_s *H = nullptr;
    double scale = 1.0;
    void apply(const double *f, double *&cur, double *&, bool zero_guess, bool) override
    {
        // the V-cycle with its symmetric smoothers is its own transpose (Hierarchy::MultTranspose is not implemented in the
        // reference either: ParELAG_Hierarchy.hpp:125-129)
        ELAG_REQUIRE(scale == 1.0 || zero_guess, "a scaled hierarchy can only be applied as a preconditioner (iterative_mode == false)");
        hierarchy_apply_raw(H, f, cur, zero_guess);
        if (scale != 1.0) k::vec_axpby(n(), scale, cur, 0.0, cur, cur);
    }
    int flips(bool) const override { return 0; }
    bool graph_capturable() const override { return hierarchy_is_local(H); }
    double result_scale() const override { return scale; }
    int plain_forms() const override { return 1 | (H->adds_in_place() ? 2 : 0); }
    void apply_plain(const double *f, double *out, bool add) override
    {
        ELAG_REQUIRE(H->finalized, "Hierarchy::Mult before Finalize()");
        if (add) H->run_add(f, out);
        else H->run(f, out, true);
    }
};
int hierarchy_height(const elag_hip_hierarchy_s *H) { return (H->reordered ? H->A_user : H->lev[0].A)->nrows(); }
}  // namespace elag

extern "C" {

int elag_hip_hierarchy_create(elag_hip_hierarchy_t *out, int num_levels)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && num_levels > 0, "Hierarchy needs at least one level");
        auto *H = new elag_hip_hierarchy_s();
        H->lev.resize((size_t)num_levels);
        *out = H;
    });
}

static Level &level_of(elag_hip_hierarchy_t H, int l)
{
    ELAG_REQUIRE(H, "null hierarchy");
    ELAG_REQUIRE(l >= 0 && l < (int)H->lev.size(), "Hierarchy::GetLevel(): Level = %d is out of range (%d levels)", l, (int)H->lev.size());
    return H->lev[(size_t)l];
}

int elag_hip_hierarchy_set_operator(elag_hip_hierarchy_t H, int level, elag_hip_parcsr_t A)
{
    return guarded([&] { ELAG_REQUIRE(A, "null operator"); level_of(H, level).A = A; H->finalized = false; });
}

int elag_hip_hierarchy_set_prolongator(elag_hip_hierarchy_t H, int coarse_level, elag_hip_parcsr_t P)
{
    return guarded([&] {
        ELAG_REQUIRE(P, "null prolongator");
        ELAG_REQUIRE(coarse_level >= 1, "\"P\" lives on levels >= 1");
        level_of(H, coarse_level).P = P;
        H->finalized = false;
    });
}

int elag_hip_hierarchy_set_smoothers(elag_hip_hierarchy_t H, int level, elag_hip_smoother_t pre, elag_hip_smoother_t post)
{
    return guarded([&] {
        Level &L = level_of(H, level);
        L.pre = pre; L.post = post;
        H->finalized = false;
    });
}

int elag_hip_hierarchy_set_coarse_solver(elag_hip_hierarchy_t H, elag_hip_coarse_t C)
{
    return guarded([&] { ELAG_REQUIRE(H, "null hierarchy"); H->coarse = C; H->finalized = false; });
}

int elag_hip_hierarchy_finalize(elag_hip_hierarchy_t H, int use_graph)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(H, "null hierarchy");
        for (size_t l = 0; l < H->lev.size(); l++) {
            Level &L = H->lev[l];
            if (L.pre) ELAG_REQUIRE(L.A && L.pre->n() == L.A->nrows(), "PreSmoother size mismatch on level %d", (int)l);
            if (L.post) ELAG_REQUIRE(L.A && L.post->n() == L.A->nrows(), "PostSmoother size mismatch on level %d", (int)l);
        }
        if (H->coarse)
            ELAG_REQUIRE(H->lev.back().A && (H->coarse->replicated ? H->coarse->nloc : H->coarse->n) == H->lev.back().A->nrows(),
                         "CoarseSolver size mismatch");
        H->finalize(use_graph != 0);
    });
}

int elag_hip_hierarchy_apply(elag_hip_hierarchy_t H, const double *rhs, double *sol, int is_preconditioner, int loc)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(H && rhs && sol, "null argument");
        ELAG_REQUIRE(H->finalized, "Hierarchy::Mult before Finalize()");
        ELAG_REQUIRE(rhs != sol, "rhs and sol must not alias");
        check_loc(loc);
        const size_t n = (size_t)H->lev[0].A->nrows();
        if (loc == ELAG_HIP_DEVICE) {
            H->apply_device(rhs, sol, is_preconditioner != 0);
        } else {
            // host vectors are staged through the level-0 scratch (stable addresses keep the graph cache warm)
            hipStream_t st = ctx().stream;
            double *drhs = H->lev[0].resid, *dsol = H->lev[0].sol;
            stage_h2d(drhs, rhs, n * sizeof(double));
            if (!is_preconditioner) stage_h2d(dsol, sol, n * sizeof(double));
            H->apply_device(drhs, dsol, is_preconditioner != 0);
            stage_d2h(sol, dsol, n * sizeof(double));
            ELAG_HIP_CHECK(hipStreamSynchronize(st));
        }
    });
}

int elag_hip_coarse_create_from_hierarchy(elag_hip_coarse_t *out, elag_hip_hierarchy_t sub, int allreduce_rhs)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && sub, "null argument");
        ELAG_REQUIRE(sub->finalized, "the sub-hierarchy must be finalised first");
        std::unique_ptr<elag_hip_coarse_s> C(new elag_hip_coarse_s());
        C->A = sub->reordered ? sub->A_user : sub->lev[0].A;
        C->n = C->A->nrows();
        C->kind = 2;
        const int n = C->n;
        double *buf = dev_alloc<double>((size_t)n);   // owned through full_rhs
        C->full_rhs = buf;
        C->sub = sub; C->sub_allreduce = allreduce_rhs;
        C->custom_apply = [sub, buf, n, allreduce_rhs](const double *rhs, double *sol) {
            Context &c = ctx();
            const double *src = rhs;
            if (allreduce_rhs && comm_active()) {
                // sum the ranks' partial restricted residuals: one collective on the comm stream, fork/join by events
                dev_copy(buf, rhs, sizeof(double) * (size_t)n, c.stream);
                ELAG_HIP_CHECK(hipEventRecord(c.ev_pack, c.stream));
                ELAG_HIP_CHECK(hipStreamWaitEvent(c.comm_stream, c.ev_pack, 0));
                comm_allreduce_device(buf, n, 0, c.comm_stream);
                ELAG_HIP_CHECK(hipEventRecord(c.ev_halo, c.comm_stream));
                ELAG_HIP_CHECK(hipStreamWaitEvent(c.stream, c.ev_halo, 0));
                src = buf;
            }
            sub->apply_device(src, sol, true);
        };
        *out = C.release();
    });
}

int elag_hip_hierarchy_destroy(elag_hip_hierarchy_t H)
{
    return guarded([&] {
        if (H) { (void)hipStreamSynchronize(ctx().stream); delete H; }
    });
}

int elag_hip_amge_default_options(elag_hip_amge_options *opt)
{
    return guarded([&] {
        ELAG_REQUIRE(opt, "null options");
        opt->fix_zero_rows = 1; opt->relax_type = ELAG_HIP_SMOOTHER_L1_JACOBI; opt->sweeps = 1; opt->damping = 1.0;
        opt->cheby_order = 2; opt->cheby_fraction = 0.3; opt->coarse_kind = 0; opt->coarse_iters = 0; opt->use_graph = 1;
        opt->reorder = -1;
    });
}

}  // extern "C"

namespace {

// Cuthill-McKee order (new -> old) of a symmetric pattern, components in index order
std::vector<int> cm_order(int n, const std::vector<int> &I, const std::vector<int> &J)
{
    std::vector<int> order; order.reserve((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    std::vector<int> nb;
    for (int start = 0; start < n; start++) {
        if (seen[(size_t)start]) continue;
        // lowest-degree unvisited vertex of this component would be the textbook choice; the first one is enough here
        seen[(size_t)start] = 1;
        size_t head = order.size();
        order.push_back(start);
        while (head < order.size()) {
            const int v = order[head++];
            nb.clear();
            for (int k = I[(size_t)v]; k < I[(size_t)v + 1]; k++) {
                const int w = J[(size_t)k];
                if (w >= 0 && w < n && !seen[(size_t)w]) { seen[(size_t)w] = 1; nb.push_back(w); }
            }
            std::sort(nb.begin(), nb.end(), [&](int x, int y) {
                const int dx = I[(size_t)x + 1] - I[(size_t)x], dy = I[(size_t)y + 1] - I[(size_t)y];
                return dx != dy ? dx < dy : x < y;
            });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    return order;
}

}  // namespace

extern "C" {

static void build_impl(elag_hip_hierarchy_t *out, elag_hip_parcsr_t A, int num_P, const elag_hip_parcsr_t *P,
                       const elag_hip_parcsr_t *D, const elag_hip_amge_options *opt_in, const elag_hip_amge_options *aux_in)
{
    {
        require_init();
        ELAG_REQUIRE(out && A && num_P >= 0 && (num_P == 0 || P), "bad argument");
        elag_hip_amge_options opt, aux;
        if (opt_in) opt = *opt_in; else elag_hip_amge_default_options(&opt);
        if (aux_in) aux = *aux_in; else elag_hip_amge_default_options(&aux);
        std::unique_ptr<elag_hip_hierarchy_s> H(new elag_hip_hierarchy_s());
        AllocScope alloc_scope(!comm_active() && !is_distributed(*A));
        H->lev.resize((size_t)num_P + 1);
        H->lev[0].A = A;
        H->phase("galerkin_products", [&] {
        for (int l = 0; l < num_P; l++) {
            ParCSR *Af = H->lev[(size_t)l].A;
            ParCSR *Pl = P[l];
            ELAG_REQUIRE(Pl, "null P on level %d", l);
            // "Number of rows in P does not match A" check of Hierarchy.cpp:344-352
            ELAG_REQUIRE(Pl->nrows() == Af->nrows(), "buildHierarchy: P[%d] has %d rows but A has %d", l, Pl->nrows(), Af->nrows());
            elag_hip_parcsr_t Ac = nullptr;
            const bool diag_relax = opt.relax_type == ELAG_HIP_SMOOTHER_L1_JACOBI || opt.relax_type == ELAG_HIP_SMOOTHER_JACOBI ||
                                    opt.relax_type == ELAG_HIP_SMOOTHER_LUMPED_JACOBI || opt.relax_type == ELAG_HIP_SMOOTHER_HYPRE_JACOBI;
            // (the same size gate as plan_fused's bandwidth-bound branch, which is the one consumer of AP_kept: ELAG_HIP_FUSE_NNZ, read at every build)
            const char *e_nnz = getenv("ELAG_HIP_FUSE_NNZ");
            const long long keep_nnz = e_nnz ? atoll(e_nnz) : (1ll << 20);
            if (!D && diag_relax && !is_distributed(*Af) && !is_distributed(*Pl) && Af->diag.nnz >= keep_nnz) {
                // the product of elag_hip_parcsr_rap, with A P kept for plan_fused (a bandwidth-bound level goes up through S = A P - W^-1 P)
                ELAG_REQUIRE(Af->ncols() == Pl->nrows(), "rap: shapes do not chain");
                std::shared_ptr<DevCSR> AP(new DevCSR());
                DevCSR prod;
                k::csr_spgemm(Af->diag, Pl->diag, *AP, false);
                Pl->ensure_transpose();
                k::csr_spgemm(*Pl->diagT, *AP, prod);
                Ac = parcsr_from_diag(std::move(prod), Pl->first_col, Pl->first_col, Pl->global_cols, Pl->global_cols);
                H->lev[(size_t)l].AP_kept = AP;
            } else {
                int rc = elag_hip_parcsr_rap(Pl, Af, Pl, &Ac);
                if (rc) fail(rc, "%s", elag_hip_last_error());
            }
            H->owned_mats.push_back(Ac);
            if (opt.fix_zero_rows) k::fix_zero_rows(Ac->diag, &Ac->offd, false);
            H->lev[(size_t)l + 1].A = Ac;
            H->lev[(size_t)l + 1].P = Pl;
        }
        });

        // ---- optional internal renumbering for gather locality -------------------------------------
        // (a Gauss-Seidel sweep is defined by the caller's row order: those hierarchies keep it)
        const bool ordered_relax = opt.relax_type == ELAG_HIP_SMOOTHER_L1_GS || opt.relax_type == ELAG_HIP_SMOOTHER_L1_GS_TRUNCATED ||
                                   opt.relax_type == ELAG_HIP_SMOOTHER_GS || opt.relax_type == ELAG_HIP_SMOOTHER_KACZMARZ;
        ELAG_REQUIRE(!(ordered_relax && opt.reorder > 0), "reorder = 1 would change what a Gauss-Seidel sweep computes");
        const bool reorder = !A->needs_exchange() && A->global_rows == A->nrows() && D == nullptr && !ordered_relax &&
                             (opt.reorder > 0 || (opt.reorder < 0 && A->nrows() >= (1 << 18) && expected_cycles() >= 64.0));
        // (the renumbering is an investment too -- 2.6 ms at cube456 x5 for ~50 us a cycle: left out in the automatic mode when the caller expects fewer than 64 cycles)
        if (reorder) H->phase("renumbering", [&] {
            const int nl = num_P + 1;
            H->order.resize((size_t)nl);
            {
                ParCSR *Ac = H->lev[(size_t)nl - 1].A;
                std::vector<int> I = host_from_dev(Ac->diag.I, (size_t)Ac->nrows() + 1);
                std::vector<int> J = host_from_dev(Ac->diag.J, (size_t)Ac->diag.nnz);
                H->order[(size_t)nl - 1] = cm_order(Ac->nrows(), I, J);
            }
            // children under their strongest coarse parent, level by level from the coarsest up -- on the device (round 2 pulled every P to
            // the host for this: 76 MB and 27 ms of a 70 ms setup at cube456 x5); the host copies of the orders are made when asked for
            std::vector<int *> d_order((size_t)nl), d_rank((size_t)nl);
            d_order[(size_t)nl - 1] = dev_from_host(H->order[(size_t)nl - 1].data(), H->order[(size_t)nl - 1].size());
            for (int l = nl - 1; l >= 0; l--) {
                const int n = H->lev[(size_t)l].A->nrows();
                d_rank[(size_t)l] = dev_alloc<int>((size_t)n);
                k::inverse_permutation(n, d_order[(size_t)l], d_rank[(size_t)l]);
                if (l > 0) {
                    d_order[(size_t)l - 1] = dev_alloc<int>((size_t)H->lev[(size_t)l - 1].A->nrows());
                    k::child_order(H->lev[(size_t)l].P->diag, d_rank[(size_t)l], d_order[(size_t)l - 1]);
                }
            }
            ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
            std::vector<ParCSR *> newA((size_t)nl), newP((size_t)nl, nullptr);
            for (int l = 0; l < nl; l++) {
                ParCSR *Al = H->lev[(size_t)l].A;
                DevCSR Ap;
                k::csr_permute(Al->diag, d_order[(size_t)l], d_rank[(size_t)l], Al->ncols(), Ap);
                newA[(size_t)l] = parcsr_from_diag(std::move(Ap), 0, 0, Al->global_rows, Al->global_cols);
                if (l > 0) {
                    ParCSR *Pl = H->lev[(size_t)l].P;
                    DevCSR Pp;
                    k::csr_permute(Pl->diag, d_order[(size_t)l - 1], d_rank[(size_t)l], Pl->ncols(), Pp);
                    newP[(size_t)l] = parcsr_from_diag(std::move(Pp), 0, 0, Pl->global_rows, Pl->global_cols);
                    if (H->lev[(size_t)l - 1].AP_kept) {
                        std::shared_ptr<DevCSR> APp(new DevCSR());
                        k::csr_permute(*H->lev[(size_t)l - 1].AP_kept, d_order[(size_t)l - 1], d_rank[(size_t)l], Pl->ncols(), *APp, false);   // (only ever a SpMV operand)
                        H->lev[(size_t)l - 1].AP_kept = APp;
                    }
                }
            }
            ELAG_HIP_CHECK(hipStreamSynchronize(ctx().stream));
            for (auto *m : H->owned_mats) delete m;   // the natural-order coarse operators
            H->owned_mats.clear();
            for (int l = 0; l < nl; l++) {
                H->lev[(size_t)l].A = newA[(size_t)l];
                H->owned_mats.push_back(newA[(size_t)l]);
                if (l > 0) { H->lev[(size_t)l].P = newP[(size_t)l]; H->owned_mats.push_back(newP[(size_t)l]); }
            }
            H->d_order0 = d_order[0];
            H->d_order = d_order;                       // (kept: elag_hip_hierarchy_get_ordering downloads a level's order on request)
            for (int l = 0; l < nl; l++) dev_free(d_rank[(size_t)l]);
            H->A_user = A;
            H->reordered = true;
        });

        // (the coarsest operator's inverse first: it runs on the second stream beside the smoothers' set-up and the fused level operators)
        H->phase("coarse_solver", [&] {
        elag_hip_coarse_t C = nullptr;
        int rc;
        if (is_distributed(*H->lev.back().A)) {
            // coarsest operator gathered onto every rank, right-hand side all-gathered at apply time
            std::vector<int64_t> starts;
            ParCSR *Arep = replicate_matrix(*H->lev.back().A, &starts);
            H->owned_mats.push_back(Arep);
            rc = elag_hip_coarse_create_replicated(&C, Arep, starts.data(), opt.coarse_kind);
        } else {
            C = coarse_create_beside(H->lev.back().A, opt.coarse_kind);
            rc = 0;
        }
        if (rc) fail(rc, "%s", elag_hip_last_error());
        H->owned_coarse = C;
        H->coarse = C;
        });
        H->phase("smoothers", [&] {
        for (int l = 0; l < num_P; l++) {
            elag_hip_smoother_t S = nullptr;
            int rc = elag_hip_smoother_create(&S, H->lev[(size_t)l].A, opt.relax_type, opt.sweeps, opt.damping, 1.0, opt.cheby_order,
                                              opt.cheby_fraction);
            if (rc) fail(rc, "%s", elag_hip_last_error());
            H->owned_smoothers.push_back(S);
            if (D) {
                // HiptmairSmootherFactory::_do_build_solver: A_aux = RAP(A, D) + FixZeroRows (:143-165), auxiliary smoother
                // with iterative_mode = false (:134), primary with iterative_mode = true (:113)
                ParCSR *Dl = D[l];
                ELAG_REQUIRE(Dl, "null D on level %d", l);
                ELAG_REQUIRE(H->lev[(size_t)l].A->ncols() == Dl->nrows(),
                             "A and D do not have compatible sizes to compute D^T*A*D on level %d (A %dx%d, D %dx%d)", l,
                             H->lev[(size_t)l].A->nrows(), H->lev[(size_t)l].A->ncols(), Dl->nrows(), Dl->ncols());
                elag_hip_parcsr_t Aaux = nullptr;
                rc = elag_hip_parcsr_rap(Dl, H->lev[(size_t)l].A, Dl, &Aaux);
                if (rc) fail(rc, "%s", elag_hip_last_error());
                H->owned_mats.push_back(Aaux);
                k::fix_zero_rows(Aaux->diag, &Aaux->offd, false);
                elag_hip_smoother_t Sa = nullptr, Sh = nullptr;
                rc = elag_hip_smoother_create(&Sa, Aaux, aux.relax_type, aux.sweeps, aux.damping, 1.0, aux.cheby_order, aux.cheby_fraction);
                if (rc) fail(rc, "%s", elag_hip_last_error());
                H->owned_smoothers.push_back(Sa);
                rc = elag_hip_hiptmair_create(&Sh, H->lev[(size_t)l].A, Dl, S, Sa);
                if (rc) fail(rc, "%s", elag_hip_last_error());
                H->owned_smoothers.push_back(Sh);
                S = Sh;
            }
            H->lev[(size_t)l].pre = S;
            H->lev[(size_t)l].post = S;  // same parameters => same operator; scratch is not shared across calls
        }
        });
        H->finalize(opt.use_graph != 0);
        *out = H.release();
    }
}

int elag_hip_hierarchy_build(elag_hip_hierarchy_t *out, elag_hip_parcsr_t A, int num_P, const elag_hip_parcsr_t *P,
                             const elag_hip_amge_options *opt)
{
    return guarded([&] { build_impl(out, A, num_P, P, nullptr, opt, nullptr); });
}

int elag_hip_hierarchy_build_hiptmair(elag_hip_hierarchy_t *out, elag_hip_parcsr_t A, int num_P, const elag_hip_parcsr_t *P,
                                      const elag_hip_parcsr_t *D, const elag_hip_amge_options *opt, const elag_hip_amge_options *aux_opt)
{
    return guarded([&] {
        ELAG_REQUIRE(D != nullptr || num_P == 0, "null D array");
        build_impl(out, A, num_P, P, D, opt, aux_opt);
    });
}

int elag_hip_hierarchy_get_ordering(elag_hip_hierarchy_t H, int level, int *order)
{
    return guarded([&] {
        Level &L = level_of(H, level);
        ELAG_REQUIRE(order && L.A, "bad argument");
        const int n = L.A->nrows();
        if (H->reordered) {
            std::vector<int> &ho = H->order[(size_t)level];
            if ((int)ho.size() != n) ho = host_from_dev((const int *)H->d_order[(size_t)level], (size_t)n);
            memcpy(order, ho.data(), sizeof(int) * (size_t)n);
        } else {
            for (int i = 0; i < n; i++) order[i] = i;
        }
    });
}

int elag_hip_hierarchy_get_operator(elag_hip_hierarchy_t H, int level, elag_hip_parcsr_t *A)
{
    return guarded([&] { ELAG_REQUIRE(A, "null output"); *A = level_of(H, level).A; });
}

int elag_hip_smoother_from_hierarchy(elag_hip_smoother_t *out, elag_hip_hierarchy_t H, double scale)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(out && H, "null argument");
        ELAG_REQUIRE(H->finalized, "the hierarchy must be finalized before it is used as a solver");
        auto *S = new elag::HierarchySolver();
        S->H = H;
        S->scale = scale;
        S->A = H->reordered ? H->A_user : H->lev[0].A;
        *out = S;
    });
}

int elag_hip_hierarchy_get_smoother(elag_hip_hierarchy_t H, int level, elag_hip_smoother_t *S)
{
    return guarded([&] { ELAG_REQUIRE(S, "null output"); *S = level_of(H, level).pre; });
}

int elag_hip_time_kernel(elag_hip_parcsr_t A, elag_hip_smoother_t S, int kernel, int reps, const double *x_dev,
                         double *y_dev, double *ms_per_launch)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(A && x_dev && y_dev && ms_per_launch && reps > 0, "bad argument");
        ELAG_REQUIRE(kernel >= 0 && kernel <= 2, "unknown kernel id %d", kernel);
        const RelaxSmoother *R = nullptr;
        if (kernel == 2) {
            R = dynamic_cast<RelaxSmoother *>(S);
            ELAG_REQUIRE(R && R->dinv, "kernel 2 needs a (l1-)Jacobi smoother");
        }
        hipStream_t st = ctx().stream;
        hipEvent_t e0, e1;
        ELAG_HIP_CHECK(hipEventCreate(&e0));
        ELAG_HIP_CHECK(hipEventCreate(&e1));
        Temp<double> f((size_t)A->nrows());
        k::vec_fill(A->nrows(), 1.0, f);
        auto launch = [&] {
            k::SpmvArgs a;
            a.x = x_dev; a.out = y_dev;
            if (kernel == 0) { a.alpha = 1.0; }
            else if (kernel == 1) { a.alpha = -1.0; a.f = f; a.gamma = 1.0; }
            else { a.alpha = -1.0; a.f = f; a.gamma = 1.0; a.s = R->dinv; a.yin = x_dev; a.beta = 1.0; }
            par_spmv(*A, a);
        };
        launch();  // warm
        ELAG_HIP_CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) launch();
        ELAG_HIP_CHECK(hipEventRecord(e1, st));
        event_wait(e1);
        float ms = 0.f;
        ELAG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_launch = (double)ms / reps;
    });
}

int elag_hip_time_hierarchy(elag_hip_hierarchy_t H, const double *rhs_dev, double *sol_dev, int reps, double *ms_per_apply)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(H && rhs_dev && sol_dev && ms_per_apply && reps > 0, "bad argument");
        ELAG_REQUIRE(H->finalized, "hierarchy not finalised");
        hipStream_t st = ctx().stream;
        hipEvent_t e0, e1;
        ELAG_HIP_CHECK(hipEventCreate(&e0));
        ELAG_HIP_CHECK(hipEventCreate(&e1));
        H->apply_device(rhs_dev, sol_dev, true);
        // timed in chunks with a drain in between, so that at most 64 cycles are ever queued ahead of the device
        double total_ms = 0.0;
        for (int done = 0; done < reps;) {
            const int chunk = std::min(64, reps - done);
            ELAG_HIP_CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < chunk; i++) H->apply_device(rhs_dev, sol_dev, true);
            ELAG_HIP_CHECK(hipEventRecord(e1, st));
            event_wait(e1);
            float ms = 0.f;
            ELAG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            done += chunk;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *ms_per_apply = total_ms / reps;
    });
}

int elag_hip_hierarchy_launch_mode(elag_hip_hierarchy_t H, int *uses_graph, int *kernels, double *graph_us, double *launches_us)
{
    return guarded([&] {
        ELAG_REQUIRE(H, "null hierarchy");
        ELAG_REQUIRE(H->finalized, "hierarchy not finalised");
        if (uses_graph) *uses_graph = H->use_graph ? 1 : 0;
        if (kernels) *kernels = H->tune_nodes;
        if (graph_us) *graph_us = H->tune_graph_us;
        if (launches_us) *launches_us = H->tune_eager_us;
    });
}

int elag_hip_hierarchy_setup_breakdown(elag_hip_hierarchy_t H, char *buf, int len)
{
    return guarded([&] {
        ELAG_REQUIRE(H && buf && len > 0, "bad argument");
        std::string out;
        char tmp[96];
        for (auto &p : H->setup_ms) {
            snprintf(tmp, sizeof tmp, "%s%s=%.4f", out.empty() ? "" : ";", p.first, p.second);
            out += tmp;
        }
        ELAG_REQUIRE((int)out.size() < len, "buffer of %d bytes is too small for %d", len, (int)out.size() + 1);
        memcpy(buf, out.c_str(), out.size() + 1);
    });
}

int elag_hip_time_hierarchy_internal(elag_hip_hierarchy_t H, int reps, double *ms_per_apply)
{
    return guarded([&] {
        require_init();
        ELAG_REQUIRE(H && ms_per_apply && reps > 0, "bad argument");
        ELAG_REQUIRE(H->finalized, "hierarchy not finalised");
        const int n = H->lev[0].A->nrows();
        hipStream_t st = ctx().stream;
        // the cycle as a solver living in the hierarchy's numbering runs it: no permutation on the way in or out
        Temp<double> rhs((size_t)n), sol((size_t)n);
        k::vec_fill(n, 2.0, rhs);
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (H->use_graph) {
            ELAG_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            try { H->iterate(0, rhs, sol, H->lev[0].alt); } catch (...) { (void)hipStreamEndCapture(st, &graph); if (graph) (void)hipGraphDestroy(graph); throw; }
            ELAG_HIP_CHECK(hipStreamEndCapture(st, &graph));
            ELAG_HIP_CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
        }
        auto once = [&] { if (exec) ELAG_HIP_CHECK(hipGraphLaunch(exec, st)); else H->iterate(0, rhs, sol, H->lev[0].alt); };
        hipEvent_t e0, e1;
        ELAG_HIP_CHECK(hipEventCreate(&e0));
        ELAG_HIP_CHECK(hipEventCreate(&e1));
        once();
        double total_ms = 0.0;
        for (int done = 0; done < reps;) {
            const int chunk = std::min(64, reps - done);
            ELAG_HIP_CHECK(hipEventRecord(e0, st));
            for (int i = 0; i < chunk; i++) once();
            ELAG_HIP_CHECK(hipEventRecord(e1, st));
            event_wait(e1);
            float ms = 0.f;
            ELAG_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            done += chunk;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        if (exec) (void)hipGraphExecDestroy(exec);
        *ms_per_apply = total_ms / reps;
    });
}

}  // extern "C"
