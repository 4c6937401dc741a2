// Header-only C++ mirror of the reference's operator interface for the V-cycle path, over the C ABI of
// include/elag_hip.h.  Class and method names follow ParElag so that a maintainer can drop these behind
// mfem::Operator:
//
//   elag_hip::ParallelCSRMatrix     <- mfem::HypreParMatrix as ParElag uses it (Mult / MultTranspose)
//   elag_hip::HypreSmootherWrapper  <- src/linalg/solver_ops/ParELAG_HypreSmootherWrapper.{hpp,cpp}
//   elag_hip::HiptmairSmoother      <- src/linalg/solver_ops/ParELAG_HiptmairSmoother.{hpp,cpp}
//   elag_hip::Hierarchy             <- src/linalg/solver_ops/ParELAG_Hierarchy.{hpp,cpp}
//
// Error behaviour matches the reference: failures throw std::runtime_error (the reference throws through
// PARELAG_TEST_FOR_EXCEPTION / PARELAG_ASSERT_HYPRE_ERROR_FLAG, src/utilities/elagError.hpp:83-135).
// Vectors are raw `double*` + a location flag; mfem::Vector::GetData() is a host pointer (ELAG_HIP_HOST).
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/elag_hip.h"

namespace elag_hip {

inline void check(int status)
{
    if (status != ELAG_HIP_OK) throw std::runtime_error(std::string("elag_hip: ") + elag_hip_last_error());
}

inline void Init(int device = 0) { check(elag_hip_init(device)); }
inline void Synchronize() { check(elag_hip_synchronize()); }
inline size_t TrimMemory() { size_t n = 0; check(elag_hip_trim_memory(&n)); return n; }
// hybrid Gauss-Seidel of the smoothers created from here on as the reference runs it on `nblocks` ranks (1: the sequential loop)
inline void SetGaussSeidelHybridBlocks(int nblocks) { check(elag_hip_set_gs_hybrid_blocks(nblocks)); }

// Minimal solver base with mfem::Solver's iterative_mode / parelag::Solver::IsPreconditioner
// (src/linalg/solver_core/ParELAG_Solver.hpp:57-103).
class Solver {
public:
    bool iterative_mode = true;
    bool IsPreconditioner() const { return !iterative_mode; }
    virtual ~Solver() = default;
    virtual void Mult(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const = 0;
    virtual void MultTranspose(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const { Mult(rhs, sol, loc); }
    virtual int Height() const = 0;
};

class ParallelCSRMatrix {
public:
    // Single-rank matrix from host CSR arrays (copied to the device).
    ParallelCSRMatrix(int nrows, int ncols, const int *I, const int *J, const double *a)
    {
        elag_hip_parcsr_desc d{};
        d.num_rows = nrows; d.num_cols_diag = ncols; d.diag_i = I; d.diag_j = J; d.diag_a = a;
        d.global_num_rows = nrows; d.global_num_cols = ncols;
        check(elag_hip_parcsr_create(&h_, &d));
    }
    // One rank's rows of a distributed matrix: the fields of hypre_ParCSRMatrix + hypre_ParCSRCommPkg.
    explicit ParallelCSRMatrix(const elag_hip_parcsr_desc &d) { check(elag_hip_parcsr_create(&h_, &d)); }
    // Adopt a handle produced by the library (RAP, Transpose, ...).
    explicit ParallelCSRMatrix(elag_hip_parcsr_t h, bool owner = true) : h_(h), owner_(owner) {}
    ~ParallelCSRMatrix() { if (owner_ && h_) elag_hip_parcsr_destroy(h_); }
    ParallelCSRMatrix(const ParallelCSRMatrix &) = delete;
    ParallelCSRMatrix &operator=(const ParallelCSRMatrix &) = delete;

    elag_hip_parcsr_t Handle() const { return h_; }
    elag_hip_parcsr_info Info() const { elag_hip_parcsr_info i; check(elag_hip_parcsr_get_info(h_, &i)); return i; }
    int Height() const { return Info().num_rows; }
    int Width() const { return Info().num_cols_diag; }

    // y = a*A*x + b*y   (HypreParMatrix::Mult(a, x, b, y))
    void Mult(const double *x, double *y, double a = 1.0, double b = 0.0, int loc = ELAG_HIP_HOST) const
    { check(elag_hip_parcsr_spmv(h_, 0, a, x, b, y, loc)); }
    void MultTranspose(const double *x, double *y, double a = 1.0, double b = 0.0, int loc = ELAG_HIP_HOST) const
    { check(elag_hip_parcsr_spmv(h_, 1, a, x, b, y, loc)); }
    // r = b - A x   (mg_utils::ComputeResidual, ParELAG_MG_Utils.hpp:405-426)
    void ComputeResidual(const double *b, const double *x, double *r, int loc = ELAG_HIP_HOST) const
    { check(elag_hip_parcsr_residual(h_, b, x, r, loc)); }

    std::unique_ptr<ParallelCSRMatrix> Transpose() const
    { elag_hip_parcsr_t t; check(elag_hip_parcsr_transpose(h_, &t)); return std::make_unique<ParallelCSRMatrix>(t); }
    int FixZeroRows() { int n = 0; check(elag_hip_parcsr_fix_zero_rows(h_, &n)); return n; }  // hypre_ParCSRMatrixFixZeroRows
    void DeleteZeros(double tol) { check(elag_hip_parcsr_delete_zeros(h_, tol)); }            // hypre_ParCSRMatrixDeleteZeros
    double Norml1() const { double v; check(elag_hip_parcsr_norm(h_, 0, &v)); return v; }      // hypre_ParCSRMatrixNorml1
    double Normlinf() const { double v; check(elag_hip_parcsr_norm(h_, 1, &v)); return v; }
    double MaxNorm() const { double v; check(elag_hip_parcsr_norm(h_, 2, &v)); return v; }
    double FrobeniusNorm() const { double v; check(elag_hip_parcsr_norm(h_, 3, &v)); return v; }

private:
    elag_hip_parcsr_t h_ = nullptr;
    bool owner_ = true;
};

// mfem::RAP(A, P) = P^T A P and mfem::RAP(Rt, A, P) (call sites Hierarchy.cpp:365,511-513).
inline std::unique_ptr<ParallelCSRMatrix> RAP(const ParallelCSRMatrix &A, const ParallelCSRMatrix &P)
{ elag_hip_parcsr_t c; check(elag_hip_parcsr_rap(P.Handle(), A.Handle(), P.Handle(), &c)); return std::make_unique<ParallelCSRMatrix>(c); }
inline std::unique_ptr<ParallelCSRMatrix> RAP(const ParallelCSRMatrix &Rt, const ParallelCSRMatrix &A, const ParallelCSRMatrix &P)
{ elag_hip_parcsr_t c; check(elag_hip_parcsr_rap(Rt.Handle(), A.Handle(), P.Handle(), &c)); return std::make_unique<ParallelCSRMatrix>(c); }
// hypre_ParMatmul / mfem::ParMult
inline std::unique_ptr<ParallelCSRMatrix> ParMult(const ParallelCSRMatrix &A, const ParallelCSRMatrix &B)
{ elag_hip_parcsr_t c; check(elag_hip_parcsr_matmul(A.Handle(), B.Handle(), &c)); return std::make_unique<ParallelCSRMatrix>(c); }
// hypre_ParCSRMatrixAdd2 (src/hypreExtension/parcsr-add.c:199-245)
inline std::unique_ptr<ParallelCSRMatrix> Add(double a, const ParallelCSRMatrix &A, double b, const ParallelCSRMatrix &B)
{ elag_hip_parcsr_t c; check(elag_hip_parcsr_add(a, A.Handle(), b, B.Handle(), &c)); return std::make_unique<ParallelCSRMatrix>(c); }
// hypre_ParCSRMatrixCompare (bit mask)
inline int Compare(const ParallelCSRMatrix &A, const ParallelCSRMatrix &B, double tol)
{ int m = 0; check(elag_hip_parcsr_compare(A.Handle(), B.Handle(), tol, &m)); return m; }

// The "Sweeps" / "Damping Factor" / "Omega" / "Cheby Poly Order" / "Cheby Poly Fraction" entries of the smoother's
// ParameterList (ParELAG_HypreSmootherWrapper.cpp:28-33), same defaults.
struct SmootherParameters {
    int sweeps = 1;
    double damping_factor = 1.0, omega = 1.0;
    int cheby_poly_order = 2;
    double cheby_poly_fraction = 0.3;
};

class HypreSmootherWrapper : public Solver {
public:
    HypreSmootherWrapper(const std::shared_ptr<ParallelCSRMatrix> &A, int type, const SmootherParameters &p = {}) : A_(A)
    {
        check(elag_hip_smoother_create(&h_, A->Handle(), type, p.sweeps, p.damping_factor, p.omega, p.cheby_poly_order,
                                       p.cheby_poly_fraction));
        iterative_mode = true;
    }
    ~HypreSmootherWrapper() override { if (h_) elag_hip_smoother_destroy(h_); }
    void Mult(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const override
    { check(elag_hip_smoother_apply(h_, rhs, sol, iterative_mode ? 0 : 1, 0, loc)); }
    void MultTranspose(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const override
    { check(elag_hip_smoother_apply(h_, rhs, sol, iterative_mode ? 0 : 1, 1, loc)); }
    int Height() const override { return A_->Height(); }
    elag_hip_smoother_t Handle() const { return h_; }

protected:
    HypreSmootherWrapper() = default;
    std::shared_ptr<ParallelCSRMatrix> A_;
    elag_hip_smoother_t h_ = nullptr;
};

class HiptmairSmoother : public HypreSmootherWrapper {
public:
    // HiptmairSmoother(A, A_aux, D, primary, auxiliary): A_aux is owned by `auxiliary` (ParELAG_HiptmairSmoother.cpp:22-46)
    HiptmairSmoother(const std::shared_ptr<ParallelCSRMatrix> &A, const std::shared_ptr<ParallelCSRMatrix> &D,
                     const std::shared_ptr<HypreSmootherWrapper> &primary, const std::shared_ptr<HypreSmootherWrapper> &auxiliary)
        : D_(D), primary_(primary), aux_(auxiliary)
    {
        A_ = A;
        check(elag_hip_hiptmair_create(&h_, A->Handle(), D->Handle(), primary->Handle(), auxiliary->Handle()));
        iterative_mode = true;
    }

private:
    std::shared_ptr<ParallelCSRMatrix> D_;
    std::shared_ptr<HypreSmootherWrapper> primary_, aux_;
};

class CoarseSolver : public Solver {
public:
    explicit CoarseSolver(const std::shared_ptr<ParallelCSRMatrix> &A, int kind = 0) : A_(A)
    { check(elag_hip_coarse_create(&h_, A->Handle(), kind, 0)); iterative_mode = false; }
    ~CoarseSolver() override { if (h_) elag_hip_coarse_destroy(h_); }
    void Mult(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const override { check(elag_hip_coarse_apply(h_, rhs, sol, loc)); }
    int Height() const override { return A_->Height(); }
    elag_hip_coarse_t Handle() const { return h_; }

private:
    std::shared_ptr<ParallelCSRMatrix> A_;
    elag_hip_coarse_t h_ = nullptr;
};

class Hierarchy : public Solver {
public:
    explicit Hierarchy(int num_levels) { check(elag_hip_hierarchy_create(&h_, num_levels)); iterative_mode = false; }
    explicit Hierarchy(elag_hip_hierarchy_t h) : h_(h) { iterative_mode = false; }
    ~Hierarchy() override { if (h_) elag_hip_hierarchy_destroy(h_); }

    // Level::Set("A"/"P"/"PreSmoother"/"PostSmoother"/"CoarseSolver", ...) of the reference; operators are shared.
    void SetOperator(int level, const std::shared_ptr<ParallelCSRMatrix> &A)
    { keep_.push_back(A); check(elag_hip_hierarchy_set_operator(h_, level, A->Handle())); n0_ = level == 0 ? A->Height() : n0_; }
    void SetProlongator(int coarse_level, const std::shared_ptr<ParallelCSRMatrix> &P)
    { keep_.push_back(P); check(elag_hip_hierarchy_set_prolongator(h_, coarse_level, P->Handle())); }
    void SetSmoothers(int level, const std::shared_ptr<HypreSmootherWrapper> &pre, const std::shared_ptr<HypreSmootherWrapper> &post)
    { keep_.push_back(pre); keep_.push_back(post); check(elag_hip_hierarchy_set_smoothers(h_, level, pre ? pre->Handle() : nullptr, post ? post->Handle() : nullptr)); }
    void SetCoarseSolver(const std::shared_ptr<CoarseSolver> &c) { keep_.push_back(c); check(elag_hip_hierarchy_set_coarse_solver(h_, c->Handle())); }
    void Finalize(bool use_graph = true) { check(elag_hip_hierarchy_finalize(h_, use_graph ? 1 : 0)); }

    void Mult(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const override
    { check(elag_hip_hierarchy_apply(h_, rhs, sol, IsPreconditioner() ? 1 : 0, loc)); }
    int Height() const override { return n0_; }

    // buildHierarchyFromDeRhamSequence + AMGeSolverFactory::_do_build_solver (Hierarchy.cpp:282-383, AMGeSolverFactory.cpp:26-169)
    static std::unique_ptr<Hierarchy> Build(const std::shared_ptr<ParallelCSRMatrix> &A,
                                            const std::vector<std::shared_ptr<ParallelCSRMatrix>> &P,
                                            const elag_hip_amge_options *opt = nullptr)
    {
        std::vector<elag_hip_parcsr_t> ph;
        for (auto &p : P) ph.push_back(p->Handle());
        elag_hip_hierarchy_t h;
        check(elag_hip_hierarchy_build(&h, A->Handle(), (int)ph.size(), ph.data(), opt));
        auto H = std::make_unique<Hierarchy>(h);
        H->keep_.push_back(A);
        for (auto &p : P) H->keep_.push_back(p);
        H->n0_ = A->Height();
        return H;
    }

    void Keep(const std::shared_ptr<void> &p) { keep_.push_back(p); }
    void SetHeight(int n) { n0_ = n; }
    elag_hip_hierarchy_t Handle() const { return h_; }

private:
    elag_hip_hierarchy_t h_ = nullptr;
    std::vector<std::shared_ptr<void>> keep_;
    int n0_ = 0;
};

// A finalized hierarchy as an inner solver object: what a parameter list means by an AMGe solver in an "A00_1 Inverse" /
// "S Inverse" entry (example_parameterlists/spe10_example_parameters.xml:114-127).  scale = -1: a hierarchy built on
// B diag(M)^-1 B^T applied as the inverse of the saddle point's Schur complement.
class HierarchySolver : public HypreSmootherWrapper {
public:
    explicit HierarchySolver(const std::shared_ptr<Hierarchy> &H, double scale = 1.0) : H_(H)
    {
        check(elag_hip_smoother_from_hierarchy(&h_, H->Handle(), scale));
        iterative_mode = false;
    }
    int Height() const override { return H_->Height(); }

private:
    std::shared_ptr<Hierarchy> H_;
};

// parelag::KrylovSolver(A, Prec, params) (src/linalg/solver_ops/ParELAG_KrylovSolver.cpp:23-96): "Solver name" CG/PCG, GMRES,
// FGMRES, BiCGSTAB, MINRES with the iteration resident on the device.
class KrylovSolver : public Solver {
public:
    KrylovSolver(const std::shared_ptr<ParallelCSRMatrix> &A, const std::shared_ptr<Hierarchy> &prec, const elag_hip_krylov_options &opt)
        : A_(A), keep_(prec)
    { check(elag_hip_krylov_create(&h_, A->Handle(), ELAG_HIP_PREC_HIERARCHY, prec->Handle(), &opt)); iterative_mode = false; }
    KrylovSolver(const std::shared_ptr<ParallelCSRMatrix> &A, const std::shared_ptr<HypreSmootherWrapper> &prec, const elag_hip_krylov_options &opt)
        : A_(A), keep_(prec)
    { check(elag_hip_krylov_create(&h_, A->Handle(), prec ? ELAG_HIP_PREC_SMOOTHER : ELAG_HIP_PREC_NONE, prec ? prec->Handle() : nullptr, &opt)); iterative_mode = false; }
    ~KrylovSolver() override { if (h_) elag_hip_krylov_destroy(h_); }
    static elag_hip_krylov_options DefaultOptions() { elag_hip_krylov_options o; check(elag_hip_krylov_default_options(&o)); return o; }
    void Mult(const double *rhs, double *sol, int loc = ELAG_HIP_HOST) const override
    { check(elag_hip_krylov_mult(h_, rhs, sol, iterative_mode ? 1 : 0, loc)); }
    int Height() const override { return A_->Height(); }
    int GetNumIterations() const { int c, it; double nrm; check(elag_hip_krylov_get_stats(h_, &c, &it, &nrm)); return it; }
    bool GetConverged() const { int c, it; double nrm; check(elag_hip_krylov_get_stats(h_, &c, &it, &nrm)); return c != 0; }
    double GetFinalNorm() const { int c, it; double nrm; check(elag_hip_krylov_get_stats(h_, &c, &it, &nrm)); return nrm; }
    std::string LaunchMode() const { return elag_hip_krylov_launch_mode(h_); }   // "graph" / "plain" / "plain: <why>" (diagnostic)

private:
    std::shared_ptr<ParallelCSRMatrix> A_;
    std::shared_ptr<void> keep_;
    elag_hip_krylov_t h_ = nullptr;
};

// ---- blocked (saddle-point) layer: MfemBlockOperator, SchurComplementFactory, Block2x2 smoothers -----------------------
class MfemBlockOperator {
public:
    // MfemBlockOperator(row_offsets, col_offsets) (ParELAG_MfemBlockOperator.hpp:45-66); offsets are local dof offsets
    MfemBlockOperator(const std::vector<int> &row_offsets, const std::vector<int> &col_offsets)
        : row_offsets_(row_offsets), col_offsets_(col_offsets), ops_((row_offsets.size() - 1) * (col_offsets.size() - 1))
    { check(elag_hip_blockop_create(&h_, GetNumBlockRows(), GetNumBlockCols(), row_offsets_.data(), col_offsets_.data())); }
    explicit MfemBlockOperator(const std::vector<int> &offsets) : MfemBlockOperator(offsets, offsets) {}
    ~MfemBlockOperator() { if (h_) elag_hip_blockop_destroy(h_); }
    MfemBlockOperator(const MfemBlockOperator &) = delete;
    MfemBlockOperator &operator=(const MfemBlockOperator &) = delete;

    int GetNumBlockRows() const { return (int)row_offsets_.size() - 1; }
    int GetNumBlockCols() const { return (int)col_offsets_.size() - 1; }
    int Height() const { return row_offsets_.back(); }
    int Width() const { return col_offsets_.back(); }
    const std::vector<int> &ViewRowOffsets() const { return row_offsets_; }
    const std::vector<int> &ViewColumnOffsets() const { return col_offsets_; }
    void SetBlock(int i, int j, const std::shared_ptr<ParallelCSRMatrix> &op)
    { check(elag_hip_blockop_set_block(h_, i, j, op->Handle())); ops_[(size_t)i * GetNumBlockCols() + j] = op; }
    bool IsZeroBlock(int i, int j) const { elag_hip_parcsr_t a = nullptr; check(elag_hip_blockop_get_block(h_, i, j, &a)); return a == nullptr; }
    std::shared_ptr<ParallelCSRMatrix> GetBlockPtr(int i, int j) const { return ops_[(size_t)i * GetNumBlockCols() + j]; }
    void Mult(const double *x, double *y, int loc = ELAG_HIP_HOST) const { check(elag_hip_blockop_mult(h_, 0, 1.0, x, 0.0, y, loc)); }
    void MultTranspose(const double *x, double *y, int loc = ELAG_HIP_HOST) const { check(elag_hip_blockop_mult(h_, 1, 1.0, x, 0.0, y, loc)); }
    elag_hip_blockop_t Handle() const { return h_; }

private:
    std::vector<int> row_offsets_, col_offsets_;
    std::vector<std::shared_ptr<ParallelCSRMatrix>> ops_;
    elag_hip_blockop_t h_ = nullptr;
};

// SchurComplementFactory(type, alpha).BuildOperator(op, state) (ParELAG_SchurComplementFactory.cpp:36-181)
inline std::shared_ptr<ParallelCSRMatrix> SchurComplement(const MfemBlockOperator &op, int s_type, double alpha = 1.0, bool negate = false)
{ elag_hip_parcsr_t s; check(elag_hip_schur_complement(op.Handle(), s_type, alpha, negate ? 1 : 0, &s)); return std::make_shared<ParallelCSRMatrix>(s); }

// BlockDiagonalSolver / BlockTriangularSolver / Block2x2LDUInverseOperator (kind = ELAG_HIP_BLOCK_*)
class BlockSolver : public HypreSmootherWrapper {
public:
    BlockSolver(const std::shared_ptr<MfemBlockOperator> &op, const std::vector<std::shared_ptr<HypreSmootherWrapper>> &inv_ops, int kind,
                double damping = 1.0)
        : op_(op), inv_(inv_ops)
    {
        std::vector<elag_hip_smoother_t> hs;
        for (auto &s : inv_) hs.push_back(s->Handle());
        check(elag_hip_block_smoother_create(&h_, op->Handle(), kind, (int)hs.size(), hs.data(), damping));
        iterative_mode = true;   // Solver{..., IsPreconditioner = false} of the block solvers
    }
    int Height() const override { return op_->Height(); }

private:
    std::shared_ptr<MfemBlockOperator> op_;
    std::vector<std::shared_ptr<HypreSmootherWrapper>> inv_;
};

// buildBlockedHierarchyFromDeRhamSequence (Hierarchy.cpp:400-541) + block smoothers + dense LU coarse solve
inline std::unique_ptr<Hierarchy> BuildBlockedHierarchy(const std::shared_ptr<MfemBlockOperator> &A,
                                                        const std::vector<std::vector<std::shared_ptr<ParallelCSRMatrix>>> &P,
                                                        const elag_hip_blocked_options *opt = nullptr, const int *fix_zero_rows = nullptr)
{
    std::vector<elag_hip_parcsr_t> ph;
    for (auto &lvl : P) for (auto &p : lvl) ph.push_back(p->Handle());
    elag_hip_hierarchy_t h;
    check(elag_hip_hierarchy_build_blocked(&h, A->Handle(), (int)P.size(), ph.data(), fix_zero_rows, opt));
    auto H = std::make_unique<Hierarchy>(h);
    H->Keep(A);
    for (auto &lvl : P) for (auto &p : lvl) H->Keep(p);
    H->SetHeight(A->Height());
    return H;
}

// ---- the integer tables of the coarsening -------------------------------------------------------------------------------------
// parelag::TopologyTable (src/topology/TopologyTable.hpp): a ragged +-1 table; products keep the columns of a row in
// first-touch order like mfem::Mult.
struct TopologyTable {
    std::vector<int64_t> ptr{0}, idx;
    std::vector<double> sgn;                 // empty: a pattern (boolean) table
    int64_t Height() const { return (int64_t)ptr.size() - 1; }
    int64_t Width = 0;
};

// MultOrientation (TopologyTable.cpp:130-139) when both tables carry signs, MultBoolean / mfem::Mult on patterns otherwise
inline TopologyTable Mult(const TopologyTable &A, const TopologyTable &B)
{
    const bool sg = !A.sgn.empty() && !B.sgn.empty();
    int64_t cap = 0;
    for (int64_t k : A.idx) cap += B.ptr[(size_t)k + 1] - B.ptr[(size_t)k];
    TopologyTable C;
    C.Width = B.Width;
    C.ptr.assign((size_t)A.Height() + 1, 0);
    C.idx.resize((size_t)cap);
    if (sg) C.sgn.resize((size_t)cap);
    int64_t nnz = 0;
    check(elag_hip_host_table_product(A.Height(), A.ptr.data(), A.idx.data(), sg ? A.sgn.data() : nullptr, B.Height(), B.ptr.data(), B.idx.data(),
                                      sg ? B.sgn.data() : nullptr, B.Width, C.ptr.data(), C.idx.data(), sg ? C.sgn.data() : nullptr, &nnz));
    C.idx.resize((size_t)nnz);
    if (sg) C.sgn.resize((size_t)nnz);
    return C;
}

// the DenseMatrix::Rank(tol) calls of AgglomeratedTopologyCheck::computeBettiNumbersAgglomeratedEntities
// (src/topology/AgglomeratedTopologyCheck.cpp:241-318), all local boundary operators of a codimension in one launch
inline std::vector<int> BatchedRank(const std::vector<int> &nrows, const std::vector<int> &ncols, const std::vector<int64_t> &t_ptr,
                                    const std::vector<int> &t_row, const std::vector<int> &t_col, const std::vector<double> &t_val, double tol = 1e-9)
{
    std::vector<int> rank(nrows.size(), 0);
    check(elag_hip_batched_rank((int)nrows.size(), nrows.data(), ncols.data(), t_ptr.data(), t_row.data(), t_col.data(), t_val.data(), tol, rank.data()));
    return rank;
}

}  // namespace elag_hip
