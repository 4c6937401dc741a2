// Blocked (saddle-point) layer: MfemBlockOperator, the Block Jacobi / Gauss-Seidel / LDU smoothers and the explicit
// Schur complement of the reference, laid out for the GPU:
//   * a block operator keeps its ParCSR blocks (needed by the triangular solves, the Schur complement and the per-block
//     Galerkin products) AND one merged "monolithic" ParCSR in which block-row i / block-column j occupy the local index
//     ranges row_off[i].. / col_off[j]..  Every product with the whole operator (residuals, V-cycle transfers) is then a
//     single fused SpMV launch with a single halo exchange instead of one launch per block.
//   * block vectors are plain contiguous device arrays; a block is a pointer offset (mfem::BlockVector views).
#pragma once
#include "smoother.hpp"

struct elag_hip_blockop_s {
    int nbr = 0, nbc = 0;
    std::vector<int> row_off, col_off;        // local offsets, nbr+1 / nbc+1 entries (MfemBlockOperator::RowOffsets_)
    std::vector<elag::ParCSR *> blk;          // nbr*nbc, null = zero block
    std::vector<char> owned;                  // blocks this object must delete
    std::unique_ptr<elag::ParCSR> mono;       // built on first use

    ~elag_hip_blockop_s();
    elag::ParCSR *block(int i, int j) const { return blk[(size_t)i * nbc + j]; }
    void set_block(int i, int j, elag::ParCSR *M, bool own);
    elag::ParCSR *monolithic();
};

namespace elag {

// BlockDiagonalSolver (kind 0), BlockTriangularSolver lower / upper (1 / 2), Block2x2LDUInverseOperator (3)
struct BlockSmoother : elag_hip_smoother_s {
    elag_hip_blockop_s *B = nullptr;
    int kind = 0;
    std::vector<elag_hip_smoother_s *> inv;   // kinds 0-2: one per block; kind 3: {A00_1, A00_2, A00_3, S}
    double damping = 1.0;
    bool fold = true;                         // kind 3: signs and the last subtraction ride on the products' epilogues (ELAG_HIP_LDU_FOLD=0: a vector pass each)
    double *resid = nullptr, *corr = nullptr, *tmp = nullptr, *spare = nullptr, *dscale = nullptr;
    ~BlockSmoother() override;
    void setup();
    void apply(const double *f, double *&cur, double *&alt, bool zero_guess, bool transpose) override;
    int flips(bool zero_guess) const override { return (dscale && !zero_guess) ? 1 : 0; }
    const double *diag_scaling() const override { return dscale; }
    bool same_scaling_as(const elag_hip_smoother_s *o) const override;
    bool graph_capturable() const override;
private:
    void inner(elag_hip_smoother_s *S, const double *rhs, double *out, double *scratch, bool transpose);
    void block_solve(const double *r, double *t, bool transpose);
};

}  // namespace elag
