"""The Gauss-Seidel relaxations of the reference's smoother table on the device (csrc/sweep.hip):
"L1 Gauss-Seidel" = 2 -- the type EVERY shipped parameter list selects (examples/testing_helpers/Create0FormParameterList.hpp:
103-116, Create2FormParameterList.hpp, example_parameterlists/*.xml) --, "L1 Gauss-Seidel Truncated" = 4, "Gauss-Seidel" = 6 and
"Hypre Jacobi" = 413 (src/linalg/factories/ParELAG_HypreSmootherFactory.cpp:28-31, 92-109).

The device runs a sweep as dependency levels; the rows of a level are independent and every row is summed by one lane in storage
order, so on an operator uploaded in the oracle's storage order the result is the sequential loop's, bit for bit.  Operators the
device builds itself (Galerkin products: columns ascending instead of hypre's diagonal-first order) agree to rounding."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from tests.helpers import oracle_h1_hierarchy, oracle_hdiv_hierarchy

pytestmark = pytest.mark.gpu

# the "Gauss-Seidel" entry of the reference's preconditioner library, verbatim (Create0FormParameterList.hpp:103-116)
REFERENCE_GS_BLOCK = {"Type": "L1 Gauss-Seidel", "Sweeps": 1, "Damping Factor": 1.0, "Omega": 1.0, "Cheby Poly Order": 2,
                      "Cheby Poly Fraction": 0.3}


@pytest.fixture(scope="module")
def la():
    from parelag_amd import linalg
    return linalg


@pytest.mark.parametrize("relax_type", [2, 4, 6])
@pytest.mark.parametrize("sweeps", [1, 2])
@pytest.mark.parametrize("iterative", [True, False])
def test_sweep_is_the_sequential_loop_bit_for_bit(la, h1_k2, relax_type, sweeps, iterative):
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)                                   # hypre's storage convention; uploaded as it is
    ref_s = orc.OSmoother(orc.OParCSR(Ad), type=relax_type, sweeps=sweeps)
    dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), relax_type, {"Sweeps": sweeps})
    ref_s.iterative_mode = dev_s.iterative_mode = iterative
    rng = np.random.default_rng(11)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    u, ur = u0.copy(), u0.copy()
    dev_s.Mult(f, u); ref_s.Mult(f, ur)
    assert np.array_equal(u, ur)
    # the symmetric sweep is self-adjoint: MultTranspose is the same map (mfem::HypreSmoother::MultTranspose)
    ut = u0.copy(); dev_s.MultTranspose(f, ut)
    assert np.array_equal(ut, ur)


@pytest.mark.parametrize("relax_type", [2, 3, 6])
def test_sweep_with_u_in_lds_is_the_same_sweep(la, h1_k2, relax_type, capfd):
    """ELAG_HIP_SWEEP_LDS=1 (off by default: measured slower, sweep.hip): the whole vector lives in the chain workgroup's LDS, every level
    joins the chain and the backward sweep follows the forward one in the same launch -- every bit of the result is the default plan's."""
    import os
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)
    dA = la.ParallelCSRMatrix.from_scipy(Ad)
    rng = np.random.default_rng(13)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    outs = []
    for lds in ("0", "1"):
        os.environ["ELAG_HIP_SWEEP_LDS"] = lds
        os.environ["ELAG_HIP_SWEEP_VERBOSE"] = "1"
        try:
            s = la.HypreSmootherWrapper(dA, relax_type, {"Sweeps": 2})
        finally:
            os.environ.pop("ELAG_HIP_SWEEP_LDS", None); os.environ.pop("ELAG_HIP_SWEEP_VERBOSE", None)
        assert ("(u in LDS)" in capfd.readouterr().err) == (lds == "1")
        s.iterative_mode = True
        u = u0.copy()
        s.Mult(f, u)
        outs.append(u)
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("relax_type", [2, 6])
def test_sweep_on_unsymmetric_patterns_and_awkward_rows(la, relax_type):
    """anti-dependencies (a_ij != 0, a_ji == 0, j > i) must also hold row j back; for type 6 also rows without entries and a
    stored zero diagonal (left alone, as in hypre_BoomerAMGRelax); a chain that makes every row its own level"""
    rng = np.random.default_rng(5)
    n = 700
    B = sp.random(n, n, density=0.01, random_state=rng, format="lil")
    B.setdiag(rng.uniform(2, 3, n))
    A = orc.diag_first(B.tocsr())
    if relax_type == 6:
        A = A.tolil()
        A[3, :] = 0                                                   # an empty row
        A = orc.diag_first(A.tocsr())
        A.eliminate_zeros()
        A = orc.diag_first(A)
        A.data[A.indptr[10]] = 0.0                                    # a stored zero on the diagonal (first entry of row 10)
    cases = [A, orc.diag_first(sp.diags([np.full(n - 1, -1.0), np.full(n, 2.0), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr())]
    for M in cases:
        f, u0 = rng.standard_normal(n), rng.standard_normal(n)
        ref_s = orc.OSmoother(orc.OParCSR(M), type=relax_type)
        dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(M), relax_type)
        u, ur = u0.copy(), u0.copy()
        dev_s.Mult(f, u); ref_s.Mult(f, ur)
        assert np.array_equal(u, ur)


def test_the_reference_s_own_smoother_block_is_accepted_unchanged(la, h1_k2):
    """Type / Sweeps / Damping Factor / Omega / Cheby keys exactly as the reference's lists spell them"""
    A, Ps, _ = h1_k2
    p = dict(REFERENCE_GS_BLOCK)
    S = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(orc.diag_first(A)), la.HYPRE_SMOOTHER_TYPES[p.pop("Type")], p)
    f = np.ones(A.shape[0]); u = np.zeros(A.shape[0]); ur = np.zeros(A.shape[0])
    S.Mult(f, u)
    orc.OSmoother(orc.OParCSR(orc.diag_first(A)), type=2).Mult(f, ur)
    assert np.array_equal(u, ur)
    # every name of the reference's table builds (ParELAG_HypreSmootherFactory.cpp:92-109); a number outside it says so
    from parelag_amd._lib import ElagHipError
    for name in la.HYPRE_SMOOTHER_TYPES:
        la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(orc.diag_first((A + 0.3 * sp.eye(A.shape[0])).tocsr())), la.HYPRE_SMOOTHER_TYPES[name])
    with pytest.raises(ElagHipError, match="not available"):
        la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(A), 7)
    # a zero l1 norm is an error in hypre_ParCSRComputeL1Norms, here too
    Z = sp.csr_matrix(A).tolil(); Z[5, :] = 0; Z = Z.tocsr(); Z.eliminate_zeros()
    with pytest.raises(ElagHipError, match="zero l1"):
        la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Z), 2)


def test_hypre_jacobi_413_is_diag_scale(la, h1_k2):
    A, _, _ = h1_k2
    S = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(A), la.HYPRE_SMOOTHER_TYPES["Hypre Jacobi"])
    rng = np.random.default_rng(3)
    f, u = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    ur = u.copy()
    S.Mult(f, u)                                                # iterative_mode = true, yet the old x is ignored (HYPRE_ParCSRDiagScale)
    orc.OSmoother(orc.OParCSR(orc.diag_first(A)), type=413).Mult(f, ur)
    assert np.array_equal(u, ur) and np.array_equal(u, f / A.diagonal())


@pytest.mark.parametrize("smoother", ["L1 Gauss-Seidel", "L1 Gauss-Seidel Truncated", "Gauss-Seidel"])
def test_vcycle_with_gauss_seidel_smoothers_matches_oracle(la, h1_k2, smoother):
    """buildHierarchyFromDeRhamSequence with the reference's default relaxation: device Galerkin operators (columns ascending)
    against the oracle's (hypre's RAP order): equal to rounding; PCG iteration counts equal"""
    A, Ps, _ = h1_k2
    t = la.HYPRE_SMOOTHER_TYPES[smoother]
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, smoother)
    h = oracle_h1_hierarchy(A, Ps, smoother=t)
    Hc = orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"])
    rng = np.random.default_rng(7)
    n = A.shape[0]
    for _ in range(2):
        b = rng.uniform(-1, 1, n)
        x, xr = np.full(n, 5.0), np.empty(n)
        H.Mult(b, x)
        Hc.apply(b, xr)
        assert np.max(np.abs(x - xr)) <= 1e-12 * np.max(np.abs(xr))
    # solver mode (iterative): sol <- sol + cycle(rhs - A sol)
    H.iterative_mode = False
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 100})
    rhs = rng.uniform(-1, 1, n)
    xs, xo = np.zeros(n), np.zeros(n)
    K.Mult(rhs, xs)
    ref = orc.mfem_cg(lambda v: A @ v, lambda r, z: Hc.apply(r, z), rhs, xo, rel_tol=1e-8, max_iter=100)
    assert K.GetConverged() and K.GetNumIterations() == ref.final_iter
    assert np.max(np.abs(xs - xo)) <= 1e-9 * np.max(np.abs(xo))


def test_gauss_seidel_cycle_converges_faster_than_l1_jacobi(la, h1_k2):
    """the reason the reference ships it: fewer PCG iterations for the same hierarchy"""
    A, Ps, _ = h1_k2
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    rhs = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    its = {}
    for sm in ("L1 Jacobi", "L1 Gauss-Seidel"):
        H = la.buildHierarchyFromDeRhamSequence(dA, dPs, sm)
        K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 200})
        x = np.zeros(A.shape[0]); K.Mult(rhs, x)
        assert K.GetConverged()
        its[sm] = K.GetNumIterations()
    assert its["L1 Gauss-Seidel"] < its["L1 Jacobi"]


def test_hiptmair_with_the_reference_s_gauss_seidel_smoothers(la):
    """MultigridTest2Form's list wraps two "L1 Gauss-Seidel" smoothers in Hiptmair (Create2FormParameterList.hpp): primary on A,
    auxiliary on D^T A D"""
    from parelag_amd import meshgen
    A, Ps, Ds, _ = meshgen.hdiv_problem(1)
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    dDs = [la.ParallelCSRMatrix.from_scipy(D) for D in Ds[:len(Ps)]]
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel", Ds=dDs, aux_smoother="L1 Gauss-Seidel")
    h = oracle_hdiv_hierarchy(A, Ps, Ds, primary=2, aux=2)
    Hc = orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"], hipt=h["hipt"])
    rng = np.random.default_rng(9)
    n = A.shape[0]
    b = rng.uniform(-1, 1, n)
    x, xr = np.zeros(n), np.empty(n)
    H.Mult(b, x)
    Hc.apply(b, xr)
    assert np.max(np.abs(x - xr)) <= 1e-11 * np.max(np.abs(xr))


@pytest.mark.parametrize("relax_type,weight,omega", [(2, 1.0, 1.0), (6, 1.0, 1.0), (2, 0.8, 1.2), (6, 0.9, 1.3)])
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_sweep_freezes_the_ghost_values(h1_k2, relax_type, weight, omega, world):
    """hybrid Gauss-Seidel across ranks (hypre_ParCSRRelax 2 / BoomerAMGRelax 6): every rank sweeps its own rows with the
    neighbours' values as they were when the sweep began -- in-process device ranks against the oracle's loop fed with exactly
    those ghost values, two sweeps (so the second one sees the neighbours' first); also hypre's weighted branches"""
    from parelag_amd import linalg as la, partition
    from tests.test_gpu_multirank import run_ranks
    A, Ps, seq = h1_k2
    d = partition.distributed_h1_hierarchy(A, Ps, seq, world)
    parts, starts, Ag = d["levels"][0]["A_parts"], d["levels"][0]["starts"], d["A_global"][0]
    n = Ag.shape[0]
    rng = np.random.default_rng(13)
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)

    def body(dist):
        r = dist.rank
        p = dict(parts[r]); p["diag"] = orc.diag_first(p["diag"])
        M = la.ParallelCSRMatrix.from_parts(p)
        S = la.HypreSmootherWrapper(M, relax_type, {"Sweeps": 2, "Damping Factor": weight, "Omega": omega})
        lo, hi = int(starts[r]), int(starts[r + 1])
        u = u0[lo:hi].copy()
        S.Mult(f[lo:hi], u)
        return u
    res = run_ranks(world, body)

    class Frozen(orc.SerialExchange):
        def __init__(self, p):
            self.p, self.u = p, None

        def forward(self, pkg, send_buf):
            return self.u[self.p["col_map_offd"]]
    ex = [Frozen(p) for p in parts]
    sm = []
    for r, p in enumerate(parts):
        M = orc.oparcsr_from_part(dict(p, diag=orc.diag_first(p["diag"])), ex[r])
        sm.append(orc.OSmoother(M, type=relax_type, sweeps=1, weight=weight, omega=omega))
    ug = u0.copy()
    for _ in range(2):
        for e in ex:
            e.u = ug.copy()
        new = ug.copy()
        for r in range(world):
            lo, hi = int(starts[r]), int(starts[r + 1])
            loc = ug[lo:hi].copy()
            sm[r].Mult(f[lo:hi], loc)
            new[lo:hi] = loc
        ug = new
    for r in range(world):
        lo, hi = int(starts[r]), int(starts[r + 1])
        assert np.array_equal(res[r], ug[lo:hi])


def _hybrid_reference(A, blocks, relax_type, sweeps, f, u, weight=1.0, omega=1.0):
    """the oracle's relaxation as hybrid Gauss-Seidel on `blocks` ranks with contiguous, even row blocks (u updated in place): rank by
    rank, rows of block b, columns split into the block's own (diag) and the others' (offd, ascending), the others' values frozen as
    they were when the symmetric sweep began"""
    n = A.shape[0]
    size, rest = n // blocks, n % blocks
    starts = np.array([b * size + min(b, rest) for b in range(blocks + 1)])
    Ac = sp.csr_matrix(A)

    class Frozen(orc.SerialExchange):
        def __init__(self, cols):
            self.cols, self.u = cols, None

        def forward(self, pkg, send_buf):
            return self.u[self.cols]
    sm, ex = [], []
    for b in range(blocks):
        lo, hi = int(starts[b]), int(starts[b + 1])
        rows = Ac[lo:hi].tocsr()
        own = (rows.indices >= lo) & (rows.indices < hi)
        rr = np.repeat(np.arange(hi - lo), np.diff(rows.indptr))
        cnt_d = np.bincount(rr[own], minlength=hi - lo); cnt_o = np.bincount(rr[~own], minlength=hi - lo)
        diag = sp.csr_matrix((rows.data[own], rows.indices[own] - lo, np.concatenate([[0], np.cumsum(cnt_d)])), shape=(hi - lo, hi - lo))
        col_map = np.unique(rows.indices[~own])
        offd = sp.csr_matrix((rows.data[~own], np.searchsorted(col_map, rows.indices[~own]), np.concatenate([[0], np.cumsum(cnt_o)])),
                             shape=(hi - lo, len(col_map)))
        part = dict(diag=diag, offd=offd, col_map_offd=col_map.astype(np.int64), first_row=lo, first_col=lo, global_shape=A.shape,
                    recv_procs=np.zeros(0, np.int32), recv_starts=np.zeros(1, np.int32), send_procs=np.zeros(0, np.int32),
                    send_starts=np.zeros(1, np.int32), send_elmts=np.zeros(0, np.int32))
        e = Frozen(col_map)
        ex.append(e)
        sm.append(orc.OSmoother(orc.oparcsr_from_part(part, e), type=relax_type, sweeps=1, weight=weight, omega=omega))
    ug = u.copy()
    for _ in range(sweeps):
        for e in ex:
            e.u = ug.copy()
        new = ug.copy()
        for b in range(blocks):
            lo, hi = int(starts[b]), int(starts[b + 1])
            loc = ug[lo:hi].copy()
            sm[b].Mult(f[lo:hi], loc)
            new[lo:hi] = loc
        ug = new
    u[:] = ug


@pytest.mark.parametrize("relax_type", [2, 4, 6])
@pytest.mark.parametrize("blocks", [2, 7, 64])
def test_virtual_ranks_are_the_reference_s_sweep_on_that_many_ranks(la, h1_k2, relax_type, blocks):
    """elag_hip_set_gs_hybrid_blocks(B): ONE device rank cuts its rows into B contiguous, even blocks and sweeps them as B MPI ranks would --
    Gauss-Seidel inside a block, the other blocks' values frozen as they were when the symmetric sweep began.  Bit for bit the oracle's loop
    run block by block on that partition with exactly those frozen values (two sweeps: the second sees the neighbours' first); B = 1 is the
    sequential loop again."""
    A = orc.diag_first(h1_k2[0])
    n = A.shape[0]
    rng = np.random.default_rng(17)
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    la.set_gs_hybrid_blocks(blocks)
    try:
        S = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(A), relax_type, {"Sweeps": 2})
    finally:
        la.set_gs_hybrid_blocks(0)
    u = u0.copy()
    S.Mult(f, u)
    ug = u0.copy()
    _hybrid_reference(A, blocks, relax_type, 2, f, ug)
    assert np.array_equal(u, ug)
    # one block: the sequential loop
    S1 = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(A), relax_type, {"Sweeps": 2})
    u1, us = u0.copy(), u0.copy()
    S1.Mult(f, u1)
    orc.OSmoother(orc.OParCSR(A), type=relax_type, sweeps=2).Mult(f, us)
    assert np.array_equal(u1, us) and not np.array_equal(u1, u)


@pytest.mark.parametrize("relax_type", [2, 4])
def test_l1_gauss_seidel_with_omega(la, h1_k2, relax_type):
    """relax_weight = 1, omega != 1: u_i += omega * res / l1_i in both directions (hypre_ParCSRRelax 2 / 4)"""
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)
    ref_s = orc.OSmoother(orc.OParCSR(Ad), type=relax_type, omega=1.3)
    dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), relax_type, {"Omega": 1.3})
    rng = np.random.default_rng(2)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    u, ur = u0.copy(), u0.copy()
    dev_s.Mult(f, u); ref_s.Mult(f, ur)
    assert np.array_equal(u, ur)


@pytest.mark.parametrize("relax_type,weight,omega", [(2, 0.8, 1.0), (4, 0.7, 1.3), (2, 1.25, 0.9), (6, 1.0, 1.2), (6, 0.8, 1.0), (6, 0.9, 1.4)])
@pytest.mark.parametrize("blocks", [1, 7])
def test_weighted_branches_of_the_gauss_seidel_relaxations(la, h1_k2, relax_type, weight, omega, blocks):
    """hypre's branches for "Damping Factor" != 1 (types 2 / 4: scaled SSOR with the saved pre-sweep vector) and for type 6 with either
    parameter off 1 (the reference forwards both keys unconditionally, ParELAG_HypreSmootherWrapper.cpp:28-33): bit for bit the oracle's
    loops, two sweeps, from a non-zero guess and from zero; also as hybrid Gauss-Seidel on 7 virtual ranks against the oracle on that partition"""
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)
    n = A.shape[0]
    rng = np.random.default_rng(21)
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    la.set_gs_hybrid_blocks(blocks)
    try:
        dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), relax_type, {"Sweeps": 2, "Damping Factor": weight, "Omega": omega})
    finally:
        la.set_gs_hybrid_blocks(0)
    if blocks == 1:
        ref = lambda f_, u_: orc.OSmoother(orc.OParCSR(Ad), type=relax_type, sweeps=2, weight=weight, omega=omega).Mult(f_, u_)
    else:
        ref = lambda f_, u_: _hybrid_reference(Ad, blocks, relax_type, 2, f_, u_, weight=weight, omega=omega)
    for iterative in (True, False):
        dev_s.iterative_mode = iterative
        u, ur = u0.copy(), (u0.copy() if iterative else np.zeros(n))
        dev_s.Mult(f, u); ref(f, ur)
        assert np.array_equal(u, ur)


def test_lumped_jacobi_is_l1_jacobi_with_row_sums(la, h1_k2):
    """"Lumped Jacobi" = 5: mfem::HypreSmoother takes l1 = A 1 (row sums) and runs relaxation type 1 with it -- on a matrix whose
    row sums are positive (a shifted stiffness matrix: the plain one has zero row sums, i.e. the reference divides by zero)"""
    A, _, _ = h1_k2
    B = orc.diag_first((A + 0.3 * sp.eye(A.shape[0])).tocsr())
    ref_s = orc.OSmoother(orc.OParCSR(B), type=5, sweeps=2, weight=0.9)
    dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(B), la.HYPRE_SMOOTHER_TYPES["Lumped Jacobi"], {"Sweeps": 2, "Damping Factor": 0.9})
    rng = np.random.default_rng(4)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    u, ur = u0.copy(), u0.copy()
    dev_s.Mult(f, u); ref_s.Mult(f, ur)
    assert np.max(np.abs(u - ur)) <= 1e-12 * np.max(np.abs(ur))
    rs = np.asarray(B.sum(axis=1)).ravel()
    assert np.max(np.abs(ur - (lambda v: v + 0.9 * (f - B @ v) / rs)(u0 + 0.9 * (f - B @ u0) / rs))) <= 1e-12 * np.max(np.abs(ur))


@pytest.mark.parametrize("omega", [1.0, 0.8])
@pytest.mark.parametrize("sweeps", [1, 2])
def test_kaczmarz_is_the_sequential_loop_bit_for_bit(la, h1_k2, omega, sweeps):
    """"Kaczmarz" = 3 (hypre_ParCSRRelax): per row res = (f_i - a_i.u) / |a_i|^2, then u_j += omega res a_ij on the row's columns,
    forward then backward.  The device orders two rows iff they share a column; the rows of a level share none."""
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)
    ref_s = orc.OSmoother(orc.OParCSR(Ad), type=3, sweeps=sweeps, omega=omega)
    dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), la.HYPRE_SMOOTHER_TYPES["Kaczmarz"], {"Sweeps": sweeps, "Omega": omega})
    rng = np.random.default_rng(17)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    for iterative in (True, False):
        ref_s.iterative_mode = dev_s.iterative_mode = iterative
        u, ur = u0.copy(), u0.copy()
        dev_s.Mult(f, u); ref_s.Mult(f, ur)
        assert np.array_equal(u, ur)
    # it is a solver for consistent systems: the error in the energy of A A^T goes down monotonically
    x = rng.standard_normal(A.shape[0]); b = Ad @ x
    dev_s.iterative_mode = True
    u = np.zeros(A.shape[0]); e0 = np.linalg.norm(x - u)
    for _ in range(3):
        dev_s.Mult(b, u)
        e1 = np.linalg.norm(x - u)
        assert e1 < e0
        e0 = e1


def test_kaczmarz_on_unsymmetric_patterns_long_rows_and_ranks(la, h1_k2):
    """rows longer than the lanes' registers hold (> 32 entries), an unsymmetric pattern, a tridiagonal chain; and two in-process
    ranks with frozen ghost values (the update touches local columns only)"""
    rng = np.random.default_rng(23)
    n = 600
    B = sp.random(n, n, density=0.02, random_state=rng, format="lil")
    B[7, :80] = rng.standard_normal(80)                                       # one long row
    B.setdiag(rng.uniform(2, 3, n))
    cases = [orc.diag_first(B.tocsr()), orc.diag_first(sp.diags([np.full(n - 1, -1.0), np.full(n, 2.0), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr())]
    for M in cases:
        f, u0 = rng.standard_normal(n), rng.standard_normal(n)
        u, ur = u0.copy(), u0.copy()
        la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(M), 3).Mult(f, u)
        orc.OSmoother(orc.OParCSR(M), type=3).Mult(f, ur)
        assert np.array_equal(u, ur)
    from parelag_amd import partition
    from parelag_amd._lib import ElagHipError
    from tests.test_gpu_multirank import run_ranks
    dup = sp.csr_matrix((np.array([2.0, 1.0, 1.0, 3.0]), np.array([0, 1, 1, 1]), np.array([0, 3, 4])), shape=(2, 2))
    with pytest.raises(ElagHipError, match="twice"):
        la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(dup), 3)
    A, Ps, seq = h1_k2
    d = partition.distributed_h1_hierarchy(A, Ps, seq, 2)
    parts, starts = d["levels"][0]["A_parts"], d["levels"][0]["starts"]
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])

    def body(dist):
        r = dist.rank
        p = dict(parts[r]); p["diag"] = orc.diag_first(p["diag"])
        S = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_parts(p), 3)
        lo, hi = int(starts[r]), int(starts[r + 1])
        u = u0[lo:hi].copy()
        S.Mult(f[lo:hi], u)
        return u
    res = run_ranks(2, body)

    class Frozen(orc.SerialExchange):
        def __init__(self, p):
            self.p = p

        def forward(self, pkg, send_buf):
            return u0[self.p["col_map_offd"]]
    for r, p in enumerate(parts):
        M = orc.oparcsr_from_part(dict(p, diag=orc.diag_first(p["diag"])), Frozen(p))
        lo, hi = int(starts[r]), int(starts[r + 1])
        loc = u0[lo:hi].copy()
        orc.OSmoother(M, type=3).Mult(f[lo:hi], loc)
        assert np.array_equal(res[r], loc)


@pytest.mark.parametrize("name,order", [("Taubin", 2), ("FIR", 2), ("FIR", 4), ("FIR", 1)])
def test_taubin_and_fir_polynomial_smoothers(la, h1_k2, name, order):
    """mfem::HypreSmoother types 1001 / 1002 with mfem's default options: the device's eigenvalue estimate (CG-Lanczos on A itself,
    halved) equals the oracle's, the polynomial is applied to 1e-12, in both modes and for two sweeps; the FIR coefficients are the
    windowed Chebyshev series of the pass band"""
    A, _, _ = h1_k2
    t = la.HYPRE_SMOOTHER_TYPES[name]
    Ad = orc.diag_first(A)
    dev_s = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), t, {"Sweeps": 2, "Cheby Poly Order": order, "Cheby Poly Fraction": 0.3})
    ref_s = orc.OSmoother(orc.OParCSR(Ad), type=t, sweeps=2, order=order, fraction=0.3)
    mx, _mn = dev_s.eig_estimate()
    assert abs(mx - ref_s.max_eig) <= 1e-10 * ref_s.max_eig
    lam = float(np.max(np.abs(np.linalg.eigvalsh(A.toarray())))) if A.shape[0] <= 1500 else None
    if lam is not None:
        assert 0.3 * lam <= 2 * mx <= 1.05 * lam                       # a Lanczos estimate from below
    rng = np.random.default_rng(29)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    for iterative in (True, False):
        ref_s.iterative_mode = dev_s.iterative_mode = iterative
        u, ur = u0.copy(), u0.copy()
        dev_s.Mult(f, u); ref_s.Mult(f, ur)
        assert np.max(np.abs(u - ur)) <= 1e-12 * np.max(np.abs(ur))
    # an injected estimate is used as given
    dev_s.set_eig_estimate(1.7 * mx, 0.0)
    ref2 = orc.OSmoother(orc.OParCSR(Ad), type=t, sweeps=2, order=order, fraction=0.3, eig=(2 * 1.7 * mx, 0.0))
    ref2.iterative_mode = dev_s.iterative_mode = True
    u, ur = u0.copy(), u0.copy()
    dev_s.Mult(f, u); ref2.Mult(f, ur)
    assert np.max(np.abs(u - ur)) <= 1e-12 * np.max(np.abs(ur))
    if name == "Taubin":
        # one pair of steps is the polynomial (1 - mu t)(1 - lambda t) in t = A / max_eig applied to the error: for f = 0, u -> p(A) u
        dev1 = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), t, {"Sweeps": 1})
        m1 = dev1.eig_estimate()[0]
        v = u0.copy(); dev1.Mult(np.zeros_like(f), v)
        w = u0.copy()
        for i in range(40):
            w = w - (0.5 if i % 2 == 0 else -0.5) * (Ad @ w) / m1
        assert np.max(np.abs(v - w)) <= 1e-11 * np.max(np.abs(w))


@pytest.mark.parametrize("smoother", ["Kaczmarz", "Taubin", "FIR"])
def test_vcycle_with_the_remaining_entries_of_the_smoother_table(la, h1_k2, smoother):
    """buildHierarchyFromDeRhamSequence with "Kaczmarz", "Taubin" and "FIR" on every level against Hierarchy::Mult composed from the
    oracle's smoothers (device Galerkin operators: columns ascending, so Kaczmarz agrees to rounding, not bit for bit)"""
    from tests.helpers import oracle_py_hierarchy
    A, Ps, _ = h1_k2
    t = la.HYPRE_SMOOTHER_TYPES[smoother]
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, smoother)
    Ho = oracle_py_hierarchy(oracle_h1_hierarchy(A, Ps, smoother=t))
    rng = np.random.default_rng(31)
    n = A.shape[0]
    b = rng.uniform(-1, 1, n)
    x, xr = np.full(n, 3.0), np.full(n, 3.0)
    H.Mult(b, x)
    Ho.Mult(b, xr)
    assert np.max(np.abs(x - xr)) <= 1e-10 * np.max(np.abs(xr))


def test_dense_tail_product_on_the_matrix_cores_is_the_scalar_product(la, h1_k2):
    """The dense operator of a Gauss-Seidel hierarchy's coarse tail is a chain of dense products; the one true GEMM of it (B_below times P^T R) runs on
    the matrix cores (dense.hip, dense_mm_mfma_kernel: v_mfma_f64_16x16x4_f64, 64 x 64 tiles) where ELAG_HIP_DENSE_MFMA=0 keeps one thread per
    entry.  Fused multiply-adds in ascending k against separately rounded products and sums: the two cycles agree to rounding, and each with the oracle."""
    import os
    A, Ps, _ = h1_k2
    t = la.HYPRE_SMOOTHER_TYPES["L1 Gauss-Seidel"]
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    h = oracle_h1_hierarchy(A, Ps, smoother=t)
    Hc = orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"])
    n = A.shape[0]
    b = np.random.default_rng(17).uniform(-1, 1, n)
    xr = np.empty(n)
    Hc.apply(b, xr)
    xs = []
    for env in ({}, {"ELAG_HIP_DENSE_MFMA": "0"}, {"ELAG_HIP_DENSE_STRIPS": "0"}):
        saved_env = {k: os.environ.get(k) for k in env}      # (a variable the run itself was started with is put back, not dropped)
        os.environ.update(env)
        try:
            H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel")
        finally:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        x = np.full(n, 3.0)
        H.Mult(b, x)
        xs.append(x)
        assert np.max(np.abs(x - xr)) <= 1e-12 * np.max(np.abs(xr))
    assert np.max(np.abs(xs[0] - xs[1])) <= 1e-13 * np.max(np.abs(xr))
    # the sweeps over the dense block: a workgroup per strip of 16 columns walking all levels (default) against a launch per level -- the same
    # operations entry by entry
    assert np.array_equal(xs[0], xs[2])


# ---- the dataflow sweep (sweep.hip, sweep_dataflow_kernel): one launch per symmetric sweep, rows waiting for the rows they read ---------------

def _with_env(env, fn):
    import os
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("relax_type", [2, 4, 6])
@pytest.mark.parametrize("sweeps", [1, 3])
def test_dataflow_sweep_is_the_sequential_loop_bit_for_bit(la, h1_k2, relax_type, sweeps, capfd):
    """forced on (ELAG_HIP_SWEEP_DATAFLOW=2) and off (0): the oracle's sequential loop both ways, from a guess and from zero, and applied
    repeatedly (the two granule arrays take turns: every launch runs on the array the launch before it zeroed)"""
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)
    dA = la.ParallelCSRMatrix.from_scipy(Ad)
    rng = np.random.default_rng(21)
    n = A.shape[0]
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    for mode, xcd in (("2", "0"), ("2", "1"), ("0", "0")):            # whole chip / confined to one XCD (plain stores through its L2) / level launches
        S = _with_env({"ELAG_HIP_SWEEP_DATAFLOW": mode, "ELAG_HIP_SWEEP_DATAFLOW_XCD": xcd, "ELAG_HIP_SWEEP_VERBOSE": "1"},
                      lambda: la.HypreSmootherWrapper(dA, relax_type, {"Sweeps": sweeps}))
        assert ("dataflow sweep" in capfd.readouterr().err) == (mode == "2")
        ref = orc.OSmoother(orc.OParCSR(Ad), type=relax_type, sweeps=sweeps)
        for iterative in (True, False):
            S.iterative_mode = ref.iterative_mode = iterative
            u, ur = u0.copy(), u0.copy()
            for _ in range(3):
                S.Mult(f, u); ref.Mult(f, ur)
                assert np.array_equal(u, ur)


@pytest.mark.parametrize("relax_type", [2, 6])
def test_dataflow_sweep_on_unsymmetric_patterns_long_and_empty_rows(la, relax_type, capfd):
    """a row that is read without reading back gets a wait-only entry (the host-built schedule); rows of more than 32 entries fetch their later
    chunks when their turn comes; type 6: rows without entries and a stored zero diagonal are left alone but still hand their value on"""
    rng = np.random.default_rng(5)
    n = 900
    B = sp.random(n, n, density=0.01, random_state=rng, format="lil")
    B.setdiag(rng.uniform(2, 3, n))
    for r in (7, 300, 640):                                           # long rows: 70 / 40 / 33 entries
        cols = rng.choice(n, size={7: 70, 300: 40, 640: 33}[r], replace=False)
        B[r, cols] = rng.uniform(-0.01, 0.01, cols.size)
        B[r, r] = 2.5
    A = orc.diag_first(B.tocsr())
    sym = orc.diag_first((A + A.T).tocsr())                            # structurally symmetric: the schedule is built on the device
    if relax_type == 6:
        A = A.tolil()
        A[3, :] = 0
        A = orc.diag_first(A.tocsr())
        A.eliminate_zeros()
        A = orc.diag_first(A)
        A.data[A.indptr[10]] = 0.0
    for M, xcd in ((A, "0"), (A, "1"), (sym, "0"), (sym, "1")):
        S = _with_env({"ELAG_HIP_SWEEP_DATAFLOW": "2", "ELAG_HIP_SWEEP_DATAFLOW_XCD": xcd, "ELAG_HIP_SWEEP_VERBOSE": "1"},
                      lambda: la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(M), relax_type, {"Sweeps": 2}))
        assert "dataflow sweep" in capfd.readouterr().err
        ref = orc.OSmoother(orc.OParCSR(M), type=relax_type, sweeps=2)
        f, u0 = rng.standard_normal(n), rng.standard_normal(n)
        u, ur = u0.copy(), u0.copy()
        S.Mult(f, u); ref.Mult(f, ur)
        assert np.array_equal(u, ur)


def test_gauss_seidel_cycle_with_and_without_the_dataflow_sweep(la, h1_k2):
    """the cycle the reference ships, with the dataflow sweeps on the levels above the dense tail and with the level launches: the same
    arithmetic row by row, so the same vector bit for bit; also replayed inside PCG (a graph where the hierarchy prefers one)"""
    A, Ps, _ = h1_k2
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    n = A.shape[0]
    rng = np.random.default_rng(33)
    b = rng.uniform(-1, 1, n)
    xs, its = [], []
    for mode in ("2", "0"):
        H = _with_env({"ELAG_HIP_SWEEP_DATAFLOW": mode}, lambda: la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel"))
        x = np.empty(n)
        for _ in range(3):
            H.Mult(b, x)
        xs.append(x)
        H.iterative_mode = False
        K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 100})
        y = np.zeros(n); K.Mult(b, y)
        assert K.GetConverged()
        its.append(K.GetNumIterations()); xs.append(y)
    assert np.array_equal(xs[0], xs[2]) and np.array_equal(xs[1], xs[3]) and its[0] == its[1]


def test_dataflow_sweep_that_gives_up_leaves_the_level_launches_in_charge():
    """every wait of the dataflow kernel is bounded; a kernel that runs out of time says so through pinned host memory and the library keeps
    the launch-per-level sweeps for the rest of the process.  Forced in a child process: the self-check at
    setup sees the give-up, warns once, and the sweep is still the sequential loop.  (The stall: a test switch keeps row 0's value from ever being
    handed on, so its readers wait until the 2 ms limit.)"""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np
        from oracle import oracle as orc
        from parelag_amd import linalg as la, meshgen
        A, _, _ = meshgen.h1_problem(2)
        Ad = orc.diag_first(A)
        S = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), 2, {"Sweeps": 1})
        rng = np.random.default_rng(1)
        f, u = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
        ur = u.copy()
        S.Mult(f, u)
        orc.OSmoother(orc.OParCSR(Ad), type=2).Mult(f, ur)
        assert np.array_equal(u, ur)
        S2 = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), 6, {"Sweeps": 1})      # (off for the process: no second warning)
        print("GAVE_UP_OK")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ELAG_HIP_SWEEP_DATAFLOW="2", ELAG_HIP_SWEEP_DATAFLOW_MS="2", ELAG_HIP_SWEEP_DATAFLOW_STALL_ROW="0")
    env.pop("ELAG_HIP_QUIET", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "GAVE_UP_OK" in r.stdout, r.stdout + r.stderr
    assert r.stderr.count("gave up waiting") == 1, r.stderr


def test_dataflow_sweep_that_gives_up_later_is_reported_by_the_next_call():
    """a give-up after the self-check -- in production: the device held by something that keeps the kernel's workgroups from running -- cannot be
    repaired (the sweep works in place), so it must not pass silently: the kernel stores a code in pinned host memory and the ABI call that
    follows fails with ELAG_HIP_ERR_STATE and says why; the library stays usable (the level launches, a new smoother)."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os
        import numpy as np
        from oracle import oracle as orc
        from parelag_amd import linalg as la, meshgen, _lib
        A, _, _ = meshgen.h1_problem(2)
        Ad = orc.diag_first(A)
        dA = la.ParallelCSRMatrix.from_scipy(Ad)
        S = la.HypreSmootherWrapper(dA, 2, {"Sweeps": 1})                # (self-check passes: nothing stalls yet)
        rng = np.random.default_rng(1)
        f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
        ur = u0.copy(); orc.OSmoother(orc.OParCSR(Ad), type=2).Mult(f, ur)
        u = u0.copy(); S.Mult(f, u)
        assert np.array_equal(u, ur)
        os.environ["ELAG_HIP_SWEEP_DATAFLOW_STALL_ROW"] = "0"             # from here on row 0 never hands its value on
        df, du = la.DeviceVector.from_numpy(f), la.DeviceVector.from_numpy(u0)
        failed = False
        try:
            S.Mult(df, du)                                               # the launch itself returns; the code arrives with the kernel
            la.synchronize()
            S.Mult(df, du)
        except _lib.ElagHipError as e:
            failed = "gave up" in str(e)
        assert failed
        os.environ.pop("ELAG_HIP_SWEEP_DATAFLOW_STALL_ROW")
        os.environ["ELAG_HIP_SWEEP_DATAFLOW"] = "0"
        S2 = la.HypreSmootherWrapper(dA, 2, {"Sweeps": 1})               # the library goes on: the launch-per-level sweeps
        u = u0.copy(); S2.Mult(f, u)
        assert np.array_equal(u, ur)
        print("REPORTED_OK")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ELAG_HIP_SWEEP_DATAFLOW="2", ELAG_HIP_SWEEP_DATAFLOW_MS="2")
    env.pop("ELAG_HIP_SWEEP_DATAFLOW_STALL_ROW", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert r.returncode == 0 and "REPORTED_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("relax_type", [2, 6])
def test_dataflow_schedule_built_on_the_device_gives_the_same_sweep(la, h1_k2, relax_type, capfd):
    """large operators get their dependency levels, their schedule and their entries from kernels (SweepSmoother::setup_device: the levels by a
    dataflow kernel in natural row order, the rows dealt to their levels in whatever order the atomics fall); forced here on a small one
    (ELAG_HIP_SWEEP_DEVICE_SETUP_ROWS): the sweep is the oracle's sequential loop bit for bit, like the host-built schedule's -- and an operator
    with an unsymmetric pattern goes back to the host path."""
    A, _, _ = h1_k2
    Ad = orc.diag_first(A)
    dA = la.ParallelCSRMatrix.from_scipy(Ad)
    rng = np.random.default_rng(44)
    n = A.shape[0]
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    ref = orc.OSmoother(orc.OParCSR(Ad), type=relax_type, sweeps=2)
    ref.iterative_mode = True
    ur = u0.copy(); ref.Mult(f, ur)
    for xcd in ("0", "1"):
        S = _with_env({"ELAG_HIP_SWEEP_DEVICE_SETUP_ROWS": "1000", "ELAG_HIP_SWEEP_DATAFLOW": "1", "ELAG_HIP_SWEEP_DATAFLOW_XCD": xcd, "ELAG_HIP_SWEEP_VERBOSE": "1"},
                      lambda: la.HypreSmootherWrapper(dA, relax_type, {"Sweeps": 2}))
        assert "schedule built on the device" in capfd.readouterr().err
        S.iterative_mode = True
        for _ in range(2):
            u = u0.copy(); S.Mult(f, u)
            assert np.array_equal(u, ur)
    # a pattern with one-way entries: the device path notices and leaves the operator to the host path (wait-only entries)
    B = sp.random(1500, 1500, density=0.01, random_state=rng, format="lil")
    B.setdiag(rng.uniform(2, 3, 1500))
    Bd = orc.diag_first(B.tocsr())
    S = _with_env({"ELAG_HIP_SWEEP_DEVICE_SETUP_ROWS": "1000", "ELAG_HIP_SWEEP_DATAFLOW": "2", "ELAG_HIP_SWEEP_VERBOSE": "1"},
                  lambda: la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Bd), relax_type, {"Sweeps": 1}))
    err = capfd.readouterr().err
    assert "schedule built on the device" not in err and "dataflow sweep" in err
    g, v0 = rng.standard_normal(1500), rng.standard_normal(1500)
    v, vr = v0.copy(), v0.copy()
    S.Mult(g, v); orc.OSmoother(orc.OParCSR(Bd), type=relax_type).Mult(g, vr)
    assert np.array_equal(v, vr)
