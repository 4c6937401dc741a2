"""Parity of the device's blocked (saddle-point) layer -- SURVEY 8f-2 -- against the oracle: MfemBlockOperator,
SchurComplementFactory, Block Jacobi / Gauss-Seidel / LDU smoothers, the dense LU coarse solve and the blocked AMGe
V-cycle of MultigridTestDarcy (examples/MultigridTestDarcy.cpp:417-441, darcy_example_parameters.xml)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from tests.helpers import BLOCK2X2_FIXTURE as FX, gmres, oracle_block_solver, oracle_blocked_hierarchy

pytestmark = pytest.mark.gpu

TOL = 1e-11


@pytest.fixture(scope="module")
def la():
    from parelag_amd import linalg
    return linalg


def close(x, ref, tol=TOL):
    return np.max(np.abs(np.asarray(x) - np.asarray(ref))) <= tol * max(np.max(np.abs(ref)), 1e-300)


def device_blockop(la, blocks, offsets):
    op = la.MfemBlockOperator(offsets)
    for (i, j), M in blocks.items():
        op.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(orc.csr(M)))
    return op


def fixture_blocks():
    return {(0, 0): FX["M"], (0, 1): FX["Bt"], (1, 0): FX["B"], (1, 1): FX["C"]}


@pytest.fixture(scope="module")
def darcy_small():
    """kuhn_cube(2) refined twice: 3 levels, 9984+4608 -> 1248+576 -> 156+72... fine enough for every code path"""
    from parelag_amd import meshgen
    d = meshgen.darcy_problem(0, seq=meshgen.Sequence.kuhn_cube(2, 2))
    d["blocks"] = {(0, 0): d["M"], (0, 1): d["Bt"], (1, 0): d["B"]}
    n0, n1 = d["M"].shape[0], d["B"].shape[0]
    d["off"] = [0, n0, n0 + n1]
    return d


# ---- block operator -------------------------------------------------------------------------------------------------

def test_block_operator_on_the_block2x2_fixture(la):
    op = device_blockop(la, fixture_blocks(), [0, 4, 6])
    assert op.GetNumBlockRows() == 2 and op.Height() == 6 and not op.IsZeroBlock(1, 1)
    r = FX["b"].copy()
    op.Mult(FX["x0"], r, alpha=-1.0, beta=1.0)
    assert np.array_equal(r, FX["residual"])          # small integers: exact
    dense = sp.bmat([[FX["M"], FX["Bt"]], [FX["B"], FX["C"]]]).toarray()
    y = np.zeros(6)
    op.MultTranspose(FX["x0"], y)
    assert close(y, dense.T @ FX["x0"])
    mono = op.Monolithic().to_scipy()
    assert np.array_equal(mono.toarray(), dense)


def test_block_operator_zero_block_rectangular_and_errors(la, darcy_small):
    d = darcy_small
    op = device_blockop(la, d["blocks"], d["off"])
    assert op.IsZeroBlock(1, 1)
    ref = sp.bmat([[d["M"], d["Bt"]], [d["B"], None]]).tocsr()
    mono = op.Monolithic().to_scipy()
    mono.sort_indices(); ref.sort_indices()
    assert np.array_equal(mono.indptr, ref.indptr) and np.array_equal(mono.indices, ref.indices)
    assert np.array_equal(mono.data, ref.data)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, ref.shape[1]); y = rng.uniform(-1, 1, ref.shape[0]); y0 = y.copy()
    oop = orc.OBlockOperator(d["off"], d["off"], {k: orc.OParCSR(orc.csr(v)) for k, v in d["blocks"].items()})
    yr = y0.copy()
    op.Mult(x, y, alpha=0.5, beta=-2.0)
    oop.matvec(0.5, x, -2.0, yr)
    assert close(y, yr, 1e-12)
    # rectangular block operator: P = blockdiag(P2, P3)
    P2, P3 = d["Ps"][0]
    coff = [0, P2.shape[1], P2.shape[1] + P3.shape[1]]
    Pop = la.MfemBlockOperator(d["off"], coff)
    Pop.SetBlock(0, 0, la.ParallelCSRMatrix.from_scipy(orc.csr(P2)))
    Pop.SetBlock(1, 1, la.ParallelCSRMatrix.from_scipy(orc.csr(P3)))
    xc = rng.uniform(-1, 1, coff[-1]); yf = np.zeros(d["off"][-1])
    Pop.Mult(xc, yf)
    assert close(yf, sp.block_diag([P2, P3]) @ xc, 1e-13)
    yc = np.zeros(coff[-1])
    Pop.MultTranspose(x, yc)
    assert close(yc, sp.block_diag([P2, P3]).T @ x, 1e-13)
    # SetBlock size check (MfemBlockOperator.cpp:107-118)
    with pytest.raises(la._lib.ElagHipError, match="wrong size"):
        op.SetBlock(1, 1, la.ParallelCSRMatrix.from_scipy(orc.csr(d["M"])))
    with pytest.raises(la._lib.ElagHipError, match="out of range"):
        op.SetBlock(2, 0, la.ParallelCSRMatrix.from_scipy(orc.csr(d["M"])))


# ---- Schur complement -----------------------------------------------------------------------------------------------

@pytest.mark.parametrize("s_type", ["DIAGONAL", "ABSROWSUM"])
@pytest.mark.parametrize("negate", [False, True])
def test_schur_complement_matches_oracle(la, darcy_small, s_type, negate):
    d = darcy_small
    for blocks, off in ((fixture_blocks(), [0, 4, 6]), (d["blocks"], d["off"])):
        op = device_blockop(la, blocks, off)
        S = la.SchurComplement(op, s_type, alpha=0.8, negate=negate).to_scipy()
        ref = orc.schur_complement(blocks[(0, 0)], blocks[(0, 1)], blocks[(1, 0)], blocks.get((1, 1)), s_type, 0.8)
        ref = orc.csr(ref * (-1.0 if negate else 1.0))
        S.sort_indices(); ref.sort_indices()
        assert np.array_equal(S.indptr, ref.indptr) and np.array_equal(S.indices, ref.indices)
        assert close(S.data, ref.data, 1e-13)
    with pytest.raises(la._lib.ElagHipError, match="invalid"):
        import ctypes
        out = ctypes.c_void_p()
        la._lib.check(la._lib.init().elag_hip_schur_complement(op.h, 7, 1.0, 0, ctypes.byref(out)))


# ---- block smoothers --------------------------------------------------------------------------------------------------

INNER = {
    "l1": (dict(type=1), ("L1 Jacobi", {})),
    "l1x2": (dict(type=1, sweeps=2, weight=0.8), ("L1 Jacobi", {"Sweeps": 2, "Damping Factor": 0.8})),
    "cheby": (dict(type=16, order=2), ("Chebyshev", {"Cheby Poly Order": 2})),
}


def device_block_solver(la, blocks, off, kind, inner, s_type="DIAGONAL", alpha=1.0, use_negative_s=True, damping=1.0):
    op = device_blockop(la, blocks, off)
    S = la.SchurComplement(op, s_type, alpha, use_negative_s) if s_type != "NONE" else op.GetBlock(1, 1)
    name, params = INNER[inner][1]
    t = la.HYPRE_SMOOTHER_TYPES[name]
    mk = lambda M: la.HypreSmootherWrapper(M, t, params)
    if kind == "Block LDU":
        i0 = mk(op.GetBlock(0, 0))
        invs = [i0, i0, i0, mk(S)]
    else:
        invs = [mk(op.GetBlock(0, 0)), mk(S)]
    sol = la.BlockSolver(op, invs, kind, damping)
    sol._keep_S = S
    eigs = None
    if t == 16:
        eigs = (invs[0].eig_estimate(), invs[-1].eig_estimate())
    return sol, eigs


@pytest.mark.parametrize("kind", ["Block Jacobi", "Block GS Lower", "Block GS Upper", "Block LDU"])
@pytest.mark.parametrize("inner", ["l1", "l1x2", "cheby"])
@pytest.mark.parametrize("iterative", [True, False])
def test_block_smoothers_match_oracle(la, darcy_small, kind, inner, iterative):
    d = darcy_small
    sol, eigs = device_block_solver(la, d["blocks"], d["off"], kind, inner, damping=0.9 if kind == "Block LDU" else 1.0)
    spec = INNER[inner][0]
    osol, _op, _S = oracle_block_solver(d["blocks"], d["off"], kind=kind, a00=spec, a11=spec, eigs=eigs,
                                       damping=0.9 if kind == "Block LDU" else 1.0)
    sol.iterative_mode = osol.iterative_mode = iterative
    rng = np.random.default_rng(11)
    n = d["off"][-1]
    b = rng.uniform(-1, 1, n); x0 = rng.uniform(-1, 1, n)
    x, xr = x0.copy(), x0.copy()
    sol.Mult(b, x)
    osol.Mult(b, xr)
    assert close(x, xr)
    if kind.startswith("Block GS"):
        x, xr = x0.copy(), x0.copy()
        sol.MultTranspose(b, x)
        osol.MultTranspose(b, xr)
        assert close(x, xr)


def test_block_gs_on_the_fixture_with_a11_and_positive_s(la):
    """the 4 stored blocks of block2x2_test.cpp, "Use Negative S" = false (negative l1 norms flip the sweep's sign)"""
    for neg in (True, False):
        sol, _ = device_block_solver(la, fixture_blocks(), [0, 4, 6], "Block GS Lower", "l1", use_negative_s=neg)
        osol, _o, _s = oracle_block_solver(fixture_blocks(), [0, 4, 6], kind="Block GS Lower", use_negative_s=neg)
        x, xr = FX["x0"].copy(), FX["x0"].copy()
        sol.Mult(FX["b"], x); osol.Mult(FX["b"], xr)
        assert close(x, xr, 1e-13)


def test_block_smoother_argument_checks(la, darcy_small):
    d = darcy_small
    op = device_blockop(la, d["blocks"], d["off"])
    s0 = la.HypreSmootherWrapper(op.GetBlock(0, 0), 1)
    with pytest.raises(la._lib.ElagHipError, match="one inverse per block"):
        la.BlockSolver(op, [s0], "Block Jacobi")
    with pytest.raises(la._lib.ElagHipError, match="wrong size"):
        la.BlockSolver(op, [s0, s0], "Block Jacobi")


# ---- coarse solve ---------------------------------------------------------------------------------------------------

def test_dense_lu_coarse_solver(la, darcy_small):
    d = darcy_small
    _H, lev = oracle_blocked_hierarchy(d["blocks"], d["Ps"], kind="Block Jacobi")
    blk = lev[-1]
    A = sp.bmat([[blk[(0, 0)], blk[(0, 1)]], [blk[(1, 0)], None]]).tocsr()
    Ad = la.ParallelCSRMatrix.from_scipy(orc.csr(A))
    C = la.CoarseSolver(Ad, kind=1)
    rng = np.random.default_rng(2)
    b = rng.uniform(-1, 1, A.shape[0]); x = np.zeros_like(b)
    C.Mult(b, x)
    ref = np.linalg.solve(A.toarray(), b)
    assert close(x, ref, 1e-10)
    assert np.max(np.abs(A @ x - b)) <= 1e-11 * np.max(np.abs(b)) * np.linalg.cond(A.toarray()) ** 0.5
    # needs pivoting: leading zero, and a singular matrix is reported
    Z = sp.csr_matrix(np.array([[0., 2, 0], [1, 0, 0], [0, 0, 4]]))
    Cz = la.CoarseSolver(la.ParallelCSRMatrix.from_scipy(orc.csr(Z)), kind=1)
    xz = np.zeros(3); Cz.Mult(np.array([2., 3, 4]), xz)
    assert np.allclose(xz, [3, 1, 1], atol=1e-15)
    with pytest.raises(la._lib.ElagHipError, match="singular"):
        la.CoarseSolver(la.ParallelCSRMatrix.from_scipy(orc.csr(sp.csr_matrix(np.array([[1., 2], [2, 4]])))), kind=1)
    with pytest.raises(la._lib.ElagHipError, match="positive definite"):
        la.CoarseSolver(Ad, kind=0)                  # the saddle-point operator is not SPD


# ---- blocked hierarchy ------------------------------------------------------------------------------------------------

def device_blocked_hierarchy(la, d, use_graph=True, **kw):
    op = device_blockop(la, d["blocks"], d["off"])
    Ps = [[la.ParallelCSRMatrix.from_scipy(orc.csr(P)) for P in Pl] for Pl in d["Ps"]]
    return la.buildBlockedHierarchyFromDeRhamSequence(op, Ps, use_graph=use_graph, **kw), op


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("kind,inner", [("Block Jacobi", "l1"), ("Block Jacobi", "l1x2"), ("Block Jacobi", "cheby"),
                                        ("Block GS", "l1"), ("Block GS Upper", "cheby"), ("Block LDU", "l1")])
def test_blocked_vcycle_matches_oracle(la, darcy_small, kind, inner, use_graph):
    d = darcy_small
    name, params = INNER[inner][1]
    H, _op = device_blocked_hierarchy(la, d, use_graph, block_smoother=kind, a00_inverse=name, a00_params=params,
                                      a11_inverse=name, a11_params=params)
    eigs = None
    if inner == "cheby":   # inject the device's eigenvalue estimates (their own parity is tested separately)
        eigs = []
        for l in range(len(d["Ps"])):
            ob = {k: orc.csr(v) for k, v in _level_blocks(d, l).items()}
            opl = device_blockop(la, ob, [0, ob[(0, 0)].shape[0], ob[(0, 0)].shape[0] + ob[(1, 0)].shape[0]])
            S = la.SchurComplement(opl, "DIAGONAL", 1.0, True)
            e0 = la.HypreSmootherWrapper(opl.GetBlock(0, 0), 16, params).eig_estimate()
            e1 = la.HypreSmootherWrapper(S, 16, params).eig_estimate()
            eigs.append((e0, e1))
    spec = INNER[inner][0]
    Ho, _lev = oracle_blocked_hierarchy(d["blocks"], d["Ps"], eigs=eigs, kind="Block GS Lower" if kind == "Block GS" else kind,
                                        a00=spec, a11=spec)
    rng = np.random.default_rng(7)
    n = d["off"][-1]
    for trial in range(3):
        b = rng.uniform(-1, 1, n)
        x, xr = np.full(n, 3.0), np.zeros(n)
        H.Mult(b, x)
        Ho.Mult(b, xr)
        assert close(x, xr, 1e-10)
    # solver mode (iterative_mode): sol += cycle(rhs - A sol)
    H.iterative_mode = True; Ho.preconditioner = False
    x0 = rng.uniform(-1, 1, n); x, xr = x0.copy(), x0.copy()
    H.Mult(b, x); Ho.Mult(b, xr)
    assert close(x, xr, 1e-10)


def _level_blocks(d, l):
    blk = {k: orc.csr(v) for k, v in d["blocks"].items()}
    for Pl in d["Ps"][:l]:
        Pc = [orc.csr(P) for P in Pl]
        blk = {(i, j): orc.rap(Pc[i], A, Pc[j]) for (i, j), A in blk.items()}
    return blk


def test_blocked_hierarchy_coarse_operators_are_the_blockwise_galerkin_products(la, darcy_small):
    d = darcy_small
    H, _op = device_blocked_hierarchy(la, d)
    for l in range(len(d["Ps"]) + 1):
        blk = _level_blocks(d, l)
        ref = sp.bmat([[blk[(0, 0)], blk[(0, 1)]], [blk[(1, 0)], None]]).tocsr()
        ref.sort_indices()
        A = H.GetOperator(l).to_scipy(); A.sort_indices()
        assert A.shape == ref.shape and np.array_equal(A.indptr, ref.indptr) and np.array_equal(A.indices, ref.indices)
        assert close(A.data, ref.data, 1e-12)


def test_gmres_with_the_blocked_vcycle_converges_like_the_oracle(la, darcy_small):
    """MultigridTestDarcy's solve: GMRES preconditioned by the blocked AMGe cycle (darcy_example_parameters.xml).
    "Use Negative S" = false keeps the Schur block's sign (S = -B D^-1 B^T is negative definite), which makes the block
    Gauss-Seidel sweep a convergent smoother; with the factories' default (true) both the oracle and the device need
    > 100 iterations -- the XML's own comment says as much ("This won't work super well")."""
    d = darcy_small
    H, op = device_blocked_hierarchy(la, d, block_smoother="Block GS", use_negative_s=False)
    Ho, _ = oracle_blocked_hierarchy(d["blocks"], d["Ps"], kind="Block GS Lower", use_negative_s=False)
    A = sp.bmat([[d["M"], d["Bt"]], [d["B"], None]]).tocsr()
    n = A.shape[0]
    b = np.concatenate([np.zeros(d["off"][1]), d["seq"].meshes[0].M3() @ np.ones(d["off"][2] - d["off"][1])])

    def dev(r, z):
        H.Mult(r, z)

    def ora(r, z):
        z[:] = 0.0
        Ho.Mult(r, z)
    x, it = gmres(lambda v: A @ v, dev, b, rtol=1e-8)
    xo, ito = gmres(lambda v: A @ v, ora, b, rtol=1e-8)
    assert abs(it - ito) <= 1 and it < 60      # (the oracle stops at 9.998e-9 on iteration 43: a threshold crossing)
    assert np.linalg.norm(A @ x - b) <= 1e-7 * np.linalg.norm(b)
    assert close(x, xo, 1e-6)


def test_heterogeneous_coefficient_darcy_cycle_matches_oracle(la):
    """BASELINE configs[4] (MultigridTestSPE10): the same blocked path with a strongly heterogeneous inverse permeability
    in the velocity mass matrix (synthetic log-normal field, contrast ~1e5; the SPE10 data file is not in the reference)."""
    from parelag_amd import meshgen
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    kinv = meshgen.lognormal_inverse_permeability(seq, 2.0)
    assert kinv.max() / kinv.min() > 1e4
    d = meshgen.darcy_problem(0, seq=seq, kinv=kinv)
    d["blocks"] = {(0, 0): d["M"], (0, 1): d["Bt"], (1, 0): d["B"]}
    d["off"] = [0, d["M"].shape[0], d["M"].shape[0] + d["B"].shape[0]]
    for kind in ("Block Jacobi", "Block LDU"):
        H, _op = device_blocked_hierarchy(la, d, block_smoother=kind, use_negative_s=False)
        Ho, _ = oracle_blocked_hierarchy(d["blocks"], d["Ps"], kind=kind, use_negative_s=False)
        b = np.random.default_rng(31).uniform(-1, 1, d["off"][-1])
        x, xr = np.zeros_like(b), np.zeros_like(b)
        H.Mult(b, x); Ho.Mult(b, xr)
        assert close(x, xr, 1e-10)


import contextlib


@contextlib.contextmanager
def _env(**kv):
    import os
    saved = {k: os.environ.get(k) for k in kv}      # (a variable the run itself was started with is put back, not dropped)
    os.environ.update(kv)
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# ---- AMGe hierarchies as the inner inverses of the block solvers ("A00_1 Inverse" = "AMGe-GS", spe10_example_parameters.xml:114-127)

def test_block_ldu_with_inner_amge_hierarchies(la, darcy_small):
    """GMRES preconditioned by Block2x2LDUInverseOperator whose four inverses are AMGe V-cycles: three on A00 = M (2-form
    hierarchy) and one on the Schur complement, applied as -(B diag(M)^-1 Bt)^-1 by a hierarchy built on the positive matrix
    with the 3-form interpolators -- the solver graph of the reference's SPE10 list with the BoomerAMG entry replaced by an
    AMGe cycle.  Far fewer iterations than with Jacobi-class inner inverses, same counts as the oracle composition."""
    from tests.helpers import oracle_h1_hierarchy, oracle_py_hierarchy, oracle_inner_smoother
    d = darcy_small
    M, B, Bt = orc.csr(d["M"]), orc.csr(d["B"]), orc.csr(d["Bt"])
    n0, n1 = M.shape[0], B.shape[0]
    A = sp.bmat([[M, Bt], [B, None]]).tocsr()
    Spos = orc.csr(B @ sp.diags(1.0 / M.diagonal()) @ Bt)
    P2s, P3s = [orc.csr(p[0]) for p in d["Ps"]], [orc.csr(p[1]) for p in d["Ps"]]
    op = device_blockop(la, d["blocks"], d["off"])

    def dev_h(Aop, Pl):
        dA = la.ParallelCSRMatrix.from_scipy(Aop)
        return la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(P) for P in Pl], "L1 Jacobi")
    HM, HS = dev_h(M, P2s), dev_h(Spos, P3s)
    iM, iS = la.HierarchySolver(HM), la.HierarchySolver(HS, scale=-1.0)
    ldu = la.Block2x2LDUInverseOperator(op, iM, iM, iM, iS)
    ldu.iterative_mode = False

    class OH:
        iterative_mode = False

        def __init__(self, Aop, Pl, scale):
            self.H = oracle_py_hierarchy(oracle_h1_hierarchy(Aop, Pl, smoother=1))
            self.scale = scale

        def Mult(self, r, z):
            z[:] = 0.0
            self.H.Mult(r, z)
            z *= self.scale
    oM, oS = OH(M, P2s, 1.0), OH(Spos, P3s, -1.0)
    oop = orc.OBlockOperator(d["off"], d["off"], {k: orc.OParCSR(orc.csr(v)) for k, v in d["blocks"].items()})
    oldu = orc.OBlock2x2LDU(oop, oM, oM, oM, oS)
    rng = np.random.default_rng(31)
    r = rng.uniform(-1, 1, n0 + n1)
    z, zr = np.full(n0 + n1, 9.0), np.zeros(n0 + n1)
    ldu.Mult(r, z)
    oldu.Mult(r, zr)
    assert close(z, zr, 1e-10)
    # the S inverse's factor -1 rides on the sign of the product that forms its right-hand side (blocked.hip: fold); with a pass of its own
    # (ELAG_HIP_LDU_FOLD=0) the result is the same to the last bit
    with _env(ELAG_HIP_LDU_FOLD="0"):
        ldu0 = la.Block2x2LDUInverseOperator(op, iM, iM, iM, iS)
    ldu0.iterative_mode = False
    z0 = np.full(n0 + n1, 7.0)
    ldu0.Mult(r, z0)
    assert np.array_equal(z, z0)
    # the inner solver alone, and its sign
    y, yr = np.zeros(n1), np.zeros(n1)
    iS.Mult(r[n0:], y)
    oS.Mult(r[n0:], yr)
    assert close(y, yr, 1e-11) and (r[n0:] @ y) < 0.0
    rhs = np.concatenate([np.zeros(n0), d["seq"].meshes[0].M3() @ np.ones(n1)])
    K = la.KrylovSolver(op, ldu, {"Solver name": "GMRES", "Relative tolerance": 1e-8, "Maximum iterations": 200, "Restart size": 50})
    x = np.zeros(n0 + n1)
    K.Mult(rhs, x)
    xo = np.zeros(n0 + n1)

    def oprec(rr, zz):
        zz[:] = 0.0
        oldu.Mult(rr, zz)
    ref = orc.mfem_gmres(lambda v: A @ v, oprec, rhs, xo, rel_tol=1e-8, max_iter=200, m=50)
    assert K.GetConverged() and ref.converged and abs(K.GetNumIterations() - ref.final_iter) <= 1 and ref.final_iter < 60
    assert close(x, xo, 1e-6)
    # against Jacobi-class inner inverses
    jM = la.HypreSmootherWrapper(op.GetBlock(0, 0), 1)
    jS = la.HypreSmootherWrapper(la.SchurComplement(op, "DIAGONAL", 1.0, False), 1)
    lduj = la.Block2x2LDUInverseOperator(op, jM, jM, jM, jS)
    lduj.iterative_mode = False
    Kj = la.KrylovSolver(op, lduj, {"Solver name": "GMRES", "Relative tolerance": 1e-8, "Maximum iterations": 400, "Restart size": 50})
    xj = np.zeros(n0 + n1)
    Kj.Mult(rhs, xj)
    assert Kj.GetNumIterations() > 2 * K.GetNumIterations()
    with pytest.raises(la._lib.ElagHipError, match="preconditioner"):
        iS.iterative_mode = True
        iS.Mult(r[n0:], y)


def test_default_darcy_solver_graph_converges_at_full_size(la):
    """BASELINE configs[3] at cube456 x3 (473 536 + 233 472 dofs): GMRES(50) preconditioned by the reference's SPE10 solver graph -- Block LDU
    with AMGe V-cycles on M and on B diag(M)^-1 B^T (la.darcy_ldu_amge_preconditioner) -- converges to the example's 1e-6 within its 300
    iterations (the blocked cycle with Jacobi-class block smoothers does not: profiles/r02_bench_darcy.json), also with a heterogeneous
    coefficient (the bench's log-normal field, contrast 2e5)."""
    import scipy.sparse as sp
    from parelag_amd import meshgen
    for kinv_sigma in (None, 2.0):
        kinv = meshgen.lognormal_inverse_permeability(meshgen.Sequence.cube456(3), kinv_sigma) if kinv_sigma else None
        dar = meshgen.darcy_problem(3, kinv=kinv)
        n0, n1 = dar["M"].shape[0], dar["B"].shape[0]
        bop = la.MfemBlockOperator([0, n0, n0 + n1])
        for (i, j), M in {(0, 0): dar["M"], (0, 1): dar["Bt"], (1, 0): dar["B"]}.items():
            bop.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(M))
        ldu = la.darcy_ldu_amge_preconditioner(bop, dar["M"], dar["B"], dar["Ps"])
        K = la.KrylovSolver(bop, ldu, {"Solver name": "GMRES", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 300,
                                       "Restart size": 50})
        b = np.concatenate([np.zeros(n0), dar["seq"].meshes[0].M3() @ np.ones(n1)])
        x = np.zeros(n0 + n1)
        K.Mult(b, x)
        assert K.GetConverged() and K.GetNumIterations() <= 60, (kinv_sigma, K.GetNumIterations())
        A = sp.bmat([[dar["M"], dar["Bt"]], [dar["B"], None]]).tocsr()
        assert np.linalg.norm(A @ x - b) <= 1e-4 * np.linalg.norm(b)
        if kinv_sigma is None:
            # at this size the M hierarchy is renumbered and adds its result in place: du~ is computed where du belongs and the third
            # application adds A_3^-1 (-A01 dp) to it (blocked.hip: fold).  With a vector pass for the sign and one for the subtraction
            # (ELAG_HIP_LDU_FOLD=0): the same application to the last bit, the same solve
            with _env(ELAG_HIP_LDU_FOLD="0"):
                ldu0 = la.darcy_ldu_amge_preconditioner(bop, dar["M"], dar["B"], dar["Ps"])
            rng = np.random.default_rng(5)
            r = rng.uniform(-1, 1, n0 + n1)
            z, z0 = np.full(n0 + n1, 3.0), np.full(n0 + n1, 4.0)
            ldu.iterative_mode = ldu0.iterative_mode = False
            ldu.Mult(r, z)
            ldu0.Mult(r, z0)
            assert np.array_equal(z, z0) and np.linalg.norm(z) > 0
            K0 = la.KrylovSolver(bop, ldu0, {"Solver name": "GMRES", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 300,
                                             "Restart size": 50})
            x0 = np.zeros(n0 + n1)
            K0.Mult(b, x0)
            assert K0.GetNumIterations() == K.GetNumIterations() and np.array_equal(x, x0)


@pytest.mark.parametrize("n", [3, 8, 9, 150, 1471])
def test_blocked_dense_lu_is_the_column_by_column_one(la, n):
    """The coarsest saddle-point operator is inverted in place in blocks of 8 pivots (dense.hip, gj_block_*_kernel: two launches per block, the
    columns put back by one gather) where the column-by-column Gauss-Jordan on [A | I] (ELAG_HIP_LU_BLOCKED=0) takes two launches per pivot.
    Same partial pivoting (the pivots are found by eliminating the panel's own columns), other rounding: both agree with a host solve on an
    indefinite saddle-point matrix [[M, B^T], [B, 0]] -- zero diagonal block, so rows must be swapped -- at sizes with a partial last block and
    the 1 471 dofs of the cube456 x3 Darcy hierarchy."""
    import os
    rng = np.random.default_rng(n)
    n1 = max(1, (2 * n) // 3)
    n0 = n - n1
    Mb = sp.diags(rng.uniform(1.0, 2.0, n1)) + 0.1 * sp.random(n1, n1, density=min(1.0, 4.0 / n1), random_state=n)
    Mb = (Mb + Mb.T).tocsr()
    if n0:
        Bb = sp.random(n0, n1, density=min(1.0, 6.0 / n1), random_state=n + 1, format="lil")
        for i in range(n0):
            Bb[i, (2 * i) % n1] = 1.0 + 0.1 * i / n0               # full row rank
            Bb[i, (2 * i + 1) % n1] = -1.0
        A = sp.bmat([[None, Bb.tocsr()], [Bb.T.tocsr(), Mb]]).tocsr()       # the ZERO block first: the leading pivots must come from below
    else:
        A = Mb
    A = orc.csr(A)
    b = rng.uniform(-1, 1, n)
    xs = []
    for env in ({}, {"ELAG_HIP_LU_BLOCKED": "0"}, {"ELAG_HIP_LU_LOOKAHEAD": "0"}):
        saved_env = {k: os.environ.get(k) for k in env}      # (a variable the run itself was started with is put back, not dropped)
        os.environ.update(env)
        try:
            C = la.CoarseSolver(la.ParallelCSRMatrix.from_scipy(A), kind=1)
        finally:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        x = np.zeros(n)
        C.Mult(b, x)
        xs.append(x)
    ref = np.linalg.solve(A.toarray(), b)
    scale = np.max(np.abs(ref)) * max(1.0, np.linalg.cond(A.toarray()) * 1e-4)
    for x in xs:
        assert np.max(np.abs(x - ref)) <= 1e-10 * scale
    assert np.max(np.abs(xs[0] - xs[1])) <= 1e-10 * scale
    # the blocked form with a step's update and the next step's panel in one launch (gj_block_step_kernel, the default) and with a launch each
    # (ELAG_HIP_LU_LOOKAHEAD=0): the same operations, the same inverse to the last bit
    assert np.array_equal(xs[0], xs[2])


@pytest.mark.parametrize("n", [1, 2, 7, 8, 50, 97, 140, 141])
def test_spd_inverse_in_registers_is_the_lds_sweep_bit_for_bit(la, n):
    """The coarsest SPD operator of the single-form hierarchies (141 dofs at cube456 x3) is inverted by one workgroup: spd_inverse_reg_kernel keeps the
    matrix in registers during the elimination, spd_inverse_lds_kernel (ELAG_HIP_SPD_REG=0) sweeps it in LDS.  Entry by entry the same
    operations in the same order: the two inverses applied to a vector agree to the last bit, and with a host solve; repeated columns of a row
    add up in storage order either way."""
    import os
    rng = np.random.default_rng(100 + n)
    R = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=n, format="csr")
    A = (R @ R.T + sp.diags(rng.uniform(1.0, 2.0, n))).tocsr()
    A = orc.csr(A)
    b = rng.uniform(-1, 1, n)
    xs = []
    for env in ({}, {"ELAG_HIP_SPD_REG": "0"}):
        saved_env = {k: os.environ.get(k) for k in env}      # (a variable the run itself was started with is put back, not dropped)
        os.environ.update(env)
        try:
            C = la.CoarseSolver(la.ParallelCSRMatrix.from_scipy(A), kind=0)
        finally:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        x = np.zeros(n)
        C.Mult(b, x)
        xs.append(x)
    ref = np.linalg.solve(A.toarray(), b)
    assert np.max(np.abs(xs[0] - ref)) <= 1e-12 * np.max(np.abs(ref)) * np.linalg.cond(A.toarray())
    assert np.array_equal(xs[0], xs[1])


def test_coarsest_inverse_beside_the_build_is_the_same_hierarchy(la, darcy_small):
    """The coarsest operator's dense inverse -- the blocked LU of the saddle-point operator, the one-workgroup inverse of an SPD operator -- runs on
    the second stream beside the rest of a build (elag_hip_coarse_s::async_stream, DenseJob in kernels.hpp) and Finalize waits for it where the
    dense tail is planned; with ELAG_HIP_COARSE_ASYNC=0 it runs where it is created.  The same launches either way: the cycle of the blocked
    Darcy hierarchy and of an H1 hierarchy are the same bits, build after build (the job's scratch and the build's are disjoint)."""
    from parelag_amd import meshgen
    d = darcy_small
    rng = np.random.default_rng(3)
    n = d["off"][-1]
    b = rng.uniform(-1, 1, n)
    A1, Ps1, _ = meshgen.h1_problem(2)
    dA1 = la.ParallelCSRMatrix.from_scipy(A1); dPs1 = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps1]
    b1 = rng.uniform(-1, 1, A1.shape[0])
    outs = {}
    for mode in ("1", "0", "1", "1"):
        with _env(ELAG_HIP_COARSE_ASYNC=mode):
            H, _op = device_blocked_hierarchy(la, d, True, block_smoother="Block Jacobi")
            H1 = la.buildHierarchyFromDeRhamSequence(dA1, dPs1, "L1 Jacobi")
        x, x1 = np.zeros(n), np.zeros(A1.shape[0])
        H.Mult(b, x); H1.Mult(b1, x1)
        outs.setdefault(mode, []).append((x, x1))
    ref = outs["0"][0]
    assert np.linalg.norm(ref[0]) > 0 and np.linalg.norm(ref[1]) > 0
    for x, x1 in outs["1"]:
        assert np.array_equal(x, ref[0]) and np.array_equal(x1, ref[1])
    # a singular coarsest operator is still reported (by Finalize, which waits for the inverse)
    Z = sp.csr_matrix(np.array([[1., 2], [2, 4]]))
    with pytest.raises(la._lib.ElagHipError, match="singular"):
        la.CoarseSolver(la.ParallelCSRMatrix.from_scipy(orc.csr(Z)), kind=1)
