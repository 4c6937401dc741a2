"""CPU checks of the oracle's blocked (saddle-point) layer against closed-form dense statements of the reference's
algorithms and against the explicit 6x6 fixture of src/linalg/unit_test/block2x2_test.cpp."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from tests.helpers import BLOCK2X2_FIXTURE as FX, oracle_block_solver, oracle_blocked_hierarchy


def fixture_blocks():
    return {(0, 0): FX["M"], (0, 1): FX["Bt"], (1, 0): FX["B"], (1, 1): FX["C"]}


def l1(M):
    return np.asarray(abs(M).sum(axis=1)).ravel() * np.sign(M.diagonal())


def test_block_operator_mult_on_the_block2x2_fixture():
    op = orc.OBlockOperator([0, 4, 6], [0, 4, 6], {k: orc.OParCSR(orc.csr(v)) for k, v in fixture_blocks().items()})
    r = FX["b"].copy()
    op.matvec(-1.0, FX["x0"], 1.0, r)              # the residual block2x2_test.cpp:197-204 prints
    assert np.array_equal(r, FX["residual"])
    dense = sp.bmat([[FX["M"], FX["Bt"]], [FX["B"], FX["C"]]]).toarray()
    y = np.zeros(6)
    op.matvecT(1.0, FX["x0"], 0.0, y)
    assert np.allclose(y, dense.T @ FX["x0"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("s_type", ["DIAGONAL", "ABSROWSUM"])
@pytest.mark.parametrize("with_a11", [True, False])
def test_schur_complement_formula(s_type, with_a11):
    M, Bt, B, Cm = (FX[k].toarray() for k in ("M", "Bt", "B", "C"))
    d = np.diag(M) if s_type == "DIAGONAL" else np.abs(M).sum(axis=1)
    ref = (Cm if with_a11 else 0.0) - 0.7 * B @ np.diag(1.0 / d) @ Bt
    S = orc.schur_complement(FX["M"], FX["Bt"], FX["B"], FX["C"] if with_a11 else None, s_type, 0.7)
    assert np.allclose(S.toarray(), ref, rtol=0, atol=1e-15)


@pytest.mark.parametrize("kind", ["Block Jacobi", "Block GS Lower", "Block GS Upper"])
@pytest.mark.parametrize("iterative", [True, False])
def test_block_jacobi_and_gs_closed_forms(kind, iterative):
    """One l1-Jacobi sweep from zero in each block => the block solves are diagonal scalings W0, W1."""
    sol, op, S = oracle_block_solver(fixture_blocks(), [0, 4, 6], kind=kind)
    sol.iterative_mode = iterative
    M, Bt, B, Cm = (FX[k].toarray() for k in ("M", "Bt", "B", "C"))
    Sd = -(Cm - B @ np.diag(1.0 / np.diag(M)) @ Bt)                 # "Use Negative S"
    assert np.allclose(S.toarray(), Sd, atol=1e-15)
    W0, W1 = 1.0 / l1(FX["M"]), 1.0 / l1(sp.csr_matrix(Sd))
    dense = np.block([[M, Bt], [B, Cm]])
    x0 = FX["x0"] if iterative else np.zeros(6)
    r = FX["b"] - dense @ x0
    if kind == "Block Jacobi":
        c0, c1 = W0 * r[:4], W1 * r[4:]
    elif kind == "Block GS Lower":
        c0 = W0 * r[:4]; c1 = W1 * (r[4:] - B @ c0)
    else:
        c1 = W1 * r[4:]; c0 = W0 * (r[:4] - Bt @ c1)
    x = FX["x0"].copy()
    sol.Mult(FX["b"], x)
    assert np.allclose(x, x0 + np.concatenate([c0, c1]), rtol=1e-14, atol=1e-14)
    if kind != "Block Jacobi":                                       # transposes: (L)^T sweep uses A_ji^T, other direction
        xt = FX["x0"].copy()
        sol.MultTranspose(FX["b"], xt)
        if kind == "Block GS Lower":
            t1 = W1 * r[4:]; t0 = W0 * (r[:4] - B.T @ t1)
        else:
            t0 = W0 * r[:4]; t1 = W1 * (r[4:] - Bt.T @ t0)
        assert np.allclose(xt, x0 + np.concatenate([t0, t1]), rtol=1e-14, atol=1e-14)


def test_block_ldu_closed_form():
    sol, op, S = oracle_block_solver(fixture_blocks(), [0, 4, 6], kind="Block LDU", damping=0.9)
    M, Bt, B, Cm = (FX[k].toarray() for k in ("M", "Bt", "B", "C"))
    W0, W1 = 1.0 / l1(FX["M"]), 1.0 / l1(S)
    r = FX["b"].copy()
    dut = W0 * r[:4]
    dp = W1 * (r[4:] - B @ dut)
    du = W0 * r[:4] - W0 * (Bt @ dp)
    x = np.full(6, 3.0)
    sol.Mult(FX["b"], x)                                             # preconditioner mode: initial guess ignored
    assert np.allclose(x, 0.9 * np.concatenate([du, dp]), rtol=1e-14, atol=1e-14)


def test_blocked_two_level_cycle_equals_dense_statement():
    from parelag_amd import meshgen
    d = meshgen.darcy_problem(0, seq=meshgen.Sequence.kuhn_cube(1, 1))
    blocks = {(0, 0): d["M"], (0, 1): d["Bt"], (1, 0): d["B"]}
    H, lev = oracle_blocked_hierarchy(blocks, d["Ps"], kind="Block Jacobi")
    M, Bt, B = d["M"].toarray(), d["Bt"].toarray(), d["B"].toarray()
    A = np.block([[M, Bt], [B, np.zeros((B.shape[0], B.shape[0]))]])
    P = sp.block_diag(d["Ps"][0]).toarray()
    Ac = P.T @ A @ P
    S = B @ np.diag(1.0 / np.diag(M)) @ Bt
    W = np.concatenate([1.0 / np.abs(M).sum(axis=1), 1.0 / np.abs(S).sum(axis=1)])
    f = np.random.default_rng(3).uniform(-1, 1, A.shape[0])
    x = W * f
    x += P @ np.linalg.solve(Ac, P.T @ (f - A @ x))
    x += W * (f - A @ x)
    y = np.full_like(f, 5.0)
    H.preconditioner = True
    H.Mult(f, y)
    assert np.max(np.abs(y - x)) <= 1e-11 * np.max(np.abs(x))
    # coarse blocks: Galerkin products, zero block stays absent
    assert (1, 1) not in lev[1]
    assert np.allclose(lev[1][(1, 0)].toarray(), (P.T @ A @ P)[lev[1][(0, 0)].shape[0]:, :lev[1][(0, 0)].shape[0]], atol=1e-13)
