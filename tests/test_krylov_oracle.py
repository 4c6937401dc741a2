"""CPU checks of the oracle's restatement of mfem's Krylov iterations against scipy / closed forms."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import oracle as orc


def spd(n, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=0.05, random_state=rng, format="csr")
    A = A + A.T + sp.diags(np.full(n, 4.0 + abs(A).sum(axis=1).max()))
    return sp.csr_matrix(A)


def test_cg_matches_textbook_pcg_and_reports_mfem_statistics():
    A = spd(200, 1)
    dinv = 1.0 / A.diagonal()
    b = np.random.default_rng(2).uniform(-1, 1, 200)
    x = np.full(200, 9.0)
    res = orc.mfem_cg(lambda v: A @ v, lambda r, z: z.__setitem__(slice(None), dinv * r), b, x, rel_tol=1e-10, max_iter=100)
    assert res.converged and 0 < res.final_iter < 100
    assert np.linalg.norm(A @ x - b) <= 1e-8 * np.linalg.norm(b)
    # final_norm = sqrt((B r, r)) of the last residual; the convergence test is (B r, r) <= rel^2 (B r0, r0)
    r = b - A @ x
    assert np.isclose(res.final_norm, np.sqrt(r @ (dinv * r)), rtol=1e-6)
    assert res.final_norm <= 1e-10 * np.sqrt(b @ (dinv * b))
    # one fewer iteration is not enough
    x2 = np.zeros(200)
    res2 = orc.mfem_cg(lambda v: A @ v, lambda r, z: z.__setitem__(slice(None), dinv * r), b, x2, rel_tol=1e-10, max_iter=res.final_iter - 1)
    assert not res2.converged and res2.final_iter == res.final_iter - 1
    # iterative_mode starts from the given x
    x3 = x.copy()
    res3 = orc.mfem_cg(lambda v: A @ v, None, b, x3, iterative_mode=True, abs_tol=1e-6, max_iter=50)
    assert res3.converged and res3.final_iter <= 1


def test_gmres_and_fgmres_minimise_the_residual_over_the_krylov_space():
    rng = np.random.default_rng(3)
    n = 120
    A = sp.csr_matrix(sp.random(n, n, density=0.08, random_state=rng) + sp.diags(np.full(n, 3.0)))
    b = rng.uniform(-1, 1, n)
    for flexible in (False, True):
        x = np.zeros(n)
        res = orc.mfem_gmres(lambda v: A @ v, None, b, x, rel_tol=1e-12, max_iter=200, m=30, flexible=flexible)
        assert res.converged
        assert np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b)
    # k steps of GMRES from zero = least squares over span{b, Ab, ...}
    k = 7
    x = np.zeros(n)
    res = orc.mfem_gmres(lambda v: A @ v, None, b, x, rel_tol=0.0, max_iter=k, m=50)
    K = np.stack([np.linalg.matrix_power(A.toarray(), p) @ b for p in range(k)], axis=1)
    y, *_ = np.linalg.lstsq(A.toarray() @ K, b, rcond=None)
    assert np.allclose(x, K @ y, rtol=1e-8, atol=1e-10)
    assert not res.converged and res.final_iter == k
    assert np.isclose(res.final_norm, np.linalg.norm(b - A @ x), rtol=1e-8)
    # left preconditioning: the monitored norm is ||M (b - A x)||
    dinv = 1.0 / A.diagonal()
    x = np.zeros(n)
    res = orc.mfem_gmres(lambda v: A @ v, lambda r, z: z.__setitem__(slice(None), dinv * r), b, x, rel_tol=1e-9, max_iter=100, m=10)
    assert res.converged and np.linalg.norm(dinv * (b - A @ x)) <= 1.0001e-9 * np.linalg.norm(dinv * b)


def test_stationary_solver_is_preconditioned_richardson():
    A = spd(80, 5)
    dinv = 1.0 / A.diagonal()
    b = np.random.default_rng(6).uniform(-1, 1, 80)
    x = np.zeros(80)
    res = orc.stationary_solver(lambda v: A @ v, lambda r, c: c.__setitem__(slice(None), dinv * r), b, x, max_iter=3)
    xr = np.zeros(80)
    for _ in range(3):
        xr += dinv * (b - A @ xr)
    assert np.allclose(x, xr, rtol=1e-13, atol=1e-15) and res.final_iter == 3 and not res.converged
    x = np.zeros(80)
    res = orc.stationary_solver(lambda v: A @ v, lambda r, c: c.__setitem__(slice(None), dinv * r), b, x, rel_tol=1e-3, max_iter=500)
    assert res.converged and np.linalg.norm(b - A @ x) < 1e-3 * np.linalg.norm(b)
