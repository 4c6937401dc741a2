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


# ---- BiCGSTAB / MINRES (KrylovSolver.cpp:39-42) ------------------------------------------------------------------------------

def _sym_indefinite(n=60, seed=0):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.r_[np.linspace(1, 5, 2 * n // 3), -np.linspace(1, 3, n - 2 * n // 3)]
    return (Q * lam) @ Q.T, rng.standard_normal(n)


def test_minres_minimises_the_residual_over_the_krylov_space():
    A, b = _sym_indefinite()
    for k in (1, 2, 3, 5):
        x = np.zeros_like(b)
        res = orc.mfem_minres(lambda v: A @ v, None, b, x, max_iter=k)
        assert not res.converged and res.final_iter == k
        Q, _ = np.linalg.qr(np.stack([np.linalg.matrix_power(A, i) @ b for i in range(k)], 1))   # orthonormal basis of K_k
        y = np.linalg.lstsq(A @ Q, b, rcond=None)[0]
        assert np.linalg.norm(x - Q @ y) <= 1e-10 * np.linalg.norm(Q @ y)
        assert np.isclose(res.final_norm, np.linalg.norm(b - A @ x), rtol=1e-10)                 # |eta| is the residual norm


def test_minres_converges_with_and_without_spd_preconditioner():
    A, b = _sym_indefinite()
    x = np.zeros_like(b)
    res = orc.mfem_minres(lambda v: A @ v, None, b, x, rel_tol=1e-10, max_iter=300)
    assert res.converged and np.linalg.norm(A @ x - b) <= 1e-9 * np.linalg.norm(b)
    d = 1.0 / (np.abs(np.diag(A)) + 1.0)
    xp = np.zeros_like(b)
    resp = orc.mfem_minres(lambda v: A @ v, lambda r, z: z.__setitem__(slice(None), d * r), b, xp, rel_tol=1e-10, max_iter=400)
    assert resp.converged and np.linalg.norm(A @ xp - b) <= 1e-7 * np.linalg.norm(b)
    # preconditioned MINRES = MINRES on the symmetrically scaled system
    s = np.sqrt(d)
    xs = np.zeros_like(b)
    orc.mfem_minres(lambda v: s * (A @ (s * v)), None, s * b, xs, max_iter=7)
    x7 = np.zeros_like(b)
    orc.mfem_minres(lambda v: A @ v, lambda r, z: z.__setitem__(slice(None), d * r), b, x7, max_iter=7)
    assert np.linalg.norm(x7 - s * xs) <= 1e-10 * np.linalg.norm(x7)
    # iterative_mode continues from the given iterate; zero right-hand side stops at once
    x0 = np.random.default_rng(9).standard_normal(b.size)
    r0 = np.linalg.norm(b - A @ x0)
    res2 = orc.mfem_minres(lambda v: A @ v, None, b, x0, iterative_mode=True, rel_tol=1e-6, max_iter=300)
    assert res2.converged and np.linalg.norm(b - A @ x0) <= 1.0001e-6 * r0
    assert orc.mfem_minres(lambda v: A @ v, None, np.zeros_like(b), np.ones_like(b), max_iter=5).final_iter == 0


def test_bicgstab_matches_its_polynomial_characterisation_and_converges():
    rng = np.random.default_rng(2)
    n = 50
    B = rng.standard_normal((n, n)) + 9 * np.eye(n)
    b = rng.standard_normal(n)
    x = np.zeros(n)
    res = orc.mfem_bicgstab(lambda v: B @ v, None, b, x, rel_tol=1e-11, max_iter=200)
    assert res.converged and 5 < res.final_iter < 100
    assert np.linalg.norm(B @ x - b) <= 1e-9 * np.linalg.norm(b)
    # after one step x = alpha r0 + omega s with the textbook alpha / omega
    x1 = np.zeros(n)
    orc.mfem_bicgstab(lambda v: B @ v, None, b, x1, max_iter=1)
    v = B @ b
    alpha = (b @ b) / (b @ v)
    s = b - alpha * v
    t = B @ s
    assert np.allclose(x1, alpha * b + ((t @ s) / (t @ t)) * s, rtol=1e-13, atol=1e-14)
    # right preconditioning by the exact inverse converges in one step
    Binv = np.linalg.inv(B)
    xp = np.zeros(n)
    resp = orc.mfem_bicgstab(lambda v: B @ v, lambda r, z: z.__setitem__(slice(None), Binv @ r), b, xp, rel_tol=1e-8, max_iter=10)
    assert resp.converged and resp.final_iter == 1 and np.linalg.norm(B @ xp - b) <= 1e-10 * np.linalg.norm(b)
    # breakdown: rho = 0 is reported as not converged at that iteration
    e0 = np.zeros(4); e0[0] = 1.0
    J = np.roll(np.eye(4), 1, axis=0)                      # r . (J r) = 0 for r = e0: alpha = inf, next rho degenerate
    xb = np.zeros(4)
    resb = orc.mfem_bicgstab(lambda v: J @ v, None, e0, xb, rel_tol=1e-12, max_iter=5)
    assert not resb.converged
