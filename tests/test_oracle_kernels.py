"""Validate the CPU oracle itself (it has no runnable reference behind it -- 'parity unpinned'):
dense/scipy recomputation, closed-form cases and the reference's algebraic invariants."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc


def rand_csr(n, m, density, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(n, m, density=density, random_state=rng, format="csr")
    A.data = rng.uniform(-1, 1, A.nnz)
    return orc.csr(A)


@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (-1.0, 1.0), (0.5, -2.0), (0.0, 3.0), (2.0, 2.0)])
def test_csr_matvec_and_transpose(alpha, beta):
    A = rand_csr(57, 43, 0.2, 1)
    rng = np.random.default_rng(2)
    x, y0 = rng.standard_normal(43), rng.standard_normal(57)
    y = y0.copy()
    orc.csr_matvec(A, alpha, x, beta, y)
    assert np.allclose(y, alpha * (A @ x) + beta * y0, rtol=1e-13, atol=1e-13)
    xt, z0 = rng.standard_normal(57), rng.standard_normal(43)
    z = z0.copy()
    orc.csr_matvecT(A, alpha, xt, beta, z)
    assert np.allclose(z, alpha * (A.T @ xt) + beta * z0, rtol=1e-13, atol=1e-13)


def test_empty_and_ragged_rows():
    A = orc.csr(sp.csr_matrix(([1.0, 2.0, 3.0], ([0, 0, 3], [0, 2, 1])), shape=(5, 3)))
    y = np.ones(5)
    orc.csr_matvec(A, 1.0, np.array([1.0, 1.0, 1.0]), 0.0, y)
    assert y.tolist() == [3.0, 0.0, 0.0, 3.0, 0.0]
    E = orc.csr(sp.csr_matrix((0, 0)))
    orc.csr_matvec(E, 1.0, np.empty(0), 0.0, np.empty(0))
    T = orc.transpose(A)
    assert (T != A.T).nnz == 0


def test_boolint_matvec():
    A = rand_csr(20, 20, 0.3, 3)
    x = np.arange(20, dtype=np.int32)
    y = np.zeros(20, dtype=np.int32)
    orc.csr_matvec_boolint(A, 1, x, 0, y)
    B = A.copy(); B.data[:] = 1
    assert np.array_equal(y, (B @ x).astype(np.int32))


def test_matmul_rap_add_vs_scipy():
    R = rand_csr(40, 15, 0.2, 4)
    A = rand_csr(40, 40, 0.15, 5)
    P = rand_csr(40, 15, 0.2, 6)
    assert abs(orc.matmul(A, P) - A @ P).max() < 1e-14
    C = orc.rap(R, A, P)
    assert abs(C - R.T @ A @ P).max() < 1e-13
    # square result: diagonal entry is stored first in every row (hypre RAP convention)
    assert all(C.indices[C.indptr[i]] == i for i in range(C.shape[0]))
    assert abs(orc.mfem_rap(R, A, P) - R.T @ A @ P).max() < 1e-13
    B = rand_csr(40, 40, 0.1, 7)
    S = orc.add(2.0, A, -3.0, B)
    assert abs(S - (2.0 * A - 3.0 * B)).max() < 1e-14
    # rectangular RAP (RT != P) does not force a diagonal
    P2 = rand_csr(40, 9, 0.2, 8)
    assert abs(orc.rap(R, A, P2) - R.T @ A @ P2).max() < 1e-13


def test_delete_zeros_sign_norms_compare():
    A = orc.csr(sp.csr_matrix(np.array([[1.0, 1e-12, 0.0], [-3.0, 0.0, 2.0], [0.0, 0.0, -1e-13]])))
    Z = orc.delete_zeros(A, 1e-9)
    assert Z.nnz == 3 and abs(Z - A).max() <= 1e-12
    S = orc.sign_transform(A)
    assert sorted(S.data.tolist()) == [-1.0, 0.0, 0.0, 1.0, 1.0]
    M = orc.OParCSR(A)
    assert M.norm_linf() == pytest.approx(5.0)
    assert M.norm_l1() == pytest.approx(4.0)
    assert M.norm_max() == 3.0
    assert M.norm_frobenius() == pytest.approx(np.sqrt(14.0 + 1e-24 + 1e-26))
    assert orc.compare(M, orc.OParCSR(A.copy()), 1e-12) == 0
    B = A.copy(); B.data[0] += 1e-3
    assert orc.compare(M, orc.OParCSR(B), 1e-6) == 64
    assert orc.compare(M, orc.OParCSR(orc.csr(sp.eye(4))), 1e-6) & 3 == 3


def test_fix_zero_rows():
    A = orc.diag_first(sp.csr_matrix(np.array([[2.0, -1.0, 0.0], [0.0, 0.0, 0.0], [0.0, -1.0, 2.0]])
                                     + sp.csr_matrix(([0.0, 0.0], ([1, 1], [1, 0])), shape=(3, 3))))
    A = orc.csr(sp.csr_matrix((np.array([2.0, -1.0, 0.0, 0.0, -1.0, 2.0]), np.array([0, 1, 1, 0, 1, 2]),
                               np.array([0, 2, 4, 6])), shape=(3, 3)))
    assert orc.fix_zero_rows(A) == 1
    assert np.allclose(A.toarray(), [[2, -1, 0], [0, 1, 0], [0, -1, 2]])


def test_l1_norm_options_and_sign_flip():
    A = orc.diag_first(sp.csr_matrix(np.array([[4.0, -1.0, 0.5], [1.0, -3.0, 0.0], [0.0, 2.0, 5.0]])))
    O = orc.csr(sp.csr_matrix(np.array([[0.25, 0.0], [0.0, -0.5], [0.0, 0.0]])))
    assert np.allclose(orc.l1_norms(A, O, 1), [5.75, -4.5, 7.0])
    assert np.allclose(orc.l1_norms(A, O, 2), [4.25, -3.5, 5.0])
    assert np.allclose(orc.l1_norms(A, O, 3), [4 ** 2 + 1 + 0.25 + 0.0625, -(1 + 9 + 0.25), 29.0])
    assert np.allclose(orc.l1_norms(A, O, 4), [4.0, -3.0, 5.0])


def test_jacobi_is_damped_richardson():
    A, Ps, _ = __import__("parelag_amd.meshgen", fromlist=["x"]).h1_problem(0)
    Ad = orc.diag_first(A)
    M = orc.OParCSR(Ad)
    s = orc.OSmoother(M, type=1, sweeps=2, weight=0.8)
    rng = np.random.default_rng(0)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    u = u0.copy()
    s.Mult(f, u)
    l1 = np.asarray(abs(A).sum(axis=1)).ravel()
    ref = u0.copy()
    for _ in range(2):
        ref = ref + 0.8 * (f - A @ ref) / l1
    assert np.allclose(u, ref, rtol=1e-13, atol=1e-14)
    s.iterative_mode = False
    s.Mult(f, u)
    ref = 0.8 * f / l1
    ref = ref + 0.8 * (f - A @ ref) / l1
    assert np.allclose(u, ref, rtol=1e-13, atol=1e-14)


def test_l1gs_matches_dense_loop():
    rng = np.random.default_rng(5)
    n = 30
    B = rng.standard_normal((n, n)); B[abs(B) < 1.0] = 0
    Ad = B + B.T + 10 * np.eye(n)
    A = orc.diag_first(sp.csr_matrix(Ad))
    s = orc.OSmoother(orc.OParCSR(A), type=2)
    f, u = rng.standard_normal(n), rng.standard_normal(n)
    ref = u.copy()
    l1 = abs(np.diag(Ad))
    for i in list(range(n)) + list(range(n - 1, -1, -1)):
        ref[i] += (f[i] - Ad[i] @ ref) / l1[i]
    s.Mult(f, u)
    assert np.allclose(u, ref, rtol=1e-12)


def test_l1gs_scaled_ssor_branch_matches_dense_loop_and_reduces_to_the_others():
    """hypre_ParCSRRelax 2 / 4 with relax_weight != 1: c1 = omega w, c2 = omega (1 - w), dif over the rows already visited
    in this direction against the vector the forward pass started from; w = 1 is the omega branch bit for bit"""
    rng = np.random.default_rng(6)
    n = 40
    B = rng.standard_normal((n, n)); B[abs(B) < 1.1] = 0
    Ad = B + B.T + 10 * np.eye(n)
    A = orc.diag_first(sp.csr_matrix(Ad))
    l1 = abs(np.diag(Ad))
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    for w, om in ((0.8, 1.0), (0.7, 1.3), (1.25, 0.9)):
        ref, v = u0.copy(), u0.copy()
        c1, c2 = om * w, om * (1 - w)
        for i in range(n):
            res = f[i] - Ad[i] @ ref
            dif = Ad[i, :i] @ (v[:i] - ref[:i])
            ref[i] += (c1 * res + c2 * dif) / l1[i]
        for i in range(n - 1, -1, -1):
            res = f[i] - Ad[i] @ ref
            dif = Ad[i, i + 1:] @ (v[i + 1:] - ref[i + 1:])
            ref[i] += (c1 * res + c2 * dif) / l1[i]
        u = u0.copy()
        orc.OSmoother(orc.OParCSR(A), type=2, weight=w, omega=om).Mult(f, u)
        assert np.allclose(u, ref, rtol=1e-12, atol=1e-13)
    vt = np.empty(n)
    ua, ub = u0.copy(), u0.copy()
    L = orc.lib()
    args = (orc.C.c_int(n), orc._ip(A.indptr), orc._ip(A.indices), orc._dp(A.data), None, None, None, orc._dp(f), orc._dp(l1), None)
    L.orc_l1gs_sweep_scaled(*args, orc.C.c_double(1.0), orc.C.c_double(1.3), orc._dp(ua), orc._dp(vt))
    L.orc_l1gs_sweep_omega(*args, orc.C.c_double(1.3), orc._dp(ub))
    assert np.array_equal(ua, ub) and np.array_equal(vt, u0)


def test_gs_weighted_branch_matches_dense_loop():
    """hypre_BoomerAMGRelax 6 with relax_weight != 1 or omega != 1: u_i = (1 - w omega) u_i + w (omega res + res0 + (1 - omega) res2) / a_ii with
    res0 over the current vector, res2 over the copy taken before the forward pass; (1, 1) falls back to plain symmetric Gauss-Seidel"""
    rng = np.random.default_rng(7)
    n = 40
    B = rng.standard_normal((n, n)); B[abs(B) < 1.1] = 0
    Ad = B + B.T + 10 * np.eye(n)
    A = orc.diag_first(sp.csr_matrix(Ad))
    off = Ad - np.diag(np.diag(Ad))
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    for w, om in ((1.0, 1.2), (0.8, 1.0), (0.9, 1.4)):
        ref, v = u0.copy(), u0.copy()
        for i in list(range(n)) + list(range(n - 1, -1, -1)):
            res0, res2 = -(off[i] @ ref), off[i] @ v
            ref[i] = (1 - w * om) * ref[i] + w * (om * f[i] + res0 + (1 - om) * res2) / Ad[i, i]
        u = u0.copy()
        orc.OSmoother(orc.OParCSR(A), type=6, weight=w, omega=om).Mult(f, u)
        assert np.allclose(u, ref, rtol=1e-12, atol=1e-13)
    # the formula at (1, 1) is the plain sweep (hypre takes the other branch there; the two agree to rounding)
    ua, ub, vt = u0.copy(), u0.copy(), np.empty(n)
    L = orc.lib()
    args = (orc.C.c_int(n), orc._ip(A.indptr), orc._ip(A.indices), orc._dp(A.data), None, None, None, orc._dp(f), None)
    L.orc_gs_sweep_sor(*args, orc.C.c_double(1.0), orc.C.c_double(1.0), orc._dp(ua), orc._dp(vt))
    L.orc_gs_sweep(*args, orc._dp(ub))
    assert np.allclose(ua, ub, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("order", [1, 2, 3, 4])
def test_cheby_coefs_identity(order):
    """1 - t s(t) = T_k((theta-t)/delta) / T_k(theta/delta)  (SURVEY A.3)."""
    mx, mn, fr = 1.7, 0.05, 0.3
    co, c = orc.cheby_coefs(mx, mn, fr, order)
    assert co == order - 1
    ub = 1.1 * mx; lb = (ub - mn) * fr + mn
    th, de = (ub + lb) / 2, (ub - lb) / 2
    T = np.polynomial.chebyshev.Chebyshev.basis(order)
    for t in np.linspace(0.01, ub, 17):
        s = sum(c[i] * t ** i for i in range(order))
        assert 1 - t * s == pytest.approx(T((th - t) / de) / T(th / de), rel=1e-10, abs=1e-12)


def test_cheby_sweep_is_polynomial_in_scaled_operator(h1_k1):
    A, _, _ = h1_k1
    Ad = orc.diag_first(A)
    M = orc.OParCSR(Ad)
    s = orc.OSmoother(M, type=16, order=3, fraction=0.3, eig=(1.8, 0.1))
    rng = np.random.default_rng(1)
    f, u0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    u = u0.copy(); s.Mult(f, u)
    d = A.diagonal(); ds = 1 / np.sqrt(d)
    At = sp.diags(ds) @ A @ sp.diags(ds)
    r = ds * (f - A @ u0)
    z = s.coefs[2] * r
    for i in (1, 0):
        z = s.coefs[i] * r + At @ z
    assert np.allclose(u, u0 + ds * z, rtol=1e-12, atol=1e-13)


def test_hypre_rng_known_values():
    # Park-Miller minimal standard: seed 1 -> 16807, 282475249, 1622650073, ...
    v = orc.hypre_random(3, 1)
    m = 2147483647
    assert np.allclose(v, 2 * np.array([16807, 282475249, 1622650073]) / m - 1, rtol=0, atol=1e-16)


def test_tql1_vs_numpy():
    rng = np.random.default_rng(3)
    d, e = rng.uniform(1, 2, 10), rng.uniform(0.1, 1, 10)
    T = np.diag(d) + np.diag(e[1:], 1) + np.diag(e[1:], -1)
    assert np.allclose(orc.tql1(d, e), np.linalg.eigvalsh(T), rtol=1e-12)


def test_eig_estimate_brackets_spectrum(h1_k1):
    A, _, _ = h1_k1
    M = orc.OParCSR(orc.diag_first(A))
    mx, mn = orc.max_eig_estimate_cg(M, 1, 10)
    d = A.diagonal(); ds = 1 / np.sqrt(d)
    ev = np.linalg.eigvalsh((sp.diags(ds) @ A @ sp.diags(ds)).toarray())
    assert 0.85 * ev[-1] <= mx <= ev[-1] * (1 + 1e-12)
    assert mn >= ev[0] * (1 - 1e-12)


def test_index_maps():
    p = orc.partition_refined(64, 8)
    assert np.array_equal(p, np.arange(64) // 8)
    q = orc.partition_refined_legacy(64, 8, 3)
    assert np.array_equal(q[:8], np.arange(8)) and np.array_equal(q[8:15], np.zeros(7))
    assert np.array_equal(np.bincount(q), np.full(8, 8))
    # connected components: part 0 = {0,1} U {4} (disconnected), part 1 = {2,3}
    conn = sp.csr_matrix(np.array([[1, 1, 0, 0, 0], [1, 1, 1, 0, 0], [0, 1, 1, 1, 0], [0, 0, 1, 1, 1], [0, 0, 0, 1, 1]]))
    n, newp = orc.connected_components(np.array([0, 0, 1, 1, 0]), conn)
    assert n == 3 and newp.tolist() == [0, 0, 2, 2, 1]
    I, J = orc.transpose_partitioning(np.array([1, 0, -1, 1, 0, 2]), 3)
    assert I.tolist() == [0, 2, 4, 5] and J.tolist() == [1, 4, 0, 3, 5]


def test_kaczmarz_matches_dense_loop_and_projects():
    """hypre_ParCSRRelax type 3 restated twice: the C loop against plain numpy on dense rows; one row step projects u onto the
    hyperplane a_i . u = f_i (omega = 1), so a consistent system's error never grows"""
    rng = np.random.default_rng(5)
    n = 40
    B = sp.random(n, n, density=0.2, random_state=rng, format="lil"); B.setdiag(rng.uniform(1, 2, n))
    A = orc.diag_first(B.tocsr())
    D = A.toarray()
    for omega in (1.0, 0.7):
        f, u = rng.standard_normal(n), rng.standard_normal(n)
        ur = u.copy()
        orc.OSmoother(orc.OParCSR(A), type=3, omega=omega).Mult(f, u)
        for i in list(range(n)) + list(range(n - 1, -1, -1)):
            res = f[i]
            for k in range(A.indptr[i], A.indptr[i + 1]):
                res -= A.data[k] * ur[A.indices[k]]
            res /= np.sum(A.data[A.indptr[i]:A.indptr[i + 1]] ** 2)
            for k in range(A.indptr[i], A.indptr[i + 1]):
                ur[A.indices[k]] += omega * res * A.data[k]
        assert np.allclose(u, ur, rtol=1e-13, atol=1e-13)
    x = rng.standard_normal(n); b = D @ x
    u = np.zeros(n); e = np.linalg.norm(x)
    S = orc.OSmoother(orc.OParCSR(A), type=3)
    for _ in range(4):
        S.Mult(b, u)
        assert np.linalg.norm(x - u) < e
        e = np.linalg.norm(x - u)


def test_taubin_and_fir_are_the_polynomials_they_claim(h1_k1):
    """types 1001 / 1002 (mfem::HypreSmoother): with f = 0 a sweep is a polynomial in t = A / max_eig applied to u -- Taubin's
    prod (1 - c_i t), FIR's sum_k c_k T_k(1 - t) with the windowed Chebyshev coefficients of the low-pass with pass band
    acos(1 - fraction max_eig / 2)"""
    A = h1_k1[0]
    Ad = orc.diag_first(A)
    lam, V = np.linalg.eigh(A.toarray())
    rng = np.random.default_rng(8)
    u0 = rng.standard_normal(A.shape[0])
    S = orc.OSmoother(orc.OParCSR(Ad), type=1001)
    assert 0.3 * lam[-1] <= 2 * S.max_eig <= 1.02 * lam[-1]
    u = u0.copy(); S.Mult(np.zeros_like(u0), u)
    t = lam / S.max_eig
    p = np.ones_like(t)
    for i in range(40):
        p *= 1.0 - (0.5 if i % 2 == 0 else -0.5) * t
    assert np.allclose(u, V @ (p * (V.T @ u0)), rtol=1e-9, atol=1e-9 * np.abs(u0).max())
    for order in (1, 2, 5):
        S = orc.OSmoother(orc.OParCSR(Ad), type=1002, order=order, fraction=0.3)
        theta = np.arccos(1.0 - 0.5 * 0.3 * S.max_eig)
        k = np.arange(order + 1)
        win = 0.42 + 0.50 * np.cos(k * np.pi / (order + 1)) + 0.08 * np.cos(2 * k * np.pi / (order + 1))
        cheb = np.r_[theta / np.pi, 2 * np.sin(k[1:] * theta) / (k[1:] * np.pi)]
        assert np.allclose(S.fir, win * cheb, rtol=1e-14)
        u = u0.copy(); S.Mult(np.zeros_like(u0), u)
        t = lam / S.max_eig
        Tk = [np.ones_like(t), 1.0 - t]                                  # x1 = x0 - A x0 / m; then x2 = 2 x1 - x0 - 2 A x1 / m = 2 (1 - t) x1 - x0
        for _ in range(2, order + 1):
            Tk.append(2 * (1.0 - t) * Tk[-1] - Tk[-2])
        p = sum(S.fir[i] * Tk[i] for i in range(order + 1))
        assert np.allclose(u, V @ (p * (V.T @ u0)), rtol=1e-9, atol=1e-9 * np.abs(u0).max())
