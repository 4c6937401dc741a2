"""Shared test helpers: build oracle-side hierarchies the way AMGeSolverFactory/Hierarchy do."""
import numpy as np
import scipy.sparse as sp

from oracle import oracle as orc


def oracle_h1_hierarchy(A, Ps, smoother=1, sweeps=1, weight=1.0, order=2, fraction=0.3, eigs=None):
    """buildHierarchyFromDeRhamSequence (ParELAG_Hierarchy.cpp:282-383): A_{l+1} = RAP(A_l, P_l) followed by
    hypre_ParCSRMatrixFixZeroRows (essential BCs), smoothers per AMGeSolverFactory.cpp:108-142, and a
    dense Cholesky coarse solve.  Returns dict with As (diag-first csr), Ps, smoother dicts, chol."""
    As = [orc.diag_first(A)]
    Pl = [orc.csr(P) for P in Ps]
    for P in Pl:
        Ac = orc.rap(P, As[-1], P)
        orc.fix_zero_rows(Ac)
        As.append(Ac)
    sms = []
    for l, Al in enumerate(As):
        d = dict(type=smoother, sweeps=sweeps, weight=weight, order=order, fraction=fraction)
        if smoother in (1, 2, 4):
            d["l1"] = orc.l1_norms(Al, None, option=smoother)
        if smoother == 16:
            M = orc.OParCSR(Al)
            mx, mn = eigs[l] if eigs is not None else orc.max_eig_estimate_cg(M, 1, 10)
            d["max_eig"], d["min_eig"] = mx, mn
            d["ds"] = 1.0 / np.sqrt(Al.data[Al.indptr[:-1]])
        sms.append(d)
    chol = orc.chol_factor(As[-1].toarray())
    return dict(As=As, Ps=Pl, sms=sms, chol=chol)


def oracle_py_hierarchy(h):
    """OHierarchy (python-composed, exchange-capable) from the pieces above."""
    levels = []
    nl = len(h["As"])
    for l in range(nl):
        A = orc.OParCSR(h["As"][l])
        lev = dict(A=A, P=orc.OParCSR(h["Ps"][l - 1]) if l > 0 else None)
        if l < nl - 1:
            s = h["sms"][l]
            eig = (s["max_eig"], s["min_eig"]) if s["type"] == 16 else None
            mk = lambda: orc.OSmoother(A, type=s["type"], sweeps=s["sweeps"], weight=s["weight"],
                                       order=s["order"], fraction=s["fraction"], eig=eig)
            lev["pre"], lev["post"] = mk(), mk()
        else:
            Lf = h["chol"]
            lev["coarse"] = lambda rhs, sol, Lf=Lf: orc.chol_solve(Lf, rhs, sol)
        levels.append(lev)
    return orc.OHierarchy(levels)


def pcg(A, M_apply, b, rtol=1e-10, maxit=200):
    """Plain preconditioned CG (mfem::CGSolver stand-in for tests). Returns x, iterations, residual history."""
    x = np.zeros_like(b)
    r = b - A @ x
    z = np.zeros_like(b); M_apply(r, z)
    p = z.copy()
    rz = r @ z
    hist = [np.sqrt(r @ r)]
    for it in range(1, maxit + 1):
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p
        r -= a * Ap
        hist.append(np.sqrt(r @ r))
        if hist[-1] <= rtol * hist[0]:
            return x, it, hist
        z = np.zeros_like(b); M_apply(r, z)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, maxit, hist


def oracle_hdiv_hierarchy(A, Ps, Ds, primary=1, aux=1):
    """2-form hierarchy with HiptmairSmoother(primary, auxiliary) on every non-coarsest level
    (Create2FormParameterList.hpp:78-133 with the relaxation types swapped for GPU-capable ones):
    A_aux = RAP(A, D) + FixZeroRows per HiptmairSmootherFactory.cpp:143-165."""
    As = [orc.diag_first(A)]
    Pl = [orc.csr(P) for P in Ps]
    for P in Pl:
        Ac = orc.rap(P, As[-1], P)
        orc.fix_zero_rows(Ac)
        As.append(Ac)
    sms, hipt = [], []
    for l, Al in enumerate(As):
        sms.append(dict(type=primary, l1=orc.l1_norms(Al, None, option=primary)))
        if l < len(As) - 1:
            D = orc.csr(Ds[l])
            X = orc.rap(D, Al, D)
            orc.fix_zero_rows(X)
            hipt.append(dict(D=D, Aaux=X, sm=dict(type=aux, l1=orc.l1_norms(X, None, option=aux))))
        else:
            hipt.append(None)
    chol = orc.chol_factor(As[-1].toarray())
    return dict(As=As, Ps=Pl, sms=sms, hipt=hipt, chol=chol)


# ---- blocked (Darcy) hierarchies ----------------------------------------------------------------------------------

BLOCK2X2_FIXTURE = dict(  # the explicit 6x6 system of src/linalg/unit_test/block2x2_test.cpp:14-34,98-131,150-157
    M=sp.csr_matrix(np.array([[2., 1, 0, 0], [3, 4, 3, 0], [0, 3, 4, 3], [0, 0, 1, 2]])),
    Bt=sp.csr_matrix(np.array([[1., 0], [1, 1], [1, 1], [0, 1]])),
    B=sp.csr_matrix(np.array([[2., 1, 2, 0], [0, 2, 1, 2]])),
    C=sp.csr_matrix(np.array([[3., 0], [1, 3]])),
    b=np.array([2., 1, 3, 1, 4, 1]), x0=np.array([1., 2, 3, 4, 5, 6]),
    residual=np.array([-7., -30, -38, -16, -21, -37]))      # b - A x0, the six numbers the test prints first


def oracle_inner_smoother(M, spec, eig=None):
    """An "A00 Inverse" / "A11 Inverse" entry: HypreSmootherWrapper on M with iterative_mode = false."""
    Mo = orc.OParCSR(orc.diag_first(M))
    s = orc.OSmoother(Mo, type=spec.get("type", 1), sweeps=spec.get("sweeps", 1), weight=spec.get("weight", 1.0),
                      order=spec.get("order", 2), fraction=spec.get("fraction", 0.3), eig=eig)
    s.iterative_mode = False
    return s


def oracle_block_solver(blocks, offsets, kind="Block Jacobi", a00=None, a11=None, s_type="DIAGONAL", alpha=1.0,
                        use_negative_s=True, damping=1.0, eigs=None):
    """Block2x2{Jacobi,GaussSeidel,LDU}SolverFactory::_do_build_block_solver on serial CSR blocks.  Returns
    (solver, block operator, S)."""
    a00, a11 = a00 or {}, a11 or {}
    op = orc.OBlockOperator(offsets, offsets, {k: orc.OParCSR(orc.csr(v)) for k, v in blocks.items()})
    if s_type.upper() == "NONE":
        S = orc.csr(blocks[(1, 1)])
    else:
        S = orc.schur_complement(blocks[(0, 0)], blocks[(0, 1)], blocks[(1, 0)], blocks.get((1, 1)), s_type, alpha)
        if use_negative_s:
            S = orc.csr(S * -1.0)
    e0, e1 = eigs if eigs is not None else (None, None)
    if kind == "Block LDU":
        i0 = [oracle_inner_smoother(blocks[(0, 0)], a00, e0) for _ in range(3)]
        sol = orc.OBlock2x2LDU(op, i0[0], i0[1], i0[2], oracle_inner_smoother(S, a11, e1), damping)
    else:
        inv = [oracle_inner_smoother(blocks[(0, 0)], a00, e0), oracle_inner_smoother(S, a11, e1)]
        if kind == "Block Jacobi":
            sol = orc.OBlockDiagonalSolver(op, inv)
        else:
            sol = orc.OBlockTriangularSolver(op, inv, lower=not kind.endswith("Upper"))
    return sol, op, S


def oracle_blocked_hierarchy(blocks, Ps, eigs=None, **smoother_kw):
    """buildBlockedHierarchyFromDeRhamSequence (ParELAG_Hierarchy.cpp:400-541) for a 2x2 system without essential labels:
    per-block Galerkin products, P = blockdiag(P_i), the block smoother on every level but the coarsest, a dense direct
    solve there.  ``Ps[l] = (P_0, P_1)``.  Returns (OHierarchy, list of per-level block dicts)."""
    lev_blocks = [{k: orc.csr(v) for k, v in blocks.items()}]
    for Pl in Ps:
        Pc = [orc.csr(P) for P in Pl]
        lev_blocks.append({(i, j): orc.rap(Pc[i], Aij, Pc[j]) for (i, j), Aij in lev_blocks[-1].items()})
    levels = []
    nl = len(lev_blocks)
    for l, blk in enumerate(lev_blocks):
        n0, n1 = blk[(0, 0)].shape[0], blk[(1, 0)].shape[0]
        off = [0, n0, n0 + n1]
        lev = {}
        if l < nl - 1:
            pre, op, _S = oracle_block_solver(blk, off, eigs=None if eigs is None else eigs[l], **smoother_kw)
            post, _o, _s = oracle_block_solver(blk, off, eigs=None if eigs is None else eigs[l], **smoother_kw)
            pre.iterative_mode = post.iterative_mode = True
            lev.update(A=op, pre=pre, post=post)
        else:
            lev["A"] = orc.OBlockOperator(off, off, {k: orc.OParCSR(v) for k, v in blk.items()})
            dense = sp.bmat([[blk.get((0, 0)), blk.get((0, 1))], [blk.get((1, 0)), blk.get((1, 1))]]).toarray()
            lev["coarse"] = lambda rhs, sol, dense=dense: sol.__setitem__(slice(None), np.linalg.solve(dense, rhs))
        if l > 0:
            P0, P1 = (orc.csr(P) for P in Ps[l - 1])
            prev = lev_blocks[l - 1]
            roff = [0, prev[(0, 0)].shape[0], prev[(0, 0)].shape[0] + prev[(1, 0)].shape[0]]
            lev["P"] = orc.OBlockOperator(roff, off, {(0, 0): orc.OParCSR(P0), (1, 1): orc.OParCSR(P1)})
        else:
            lev["P"] = None
        levels.append(lev)
    return orc.OHierarchy(levels), lev_blocks


def gmres(A_mul, M_apply, b, rtol=1e-8, maxit=200, restart=50):
    """Right-preconditioned restarted GMRES (mfem::GMRESSolver stand-in for the tests).  Returns x, iterations."""
    n = len(b)
    x = np.zeros(n)
    r = b - A_mul(x)
    beta0 = np.linalg.norm(r)
    its = 0
    while its < maxit:
        beta = np.linalg.norm(r)
        if beta <= rtol * beta0:
            break
        V = [r / beta]
        Z = []
        H = np.zeros((restart + 1, restart))
        g = np.zeros(restart + 1); g[0] = beta
        k_done = 0
        for k in range(restart):
            z = np.zeros(n); M_apply(V[k], z)
            Z.append(z)
            w = A_mul(z)
            for i in range(k + 1):
                H[i, k] = w @ V[i]
                w = w - H[i, k] * V[i]
            H[k + 1, k] = np.linalg.norm(w)
            V.append(w / H[k + 1, k] if H[k + 1, k] > 0 else w)
            its += 1; k_done = k + 1
            y, res, _rk, _sv = np.linalg.lstsq(H[:k + 2, :k + 1], g[:k + 2], rcond=None)
            rn = np.linalg.norm(g[:k + 2] - H[:k + 2, :k + 1] @ y)
            if rn <= rtol * beta0 or its >= maxit:
                break
        for i in range(k_done):
            x = x + y[i] * Z[i]
        r = b - A_mul(x)
    return x, its
