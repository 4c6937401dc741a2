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
        if smoother in (1, 2):
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
