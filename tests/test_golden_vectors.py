"""Golden vectors of the V-cycle path (tests/golden/oracle_h1_k1.npz, written by tools/make_oracle_golden.py from the CPU
oracle: nothing can be captured from the reference itself).  The CPU half pins the oracle against change; the
GPU half checks the device against the same stored numbers through the C ABI."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import oracle_h1_hierarchy

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_h1_k1.npz"))


def rel(x, ref):
    return np.max(np.abs(np.asarray(x) - np.asarray(ref))) / max(np.max(np.abs(ref)), 1e-300)


def test_oracle_reproduces_its_golden_vectors(h1_k1):
    A, Ps, seq = h1_k1
    n = A.shape[0]
    assert (n, A.nnz) == (int(G["n"]), int(G["nnz"]))
    f, u0 = G["f"], G["u0"]
    rng = np.random.default_rng(int(G["seed"]))
    assert np.array_equal(rng.uniform(-1, 1, n), f) and np.array_equal(rng.uniform(-1, 1, n), u0)   # inputs are reproducible
    Ao = orc.OParCSR(orc.diag_first(A))
    assert np.array_equal(orc.l1_norms(orc.diag_first(A), None, 1), G["l1"])
    y = np.zeros(n); Ao.matvec(1.0, u0, 0.0, y)
    assert rel(y, G["spmv"]) <= 1e-15
    s = orc.OSmoother(Ao, type=1, sweeps=2, weight=0.9); u = u0.copy(); s.Mult(f, u)
    assert rel(u, G["l1_jacobi_2sweeps_w09"]) <= 1e-15
    mx, mn = orc.max_eig_estimate_cg(Ao, 1, 10)
    assert np.allclose([mx, mn], G["eig_estimate"], rtol=1e-13)
    c = orc.OSmoother(Ao, type=16, order=3, eig=tuple(G["eig_estimate"])); u = u0.copy(); c.Mult(f, u)
    assert rel(u, G["cheby_order3"]) <= 1e-14
    Ac = orc.rap(orc.csr(Ps[0]), orc.diag_first(A), orc.csr(Ps[0])); orc.fix_zero_rows(Ac); Ac.sort_indices()
    assert np.array_equal(Ac.indptr, G["rap_indptr"]) and np.array_equal(Ac.indices, G["rap_indices"])
    assert rel(Ac.data, G["rap_data"]) <= 1e-15
    h = oracle_h1_hierarchy(A, Ps, smoother=1)
    x = np.empty(n); orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"]).apply(np.full(n, 2.0), x)
    assert rel(x, G["vcycle_b2"]) <= 1e-14
    assert np.array_equal(orc.partition_refined(seq.meshes[0].nt, seq.meshes[1].nt), G["partition"])


@pytest.mark.gpu
def test_device_matches_the_golden_vectors(h1_k1):
    from parelag_amd import linalg as la
    A, Ps, seq = h1_k1
    n = A.shape[0]
    f, u0 = G["f"], G["u0"]
    dA = la.ParallelCSRMatrix.from_scipy(A)
    y = np.zeros(n); dA.Mult(u0, y)
    assert rel(y, G["spmv"]) <= 1e-13
    s = la.HypreSmootherWrapper(dA, 1, {"Sweeps": 2, "Damping Factor": 0.9}); u = u0.copy(); s.Mult(f, u)
    assert rel(u, G["l1_jacobi_2sweeps_w09"]) <= 1e-13
    c = la.HypreSmootherWrapper(dA, 16, {"Cheby Poly Order": 3})
    assert np.allclose(c.eig_estimate(), G["eig_estimate"], rtol=1e-9)
    c.set_eig_estimate(*G["eig_estimate"]); u = u0.copy(); c.Mult(f, u)
    assert rel(u, G["cheby_order3"]) <= 1e-12
    dP = la.ParallelCSRMatrix.from_scipy(Ps[0])
    Ac = la.RAP(dA, dP); Ac.fix_zero_rows(); C = Ac.to_scipy(); C.sort_indices()
    assert np.array_equal(C.indptr, G["rap_indptr"]) and np.array_equal(C.indices, G["rap_indices"])     # pattern bit-exact
    assert rel(C.data, G["rap_data"]) <= 1e-13
    H = la.buildHierarchyFromDeRhamSequence(dA, [dP], "L1 Jacobi")
    x = np.empty(n); H.Mult(np.full(n, 2.0), x)
    assert rel(x, G["vcycle_b2"]) <= 1e-12
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-10, "Maximum iterations": 50})
    xs = np.zeros(n); K.Mult(f, xs)
    assert K.GetNumIterations() == int(G["pcg_iterations"]) and rel(xs, G["pcg_solution"]) <= 1e-9
    assert np.array_equal(la.partition_refined(seq.meshes[0].nt, seq.meshes[1].nt), G["partition"])
