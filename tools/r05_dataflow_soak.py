#!/usr/bin/env python3
"""Round 5 soak: the Gauss-Seidel cycle with the dataflow sweeps applied N times to the same right-hand side -- every result must be the first one
bit for bit (the arithmetic is deterministic; a stale or torn hand-off would show as a different vector or as a give-up) -- then the same through PCG
(the cycle replayed inside the iteration graph), and the one-XCD form on the level-1 operator.  python tools/r05_dataflow_soak.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
from parelag_amd import linalg as la, meshgen
from oracle import oracle as orc
A, Ps, _ = meshgen.h1_problem(3)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel")
n = A.shape[0]
rng = np.random.default_rng(0)
b = rng.uniform(-1, 1, n)
db, dx = la.DeviceVector.from_numpy(b), la.DeviceVector(n, 0.0)
H.Mult(db, dx); first = dx.to_numpy().copy()
t0 = time.time(); bad = 0
for i in range(N):
    H.Mult(db, dx)
    if i % 10 == 0 and not np.array_equal(dx.to_numpy(), first):
        bad += 1
print("cycle: %d applications, %d checked, %d differ, %.2f s" % (N, (N + 9) // 10, bad, time.time() - t0), flush=True)
H.iterative_mode = False
K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-10, "Maximum iterations": 100})
x = la.DeviceVector(n, 0.0); K.Mult(db, x); ref = x.to_numpy().copy(); its = K.GetNumIterations()
badk = 0
for i in range(max(1, N // 50)):
    x = la.DeviceVector(n, 0.0); K.Mult(db, x)
    if K.GetNumIterations() != its or not np.array_equal(x.to_numpy(), ref):
        badk += 1
print("PCG (%s): %d solves of %d iterations, %d differ" % (K.launch_mode(), max(1, N // 50), its, badk), flush=True)
# the smoother alone on an operator with narrow levels (the one-XCD form) and on level 0 (whole chip)
from tests.helpers import oracle_h1_hierarchy
ops = [orc.diag_first(orc.csr(M)) for M in oracle_h1_hierarchy(A, Ps)["As"][:2]]
for lvl, Ad in enumerate(ops):
    m = Ad.shape[0]
    S = la.HypreSmootherWrapper(la.ParallelCSRMatrix.from_scipy(Ad), 2, {"Sweeps": 1}); S.iterative_mode = True
    f, u0 = rng.standard_normal(m), rng.standard_normal(m)
    ref_s = orc.OSmoother(orc.OParCSR(Ad), type=2); ref_s.iterative_mode = True
    ur = u0.copy(); ref_s.Mult(f, ur)
    df = la.DeviceVector.from_numpy(f)
    bads = 0
    for i in range(N):
        du = la.DeviceVector.from_numpy(u0)
        S.Mult(df, du)
        if i % 10 == 0 and not np.array_equal(du.to_numpy(), ur):
            bads += 1
    print("smoother on level %d (%d rows): %d sweeps, %d differ from the oracle's loop" % (lvl, m, N, bads), flush=True)
print("SOAK_OK" if bad + badk == 0 else "SOAK_FAILED")
