#!/usr/bin/env python3
"""Round 5: the dense inverse of the Darcy hierarchy's coarsest (saddle-point) operator by phase (ELAG_HIP_LU_TIMING=1 prints the panel kernel's clocks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ELAG_HIP_LU_TIMING", "1")
import numpy as np, scipy.sparse as sp
from parelag_amd import linalg as la
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1471
rng = np.random.default_rng(0)
n0 = (2 * n) // 3
M = sp.random(n0, n0, density=8.0 / n0, random_state=rng); M = (M + M.T) * 0.5 + sp.diags(np.full(n0, 4.0))
B = sp.random(n - n0, n0, density=6.0 / n0, random_state=rng)
A = sp.bmat([[M, B.T], [B, -1e-3 * sp.identity(n - n0)]]).tocsr()
dA = la.ParallelCSRMatrix.from_scipy(A)
for rep in range(3):
    t0 = time.perf_counter()
    C = la.CoarseSolver(dA, kind=1)
    la.synchronize()
    print("n = %d: dense LU inverse %.2f ms" % (n, (time.perf_counter() - t0) * 1e3), flush=True)
b = rng.standard_normal(n); x = np.zeros(n); C.Mult(b, x)
print("residual", np.linalg.norm(A @ x - b) / np.linalg.norm(b))
