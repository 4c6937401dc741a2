#!/usr/bin/env python3
"""Times the individual SpMV-class kernels of the cube456 x<k> hierarchy's two finest levels (tuning aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, _ = meshgen.h1_problem(k, nlevels=3)
def t(M, name, reps=30):
    n, m = M.Height(), M.Width()
    x = la.DeviceVector.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)); y = la.DeviceVector(n)
    ms = min(la.time_kernel(M, "spmv", x, y, reps=reps) for _ in range(3))
    b = 12 * M.nnz + 4 * (n + 1) + 8 * m + 8 * n
    print(f"{name:12s} rows={n:8d} cols={m:8d} nnz={M.nnz:9d}  {ms*1e3:8.1f} us  {b/ms/1e6:7.1f} GB/s", flush=True)
dA = la.ParallelCSRMatrix.from_scipy(A); t(dA, "A0")
dP = la.ParallelCSRMatrix.from_scipy(Ps[0]); t(dP, "P0")
dR = dP.Transpose(); t(dR, "R0=P0^T")
dA1 = la.RAP(dA, dP); t(dA1, "A1")
dP1 = la.ParallelCSRMatrix.from_scipy(Ps[1]); t(dP1, "P1"); t(dP1.Transpose(), "R1")
