#!/usr/bin/env python3
"""The level-0 restriction R0 = P0^T of cube456 x<k> in the hierarchy's own (renumbered) ordering, timed in every kernel form
(tuning aid): stream kernel, x-cache, 1..16 lanes per row."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, _ = meshgen.h1_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi")
o0, o1 = H.GetOrdering(0), H.GetOrdering(1)
R = sp.csr_matrix(Ps[0])[o0][:, o1].T.tocsr()
R.sort_indices()
dR = la.ParallelCSRMatrix.from_scipy(R)
n, m = dR.Height(), dR.Width()
x = la.DeviceVector.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)); y = la.DeviceVector(n)
b = 12 * R.nnz + 4 * (n + 1) + 8 * m + 8 * n
for code, name in ((0, "heuristic"), (-2, "stream"), (-1, "x-cache"), (1, "1 lane"), (2, "2 lanes"), (4, "4 lanes"), (8, "8 lanes"), (16, "16 lanes")):
    try:
        dR.set_spmv_lanes(code)
        ms = min(la.time_kernel(dR, "spmv", x, y, reps=30) for _ in range(3))
        print(f"{name:10s} {dR.spmv_kernel():22s} {ms*1e3:8.1f} us  {b/ms/1e6:7.1f} GB/s", flush=True)
    except Exception as e:
        print(name, "failed:", str(e)[:100])
