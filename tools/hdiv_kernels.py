#!/usr/bin/env python3
"""Times the SpMV-class kernels of the H(div) + Hiptmair hierarchy of cube456 x<k> one by one, for several kernel forms (tuning aid):
A, D, D^T, P, P^T of the two finest levels; lanes 0 = library heuristic, -2 = stream kernel, 1..16 = row-group kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
forms = [int(v) for v in sys.argv[2:]] or [0, -2, 1, 2, 4, 8]
A, Ps, Ds, _ = meshgen.hdiv_problem(k)
def t(M, name, reps=50):
    n, m = M.Height(), M.Width()
    x = la.DeviceVector.from_numpy(np.random.default_rng(1).uniform(-1, 1, m)); y = la.DeviceVector(n)
    b = 12 * M.nnz + 4 * (n + 1) + 8 * m + 8 * n
    out = []
    for f in forms:
        try:
            M.set_spmv_lanes(f)
        except Exception as e:
            out.append("%4d: n/a" % f); continue
        ms = min(la.time_kernel(M, "spmv", x, y, reps=reps) for _ in range(3))
        out.append("%4d:%6.1fus %5.0fGB/s" % (f, ms * 1e3, b / ms / 1e6))
    print(f"{name:8s} {n:7d}x{m:7d} nnz={M.nnz:8d} avg={M.nnz/max(n,1):4.1f} | " + " | ".join(out), flush=True)
dA = la.ParallelCSRMatrix.from_scipy(A); t(dA, "A0")
dD = la.ParallelCSRMatrix.from_scipy(Ds[0]); t(dD, "D0"); t(dD.Transpose(), "D0^T")
dP = la.ParallelCSRMatrix.from_scipy(Ps[0]); t(dP, "P0"); t(dP.Transpose(), "P0^T")
dX = la.RAP(dA, dD); t(dX, "Aaux0")
dA1 = la.RAP(dA, dP); t(dA1, "A1")
dD1 = la.ParallelCSRMatrix.from_scipy(Ds[1]); t(dD1, "D1"); t(dD1.Transpose(), "D1^T")
dP1 = la.ParallelCSRMatrix.from_scipy(Ps[1]); t(dP1, "P1"); t(dP1.Transpose(), "P1^T")
print("values of D0: distinct", np.unique(Ds[0].data)[:10], " P0 distinct:", np.unique(np.round(Ps[0].data, 12))[:12])
