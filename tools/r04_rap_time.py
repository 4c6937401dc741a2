#!/usr/bin/env python3
"""The fine-level Galerkin product of cube456 x<k>, timed like bench.py's roofline.rap (wall time around a drained call, the median of n)."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = int(sys.argv[2]) if len(sys.argv) > 2 else 15
A, Ps, _ = meshgen.h1_problem(k, nlevels=2)
dA, dP = la.ParallelCSRMatrix.from_scipy(A), la.ParallelCSRMatrix.from_scipy(Ps[0])
for _ in range(3):
    C = la.RAP(dA, dP); la.synchronize()
ts = []
for _ in range(n):
    la.synchronize(); t0 = time.perf_counter(); C = la.RAP(dA, dP); la.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("RAP ms: median %.4f  min %.4f  (%d products, nnz %d)" % (statistics.median(ts), min(ts), n, C.nnz))
