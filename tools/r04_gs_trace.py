#!/usr/bin/env python3
"""A few L1 Gauss-Seidel V-cycles at cube456 x<k> without a graph (for rocprofv3 --kernel-trace: the timeline of one cycle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
A, Ps, _ = meshgen.h1_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A)
dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel", use_graph=False)
n = A.shape[0]
b, x = la.DeviceVector(n, 1.0), la.DeviceVector(n, 0.0)
for _ in range(6):
    H.Mult(b, x)
la.synchronize()
print("ms per cycle", H.time_apply(b, x, 20))
