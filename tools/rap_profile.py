#!/usr/bin/env python3
"""Fine-level Galerkin product of cube456 x<k> a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, _ = meshgen.h1_problem(k, nlevels=2)
dA, dP = la.ParallelCSRMatrix.from_scipy(A), la.ParallelCSRMatrix.from_scipy(Ps[0])
la.RAP(dA, dP); la.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); C = la.RAP(dA, dP); la.synchronize(); print("RAP ms", (time.perf_counter() - t0) * 1e3, C.nnz)
