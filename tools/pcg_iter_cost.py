#!/usr/bin/env python3
"""Marginal cost of a PCG + V-cycle iteration on cube456 x<k>: solves with the iteration count pinned by "Maximum iterations"
(tolerance 0), so that time(N2) - time(N1) prices N2 - N1 iterations without the setup and polling share (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, _ = meshgen.h1_problem(k)
n = A.shape[0]
dA = la.ParallelCSRMatrix.from_scipy(A)
dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi")
b = la.DeviceVector(n, 1.0); x = la.DeviceVector(n, 0.0)
print("cycle (natural in/out) %.4f ms, cycle in the hierarchy's numbering %.4f ms" % (H.time_apply(b, x, 20), H.time_apply_internal(20)))
res = {}
for ce in (8, 64):
    for N in (8, 16, 32, 64):
        K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 0.0, "Absolute tolerance": 0.0, "Maximum iterations": N, "Check every": ce})
        K.Mult(b, x); la.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            K.Mult(b, x)
        la.synchronize()
        res[(ce, N)] = (time.perf_counter() - t0) / 3 * 1e3
        print("check_every %2d  N=%3d: %8.3f ms per solve (%d iterations done)" % (ce, N, res[(ce, N)], K.GetNumIterations()), flush=True)
    print("  marginal: %.4f ms per iteration" % ((res[(ce, 64)] - res[(ce, 16)]) / 48))
