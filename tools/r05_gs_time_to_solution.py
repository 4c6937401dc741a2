#!/usr/bin/env python3
"""Round 5: the reference's default configuration (MultigridTest0Form, cube456 x3, "L1 Gauss-Seidel", PCG to 1e-6) end to end: hierarchy set-up + solve,
with the dense tail wherever it is the cheaper cycle (the default) and with ELAG_HIP_DENSE_TAIL_CYCLES = the cycles one solve runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
A, Ps, _ = meshgen.h1_problem(3)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
n = A.shape[0]
rhs = la.DeviceVector(n, 2.0)
la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel"); la.synchronize()       # (the process's first build: kernels loaded, self-check done)
for cyc in (None, "8", "20", None, "8"):
    if cyc is None:
        os.environ.pop("ELAG_HIP_DENSE_TAIL_CYCLES", None)
    else:
        os.environ["ELAG_HIP_DENSE_TAIL_CYCLES"] = cyc
    la.synchronize(); t0 = time.perf_counter()
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel"); la.synchronize()
    t1 = time.perf_counter()
    H.iterative_mode = False
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 200})
    x = la.DeviceVector(n, 0.0); K.Mult(rhs, x); la.synchronize()
    t2 = time.perf_counter()
    x2 = la.DeviceVector(n, 0.0); K.Mult(rhs, x2); la.synchronize()
    t3 = time.perf_counter()
    ms = H.time_apply(rhs, la.DeviceVector(n, 1.0), 100)
    print("ELAG_HIP_DENSE_TAIL_CYCLES=%s: setup %.2f ms, first solve %.2f ms (%d iterations), second solve %.2f ms, cycle %.3f ms, kernels per cycle %s: setup + one solve %.2f ms" %
          (cyc, (t1 - t0) * 1e3, (t2 - t1) * 1e3, K.GetNumIterations(), (t3 - t2) * 1e3, ms, H.launch_mode().get("kernels_per_cycle"), (t1 - t0) * 1e3 + (t3 - t2) * 1e3), flush=True)
