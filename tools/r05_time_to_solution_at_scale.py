#!/usr/bin/env python3
"""Round 5: hierarchy set-up + one PCG solve of MultigridTest0Form at cube456 x<k> (L1 Jacobi), as the library prices its set-up by default (many
cycles expected) and with ELAG_HIP_EXPECTED_CYCLES=16 (one solve): python tools/r05_time_to_solution_at_scale.py [k = 5]"""
import os, sys, time
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # (the checker's pools at the size of the CPU quota: bench.py, top)
    os.environ.setdefault(_v, "16")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, _ = meshgen.h1_problem(k)
n = A.shape[0]
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
b = la.DeviceVector(n, 2.0)
for hint in (None, "16", None, "16"):
    os.environ.pop("ELAG_HIP_EXPECTED_CYCLES", None)
    if hint: os.environ["ELAG_HIP_EXPECTED_CYCLES"] = hint
    la.synchronize(); t0 = time.perf_counter()
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi")
    H.iterative_mode = False
    la.synchronize(); t1 = time.perf_counter()
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 200})
    x = la.DeviceVector(n, 0.0)
    K.Mult(b, x); la.synchronize(); t2 = time.perf_counter()
    print("cube456 x%d, n = %d, ELAG_HIP_EXPECTED_CYCLES=%s: set-up %.2f ms, solve %.2f ms (%d iterations), together %.2f ms | %s" %
          (k, n, hint, (t1 - t0) * 1e3, (t2 - t1) * 1e3, K.GetNumIterations(), (t2 - t0) * 1e3, " ".join("%s=%.2f" % kv for kv in H.setup_breakdown().items())), flush=True)
    del K, H
os.environ.pop("ELAG_HIP_EXPECTED_CYCLES", None)
