#!/usr/bin/env python3
"""Round 5: what ONE run of the reference's default example pays on the device side, phase by phase, the first time in a process and once more --
library start, upload of the operator and the interpolators, hierarchy build, solver construction, PCG solve (cube456 x3):
    python tools/r05_single_run.py ["L1 Jacobi" | "L1 Gauss-Seidel"]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sm = sys.argv[1] if len(sys.argv) > 1 else "L1 Gauss-Seidel"
t0 = time.perf_counter()
from parelag_amd import linalg as la, meshgen
from parelag_amd import _lib
_lib.init()
la.synchronize()
t_init = time.perf_counter() - t0
A, Ps, _ = meshgen.h1_problem(3)
n = A.shape[0]
b = np.full(n, 2.0)
def once(tag):
    la.synchronize(); t = [time.perf_counter()]
    dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]; la.synchronize(); t.append(time.perf_counter())
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, sm); la.synchronize(); t.append(time.perf_counter())
    H.iterative_mode = False
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 200}); la.synchronize(); t.append(time.perf_counter())
    db, dx = la.DeviceVector.from_numpy(b), la.DeviceVector(n, 0.0); la.synchronize(); t.append(time.perf_counter())
    K.Mult(db, dx); la.synchronize(); t.append(time.perf_counter())
    x = dx.to_numpy(); t.append(time.perf_counter())
    d = np.diff(t) * 1e3
    print("%s (%s): upload %.2f ms, hierarchy %.2f, solver %.2f, vectors %.2f, solve %.2f (%d iterations), download %.2f; total %.2f ms" %
          (tag, sm, d[0], d[1], d[2], d[3], d[4], K.GetNumIterations(), d[5], d.sum()), flush=True)
print("import + library start %.1f ms%s" % (t_init * 1e3, "".join("; %s=%s" % (q, os.environ[q]) for q in ("ELAG_HIP_EXPECTED_CYCLES", "ELAG_HIP_DENSE_TAIL_CYCLES") if q in os.environ)))
once("first run of the process")
once("second")
once("third")
