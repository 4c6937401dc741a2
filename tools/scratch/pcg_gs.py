import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["ELAG_HIP_KRYLOV_TIMING"] = "1"
import numpy as np
from parelag_amd import linalg as la, meshgen
A, Ps, _ = meshgen.h1_problem(3)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
n = A.shape[0]
rhs = la.DeviceVector(n, 2.0)
for sm in ("L1 Gauss-Seidel", "L1 Jacobi"):
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, sm); la.synchronize()
    H.iterative_mode = False
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 200})
    x = la.DeviceVector(n, 0.0)
    for rep in range(4):
        la.synchronize(); t0 = time.perf_counter(); K.Mult(rhs, x); la.synchronize(); t1 = time.perf_counter()
        print(sm, "solve %.3f ms, %d iterations, launch mode %s" % ((t1 - t0) * 1e3, K.GetNumIterations(), K.launch_mode()), flush=True)
    ms = H.time_apply(rhs, la.DeviceVector(n, 1.0), 100)
    print(sm, "cycle %.3f ms" % ms, H.launch_mode())
