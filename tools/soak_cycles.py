"""Soak: 200 000 back-to-back V-cycles of the headline hierarchy (cube456 x3, L1 Jacobi) on device vectors; prints how Finalize chose to issue
the cycle, the sustained microseconds per cycle (host clock) and whether the last result equals the first bit for bit."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from parelag_amd import meshgen, linalg as la
A, Ps, seq = meshgen.h1_problem(3)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi")
print(H.launch_mode())
n = A.shape[0]
rhs = la.DeviceVector.from_numpy(np.random.default_rng(0).uniform(-1, 1, n)); sol = la.DeviceVector(n, 0.0)
H.Mult(rhs, sol); ref = sol.to_numpy().copy()
t = time.perf_counter()
N = 200000
for i in range(N):
    H.Mult(rhs, sol)
la.synchronize()
dt = time.perf_counter() - t
out = sol.to_numpy()
print("cycles", N, "us/cycle %.2f" % (dt / N * 1e6), "identical", np.array_equal(out, ref))
