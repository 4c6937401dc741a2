"""Soak of the setup path: N builds of the cube456 x<k> hierarchy (Galerkin products, renumbering, fused / flattened operators, dense tail)
with a V-cycle each -- device memory in use before and after (a leak shows as growth), every result equal to the first bit for bit.
usage: soak_builds.py [k] [N] [smoother]"""
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from parelag_amd import meshgen, linalg as la, _lib
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sm = sys.argv[3] if len(sys.argv) > 3 else "L1 Jacobi"
A, Ps, seq = meshgen.h1_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
hip = C.CDLL("libamdhip64.so")
def used_mb():
    free, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(free), C.byref(tot))
    return (tot.value - free.value) / 2**20
n = A.shape[0]
rhs = la.DeviceVector.from_numpy(np.random.default_rng(0).uniform(-1, 1, n))
ref, m0, t0 = None, None, time.perf_counter()
for i in range(N):
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, sm)
    sol = la.DeviceVector(n, 0.0)
    H.Mult(rhs, sol)
    x = sol.to_numpy()
    if ref is None: ref = x.copy()
    assert np.array_equal(x, ref), "build %d gives another cycle" % i
    del H, sol
    if i == 4: m0 = used_mb()
la.synchronize()
print("builds", N, "smoother", sm, "ms/build+cycle %.2f" % ((time.perf_counter() - t0) / N * 1e3), "device MB after 5 builds %.1f, at the end %.1f" % (m0, used_mb()), "identical: True")
