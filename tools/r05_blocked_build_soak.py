#!/usr/bin/env python3
"""Round 5: the blocked Darcy hierarchy and the H(div) Hiptmair hierarchy of cube456 x3 built over and over with the coarsest operator's inverse running
on the second stream beside the build -- every build's cycle must be the first one's bits (the job's scratch and the build's never meet), device memory constant:
    python tools/r05_blocked_build_soak.py [builds = 100]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
hip = C.CDLL("libamdhip64.so")
def used_mb():
    free, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(free), C.byref(tot))
    return (tot.value - free.value) / 2**20
dar = meshgen.darcy_problem(3)
n0, n1 = dar["M"].shape[0], dar["B"].shape[0]
op = la.MfemBlockOperator([0, n0, n0 + n1])
for (i, j), M in {(0, 0): dar["M"], (0, 1): dar["Bt"], (1, 0): dar["B"]}.items():
    op.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(M))
dPs = [[la.ParallelCSRMatrix.from_scipy(P) for P in Pl] for Pl in dar["Ps"]]
b = np.random.default_rng(0).uniform(-1, 1, n0 + n1)
db = la.DeviceVector.from_numpy(b)
for tag, env in (("inverse beside the build", {}), ("inverse where it is created", {"ELAG_HIP_COARSE_ASYNC": "0"})):
    os.environ.pop("ELAG_HIP_COARSE_ASYNC", None); os.environ.update(env)
    ref, m0, t0 = None, None, time.perf_counter()
    for i in range(N if not env else 5):
        H = la.buildBlockedHierarchyFromDeRhamSequence(op, dPs, block_smoother="Block Jacobi")
        dx = la.DeviceVector(n0 + n1, 0.0)
        H.Mult(db, dx)
        x = dx.to_numpy()
        if ref is None: ref = x.copy(); first = first if "first" in dir() else x.copy()
        assert np.array_equal(x, ref) and np.array_equal(x, first), "build %d gives another cycle" % i
        del H, dx
        if i == 4: m0 = used_mb()
    la.synchronize()
    k = N if not env else 5
    print("Darcy x3 blocked hierarchy, %s: %d builds, %.2f ms per build + cycle, device MB after 5 builds %.1f, at the end %.1f, every cycle the first one's bits" %
          (tag, k, (time.perf_counter() - t0) / k * 1e3, m0, used_mb()), flush=True)
os.environ.pop("ELAG_HIP_COARSE_ASYNC", None)
