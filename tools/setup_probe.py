#!/usr/bin/env python3
"""Builds the headline hierarchy (cube456 x<k>, L1 Jacobi) a few times and prints the setup breakdown of the last build: the
subject of `rocprofv3 --hip-trace --stats` / `--kernel-trace --stats` runs that show where Finalize's wall time goes.

    python tools/setup_probe.py [refines] [builds] [smoother]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import linalg as la, meshgen, _lib
_lib.init()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sm = sys.argv[3] if len(sys.argv) > 3 else "L1 Jacobi"
A, Ps, _ = meshgen.h1_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A)
dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
for i in range(nb):
    la.synchronize()
    t0 = time.perf_counter()
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, sm)
    la.synchronize()
    dt = time.perf_counter() - t0
    sb = H.setup_breakdown()
    print("build %d: %.2f ms | %s" % (i, dt * 1e3, " ".join("%s=%.2f" % kv for kv in sb.items())), flush=True)
    del H
