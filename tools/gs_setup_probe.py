"""Hierarchy setup with the reference's shipped relaxation ("L1 Gauss-Seidel") on 1 / 256 / 2048 virtual ranks: build time by phase, first
and second Mult (tuning aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
A, Ps, _ = meshgen.h1_problem(3)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
for B in (1, 256, 2048, 1):
    la.set_gs_hybrid_blocks(B)
    t0 = time.time(); H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel"); la.synchronize(); t1 = time.time()
    rhs, sol = la.DeviceVector(A.shape[0], 2.0), la.DeviceVector(A.shape[0], 1.0)
    t2 = time.time(); H.Mult(rhs, sol); la.synchronize(); t3 = time.time(); H.Mult(rhs, sol); la.synchronize(); t4 = time.time()
    print(B, "build %.1f ms" % ((t1 - t0) * 1e3), "first Mult %.1f ms, second %.2f ms" % ((t3 - t2) * 1e3, (t4 - t3) * 1e3), H.setup_breakdown(), H.launch_mode() if hasattr(H, "launch_mode") else "")
    del H
