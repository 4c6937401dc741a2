#!/usr/bin/env python3
"""Repeated builds of the H(div) x<k> Hiptmair hierarchy with the setup breakdown: python tools/r04_hdiv_setup_probe.py [k] [builds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
A, Ps, Ds, _ = meshgen.hdiv_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]; dDs = [la.ParallelCSRMatrix.from_scipy(D) for D in Ds[:len(Ps)]]
for i in range(nb):
    la.synchronize(); t0 = time.perf_counter()
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi", Ds=dDs, aux_smoother="L1 Jacobi")
    la.synchronize(); dt = time.perf_counter() - t0
    print("build %d: %.2f ms | %s" % (i, dt * 1e3, " ".join("%s=%.2f" % kv for kv in H.setup_breakdown().items())), flush=True)
    del H
