#!/usr/bin/env python3
"""The L1 Gauss-Seidel V-cycle of cube456 x<k> with the sweeps on B virtual ranks (elag_hip_set_gs_hybrid_blocks): cycle time and
what the PCG of the example makes of it.  usage: gs_blocks_probe.py [k] [B ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
Bs = [int(v) for v in sys.argv[2:]] or [1, 8, 32, 128, 512]
A, Ps, _ = meshgen.h1_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
n = A.shape[0]
b = la.DeviceVector(n, 1.0)
for B in Bs:
    la.set_gs_hybrid_blocks(B)
    os.environ["ELAG_HIP_SWEEP_VERBOSE"] = "1"
    t0 = time.time()
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Gauss-Seidel")
    ts = time.time() - t0
    os.environ.pop("ELAG_HIP_SWEEP_VERBOSE")
    rhs, sol = la.DeviceVector(n, 2.0), la.DeviceVector(n, 1.0)
    ms = H.time_apply(rhs, sol, 50)
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 200})
    x = la.DeviceVector(n, 0.0)
    K.Mult(b, x); la.synchronize()
    t0 = time.perf_counter(); x2 = la.DeviceVector(n, 0.0); K.Mult(b, x2); la.synchronize(); tsolve = time.perf_counter() - t0
    print(f"B={B:5d}: {ms*1e3:8.1f} us per V-cycle = {1e3/ms:8.1f} /s   setup {ts*1e3:7.1f} ms   PCG {K.GetNumIterations()} iterations, {tsolve*1e3:.2f} ms", flush=True)
    del K, H
la.set_gs_hybrid_blocks(0)
