#!/usr/bin/env python3
"""Round 5: the one 30-80 ms stall that follows a Darcy set-up by ~0.1 s -- when does it come, how long is it, what does it follow?
After the set-up the blocked cycle (87 us) is applied in a loop with a drain after every call; calls longer than 2 ms are printed with their start time.
Variants: python tools/r05_stall_probe.py [keep|drop] [trim|notrim]   (keep / drop the host copies of the operators; trim the library's idle memory)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
keep = (sys.argv[1] if len(sys.argv) > 1 else "keep") == "keep"
trim = (sys.argv[2] if len(sys.argv) > 2 else "notrim") == "trim"
from parelag_amd import linalg as la, meshgen, _lib
seq = meshgen.Sequence.cube456(3)
dar = meshgen.darcy_problem(3, seq=seq)
Ps = dar["Ps"]
n0, n1 = dar["M"].shape[0], dar["B"].shape[0]
bop = la.MfemBlockOperator([0, n0, n0 + n1])
for (i, j), M in {(0, 0): dar["M"], (0, 1): dar["Bt"], (1, 0): dar["B"]}.items():
    bop.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(M))
dPs = [[la.ParallelCSRMatrix.from_scipy(P) for P in Pl] for Pl in Ps]
la.synchronize()
gc.disable()
t_setup0 = time.perf_counter()
H = la.buildBlockedHierarchyFromDeRhamSequence(bop, dPs, block_smoother="Block Jacobi", a00_inverse="L1 Jacobi", a11_inverse="L1 Jacobi", use_negative_s=True)
ldu = la.darcy_ldu_amge_preconditioner(bop, dar["M"], dar["B"], Ps, "L1 Jacobi", dPs=dPs)
la.synchronize()
t0 = time.perf_counter()
print("set-up %.1f ms" % ((t0 - t_setup0) * 1e3), flush=True)
if not keep:
    del dar, seq
if trim:
    print("trimmed %d MB" % (_lib.trim_memory() >> 20))
n = n0 + n1
rhs, sol = la.DeviceVector(n, 2.0), la.DeviceVector(n, 1.0)
what = sys.argv[3] if len(sys.argv) > 3 else "cycles"
big = None
if what == "alloc":                       # a block of the size of GMRES(50)'s basis, untouched
    big = la.DeviceVector(n * 51, 0.0)
K = None
if what in ("krylov", "solves"):
    K = la.KrylovSolver(bop, ldu, {"Solver name": "GMRES", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 300, "Restart size": 50})
    b_solve = la.DeviceVector.from_numpy(np.concatenate([np.zeros(n0), np.ones(n1)]))
    x_solve = la.DeviceVector(n, 0.0)
if what == "solves" and len(sys.argv) > 4:
    # something is released right before the timed solves: "krylov" = another solver object (its basis goes to the library's pool, its state block and
    # pinned mirrors to hipFree / hipHostFree), "hier" = another blocked hierarchy, "host" = 1 GB of host memory that was the source of an upload
    if sys.argv[4] == "krylov":
        K0 = la.KrylovSolver(bop, ldu, {"Solver name": "GMRES", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 300, "Restart size": 50})
        K0.Mult(b_solve, x_solve); la.synchronize()
        del K0
    elif sys.argv[4] == "hier":
        H0 = la.buildBlockedHierarchyFromDeRhamSequence(bop, dPs, block_smoother="Block Jacobi", a00_inverse="L1 Jacobi", a11_inverse="L1 Jacobi", use_negative_s=True)
        la.synchronize()
        del H0
    elif sys.argv[4] == "host":
        big_h = np.ones(128 * 1024 * 1024)
        big_d = la.DeviceVector.from_numpy(big_h); la.synchronize()
        del big_h
    elif sys.argv[4] == "devfree":
        big_d = la.DeviceVector(256 * 1024 * 1024, 0.0); la.synchronize()      # 2 GB: beyond the pool
        del big_d
    t0 = time.perf_counter()
long_calls, calls = [], 0
while time.perf_counter() - t0 < 0.6:
    a = time.perf_counter()
    if what == "solves":
        K.Mult(b_solve, x_solve)
    else:
        H.Mult(rhs, sol)
    la.synchronize()
    b = time.perf_counter()
    calls += 1
    if b - a > (30e-3 if what == "solves" else 2e-3):
        long_calls.append(((a - t0) * 1e3, (b - a) * 1e3))
print("after the set-up:", what)
print("host copies %s, %s: %d cycles in 0.6 s; calls over 2 ms (start ms after set-up, length ms): %s" %
      ("kept" if keep else "dropped", "trim" if trim else "no trim", calls, [(round(x, 1), round(y, 1)) for x, y in long_calls]), flush=True)
