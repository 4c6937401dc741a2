#!/usr/bin/env python3
"""Round 5: what a PCG pass costs with the direction update, the product and (d, A d) in one launch (cg_dir_amul_dot_kernel) and as three launches
(ELAG_HIP_CG_LEAN=0) -- the two solvers side by side in ONE process, solves of a pinned number of passes (tolerance 0) taken in turns, medians:
    python tools/r05_pcg_pass_probe.py [refines = 3] [passes = 256]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
A, Ps, _ = meshgen.h1_problem(k)
n = A.shape[0]
dA = la.ParallelCSRMatrix.from_scipy(A)
H = la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(P) for P in Ps], "L1 Jacobi")
b, x = la.DeviceVector(n, 1.0), la.DeviceVector(n, 0.0)
def make(env, passes):
    saved = {q: os.environ.get(q) for q in env}
    os.environ.update(env)
    try:
        return la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 0.0, "Absolute tolerance": 0.0, "Maximum iterations": passes, "Check every": passes})
    finally:
        for q, v in saved.items():
            os.environ.pop(q, None) if v is None else os.environ.__setitem__(q, v)
def solve_ms(K):
    la.synchronize(); t0 = time.perf_counter(); K.Mult(b, x); la.synchronize()
    return (time.perf_counter() - t0) * 1e3
res = {}
for tag, env in (("three launches", {"ELAG_HIP_CG_LEAN": "0"}), ("one launch", {"ELAG_HIP_CG_LEAN": "1"})):
    res[tag] = (make(env, N), make(env, 2 * N), [], [])
for rep in range(9):
    for tag, (K1, K2, t1, t2) in res.items():
        t1.append(solve_ms(K1)); t2.append(solve_ms(K2))
xs = {}
for tag, (K1, K2, t1, t2) in res.items():
    m1, m2 = float(np.median(t1[2:])), float(np.median(t2[2:]))
    K2.Mult(b, x); xs[tag] = x.to_numpy()
    print("%-14s: %d passes %.3f ms, %d passes %.3f ms: %.2f us a pass (launch mode %s)" % (tag, N, m1, 2 * N, m2, (m2 - m1) / N * 1e3, K2.launch_mode()), flush=True)
print("cube456 x%d, n = %d; the same iterate to the last bit: %s" % (k, n, np.array_equal(xs["three launches"], xs["one launch"])))
