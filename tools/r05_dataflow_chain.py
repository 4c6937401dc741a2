#!/usr/bin/env python3
"""Round 5: the price of one hand-off of the Gauss-Seidel dataflow sweep -- a tridiagonal operator makes every row its own dependency level,
so a symmetric sweep is 2 n hops and nothing else:  python tools/r05_dataflow_chain.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
from parelag_amd import linalg as la
from oracle import oracle as orc
A = orc.diag_first(sp.diags([np.full(n - 1, -1.0), np.full(n, 2.0), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsr())
dA = la.ParallelCSRMatrix.from_scipy(A)
rng = np.random.default_rng(0)
f, u0 = rng.standard_normal(n), rng.standard_normal(n)
ref = orc.OSmoother(orc.OParCSR(A), type=2); ref.iterative_mode = True
ur = u0.copy(); ref.Mult(f, ur)
for xcd in ("0", "1"):
    os.environ["ELAG_HIP_SWEEP_DATAFLOW"] = "2"; os.environ["ELAG_HIP_SWEEP_DATAFLOW_XCD"] = xcd
    S = la.HypreSmootherWrapper(dA, 2, {"Sweeps": 1}); S.iterative_mode = True
    u = u0.copy(); S.Mult(f, u)
    df, du = la.DeviceVector.from_numpy(f), la.DeviceVector.from_numpy(u0)
    for _ in range(3):
        S.Mult(df, du)
    la.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        S.Mult(df, du)
    la.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
    print("one XCD" if xcd == "1" else "whole chip", "n = %d: %.1f us per symmetric sweep = %.3f us per hop; the sequential loop's bits: %s" % (n, us, us / (2 * n), np.array_equal(u, ur)), flush=True)
