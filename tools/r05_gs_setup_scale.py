#!/usr/bin/env python3
"""Round 5: where the set-up of a Gauss-Seidel smoother goes on a large operator (cube456 x k level 0): total, with the dataflow schedule and without."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
from parelag_amd import linalg as la, meshgen
from oracle import oracle as orc
A, Ps, _ = meshgen.h1_problem(k)
Ad = orc.diag_first(A)
dA = la.ParallelCSRMatrix.from_scipy(Ad)
S0 = la.HypreSmootherWrapper(dA, 2, {"Sweeps": 1}); la.synchronize()      # (first: the process's self-check)
for mode in ("0", "1", "0", "1"):
    os.environ["ELAG_HIP_SWEEP_DATAFLOW"] = mode
    t0 = time.perf_counter(); S = la.HypreSmootherWrapper(dA, 2, {"Sweeps": 1}); la.synchronize()
    print("x%d n=%d nnz=%d dataflow=%s: smoother setup %.2f ms" % (k, Ad.shape[0], Ad.nnz, mode, (time.perf_counter() - t0) * 1e3), flush=True)
