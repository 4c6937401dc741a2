#!/usr/bin/env python3
"""Times the fine-level SpMV on cube456 x<k> for several lanes-per-row settings (tuning aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, _, _ = meshgen.h1_problem(k, nlevels=1)
n, nnz = A.shape[0], A.nnz
b = 12 * nnz + 4 * (n + 1) + 16 * n
dA = la.ParallelCSRMatrix.from_scipy(A)
x = la.DeviceVector.from_numpy(np.random.default_rng(1).uniform(-1, 1, n)); y = la.DeviceVector(n)
for lanes in [int(v) for v in sys.argv[2:]] or [0]:
    dA.set_spmv_lanes(lanes)
    ms = min(la.time_kernel(dA, "spmv", x, y, reps=30) for _ in range(3))
    print(f"k={k} lanes={lanes:2d}: {ms*1e3:8.1f} us  {b/ms/1e6:7.1f} GB/s  ({b/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)
