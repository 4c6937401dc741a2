#!/usr/bin/env python3
"""FETCH_SIZE calibration for the stream SpMV (MI355X_MICROARCH.md, HBM: 'calibrate on a known byte count in your
own access pattern').  Same row structure and values as the cube456 x<k> operator but every column index is 0, so
the gather always hits one cached line and the HBM read traffic of a launch is exactly the matrix stream:
12 nnz + 4 (n+1) bytes (+ 8 n for the y write, counted by WRITE_SIZE)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, _, _ = meshgen.h1_problem(k, nlevels=1)
n, nnz = A.shape[0], A.nnz
B = sp.csr_matrix((A.data, np.zeros(nnz, np.int32), A.indptr), shape=A.shape)
dB = la.ParallelCSRMatrix.from_scipy(B)
dB.set_spmv_lanes(-2)     # the plain stream kernel: its HBM read bytes are exactly the J / a / I streams counted below
x = la.DeviceVector.from_numpy(np.ones(n)); y = la.DeviceVector(n)
ms = la.time_kernel(dB, "spmv", x, y, reps=50)
print(json.dumps({"calibration": True, "n": n, "nnz": nnz, "known_read_bytes": 12 * nnz + 4 * (n + 1), "ms_per_launch": ms}))
