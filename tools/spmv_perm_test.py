#!/usr/bin/env python3
"""Does a hierarchy-derived dof ordering speed up the fine-level SpMV? (experiment)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, seq = meshgen.h1_problem(k)
A = sp.csr_matrix(A); A.sort_indices()
n, nnz = A.shape
b = 12 * A.nnz + 4 * (n + 1) + 16 * n
def timeit(M, name):
    dA = la.ParallelCSRMatrix.from_scipy(M)
    x = la.DeviceVector.from_numpy(np.random.default_rng(1).uniform(-1, 1, n)); y = la.DeviceVector(n)
    ms = min(la.time_kernel(dA, "spmv", x, y, reps=30) for _ in range(3))
    print(f"{name:10s}: {ms*1e3:8.1f} us  {b/ms/1e6:7.1f} GB/s  ({b/ms/1e6/8000:.3f} of 8 TB/s)", flush=True)
timeit(A, "natural")
# hierarchical ordering: coarsest by RCM of the pattern, children sorted by parent rank
Ac = seq.meshes[-1].M0_P1_stiffness()
order = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(Ac), symmetric_mode=True))
for P in Ps[::-1]:
    Pp = sp.csr_matrix(P).copy(); Pp.data = np.where(Pp.data == 0, 1e-3, abs(Pp.data))
    parent = np.asarray(Pp.argmax(axis=1)).ravel()
    rank_c = np.empty(len(order), np.int64); rank_c[order] = np.arange(len(order))
    order = np.argsort(rank_c[parent], kind='stable')
B = A[order][:, order].tocsr(); B.sort_indices()
timeit(B, "hier")
v = seq.meshes[0].vertices
q = np.minimum((v * 1024).astype(np.int64), 1023)
def spread(x):
    x = (x | (x << 16)) & 0x030000FF
    x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3
    x = (x | (x << 2)) & 0x09249249
    return x
pm = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind='stable')
C = A[pm][:, pm].tocsr(); C.sort_indices()
timeit(C, "morton")
