#!/usr/bin/env python3
"""Prototype timing for the three-stage ("flattened") form of a launch-bound l1-Jacobi V-cycle: are the composed operators cheap
enough?  Times every segment of every stage alone (dependent back-to-back launches of the same product) at several lane counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from parelag_amd import linalg as la, meshgen, _lib

_lib.init()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
A, Ps, _ = meshgen.h1_problem(k)
As = [A.tocsr()]
for P in Ps:
    Ac = (P.T @ As[-1] @ P).tocsr()
    zr = np.asarray(abs(Ac).sum(axis=1)).ravel() == 0
    As.append((Ac + sp.diags(zr.astype(float))).tocsr())           # FixZeroRows
start = int(sys.argv[2]) if len(sys.argv) > 2 else 0
As, Ps = As[start:], Ps[start:]
nl = len(As)
G, U, V = [], [], []
for l in range(nl - 1):
    A_, P_ = As[l], Ps[l]
    l1 = np.asarray(abs(A_).sum(axis=1)).ravel()
    Wl = sp.diags(np.where(l1 > 0, 1.0 / np.where(l1 > 0, l1, 1), 0.0)); n = A_.shape[0]
    G.append((P_.T @ (sp.eye(n) - A_ @ Wl)).tocsr())
    S = (A_ @ P_ - sp.diags(l1) @ P_).tocsr()
    U.append((Wl @ (2 * sp.eye(n) - A_ @ Wl)).tocsr())
    V.append((-Wl @ S).tocsr())
Gc, Vc = [G[0]], [V[0]]
for l in range(1, nl - 1):
    Gc.append((G[l] @ Gc[-1]).tocsr()); Vc.append((Vc[-1] @ V[l]).tocsr())
Binv = sp.csr_matrix(np.linalg.inv(As[-1].toarray()))
segs = {"s1:U0": U[0]}
for l in range(nl - 1): segs["s1:Gc%d" % (l + 1)] = Gc[l]
for l in range(1, nl - 1): segs["s2:U%d" % l] = U[l]
segs["s2:Binv"] = Binv
for l in range(nl - 1): segs["s3:Vc%d" % (l + 1)] = Vc[l]
for name, M in segs.items():
    M = M.tocsr(); M.sort_indices()
    d = la.ParallelCSRMatrix.from_scipy(M)
    x = la.DeviceVector.from_numpy(np.random.default_rng(1).uniform(-1, 1, M.shape[1])); y = la.DeviceVector(M.shape[0])
    res = []
    for lanes in (1, 2, 4, 8, 16, 32, 64):
        if lanes > 4 * M.nnz / M.shape[0] + 1: break
        d.set_spmv_lanes(lanes)
        res.append("%d:%.2f" % (lanes, la.time_kernel(d, "spmv", x, y, reps=200) * 1e3))
    print("%-10s %7d x %7d nnz %8d avg %7.1f | us by lanes: %s" % (name, M.shape[0], M.shape[1], M.nnz, M.nnz / M.shape[0], " ".join(res)), flush=True)
