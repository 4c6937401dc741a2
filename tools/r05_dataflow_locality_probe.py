#!/usr/bin/env python3
"""Round 5: does the dataflow sweep's hand-off get cheaper when the rows a level hands over sit next to each other?  The operator of cube456 x k
renumbered by (dependency level, row) -- row number = schedule position, so the granules, the vector entries and the rows of a level are
contiguous -- against the operator as it is (same dependency levels, same work; another order of elimination, so other numbers):
    python tools/r05_dataflow_locality_probe.py [k]"""
import os, sys, time, ctypes, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
from parelag_amd import linalg as la, meshgen
from oracle import oracle as orc
src = "void levels(int n, const int *I, const int *J, int *lev) { for (int i = 0; i < n; ++i) { int l = 0; for (int p = I[i]; p < I[i+1]; ++p) { int j = J[p]; if (j < i && lev[j] + 1 > l) l = lev[j] + 1; } lev[i] = l; } }"
d = tempfile.mkdtemp(); open(d + "/lev.c", "w").write(src)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", d + "/lev.so", d + "/lev.c"])
lib = ctypes.CDLL(d + "/lev.so")
A, Ps, _ = meshgen.h1_problem(k)
A = orc.diag_first(orc.csr(A)) if hasattr(orc, "diag_first") else A.tocsr()
n = A.shape[0]
As = A.tocsr().copy(); As.sort_indices()
lev = np.zeros(n, np.int32)
I = As.indptr.astype(np.int32); J = As.indices.astype(np.int32)
lib.levels(n, I.ctypes.data_as(ctypes.c_void_p), J.ctypes.data_as(ctypes.c_void_p), lev.ctypes.data_as(ctypes.c_void_p))
perm = np.argsort(lev, kind="stable")
Ap = As[perm][:, perm].tocsr()
Ap = orc.diag_first(orc.csr(Ap))
rng = np.random.default_rng(0)
f, u0 = rng.standard_normal(n), rng.standard_normal(n)
os.environ["ELAG_HIP_SWEEP_DATAFLOW"] = "2"
for name, M in (("as it is", A), ("renumbered by (level, row)", Ap)):
    dM = la.ParallelCSRMatrix.from_scipy(M)
    S = la.HypreSmootherWrapper(dM, 2, {"Sweeps": 1}); S.iterative_mode = True
    df, du = la.DeviceVector.from_numpy(f), la.DeviceVector.from_numpy(u0)
    for _ in range(5):
        S.Mult(df, du)
    la.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        S.Mult(df, du)
    la.synchronize()
    print("cube456 x%d (%d rows, %d levels) %s: %.1f us per symmetric sweep" % (k, n, int(lev.max()) + 1, name, (time.perf_counter() - t0) / 30 * 1e6), flush=True)
