#!/usr/bin/env python3
"""Round 5: the one-launch PCG pass (cg_dir_amul_dot_kernel) solved over and over -- every iterate must be the first solve's bits and the three-launch
form's (ELAG_HIP_CG_LEAN=0), for every lane count of the row-group product and with the AMGe cycle as preconditioner; device memory must not move:
    python tools/r05_cg_lean_soak.py [solves = 500]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from parelag_amd import linalg as la, meshgen, _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(9)
def make(env, *args):
    saved = {q: os.environ.get(q) for q in env}
    os.environ.update(env)
    try:
        return la.KrylovSolver(*args)
    finally:
        for q, v in saved.items():
            os.environ.pop(q, None) if v is None else os.environ.__setitem__(q, v)
cases = []
A, Ps, _ = meshgen.h1_problem(3)
dA = la.ParallelCSRMatrix.from_scipy(A)
H = la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(P) for P in Ps], "L1 Jacobi")
cases.append(("cube456 x3 + AMGe cycle", dA, H, A.shape[0], {"Solver name": "PCG", "Relative tolerance": 1e-10, "Maximum iterations": 100}))
for n, half in ((9000, 1.0), (30000, 3.0), (131072, 2.0), (20000, 25.0)):
    S0 = sp.random(n, n, density=half / n, random_state=rng)
    M = (S0 + S0.T + sp.diags(np.full(n, 4.0 * half + 2.0))).tocsr()
    M.sort_indices()
    dM = la.ParallelCSRMatrix.from_scipy(M)
    cases.append(("random SPD n=%d, %.0f entries a row" % (n, 2 * half + 1), dM, la.HypreSmootherWrapper(dM, 1, {"Sweeps": 1}), n,
                  {"Solver name": "PCG", "Relative tolerance": 1e-10, "Maximum iterations": 60}))
free0 = None
for name, dM, P, n, params in cases:
    b = rng.uniform(-1, 1, n)
    K0, K1 = make({"ELAG_HIP_CG_LEAN": "0"}, dM, P, params), make({"ELAG_HIP_CG_LEAN": "1"}, dM, P, params)
    x0 = np.zeros(n); K0.Mult(b, x0)
    db, dx = la.DeviceVector.from_numpy(b), la.DeviceVector(n, 0.0)
    bad = 0
    t0 = time.perf_counter()
    for i in range(N):
        K1.Mult(db, dx)
        if i % 25 == 0 or i == N - 1:
            bad += 0 if np.array_equal(dx.to_numpy(), x0) else 1
    la.synchronize()
    dt = time.perf_counter() - t0
    print("%-40s: %d solves of %d passes in %.2f s, %d of the checked ones differ from the three-launch solve" % (name, N, K1.GetNumIterations(), dt, bad), flush=True)
    assert bad == 0
import resource
print("rss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024)
