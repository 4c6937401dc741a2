#!/usr/bin/env python3
"""Round 5: what a link of GMRES' modified Gram-Schmidt chain costs -- the launch per link (gmres_orth_fused_kernel) against the chain in one
launch (gmres_chain_kernel) -- on a system of the Darcy x3 size without a preconditioner: one restart cycle of m steps has m (m + 1) / 2 links.
    python tools/r05_gmres_chain_probe.py [n] [m]
Under rocprofv3 --kernel-trace (ELAG_HIP_GMRES_GRAPH=0) the chain kernel's dispatches of a cycle have i + 2 hand-offs each, i = 0 .. m - 1:
the slope of duration against i is the cost of a link, the intercept the fixed part (loading w, the norm, writing w and v_{i+1})."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from parelag_amd import linalg as la
n = int(sys.argv[1]) if len(sys.argv) > 1 else 706000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(1)
off = sp.diags([rng.uniform(-1, 1, n - 1), rng.uniform(-1, 1, n - 7)], [1, -7])
A = (off + sp.diags(np.full(n, 2.5))).tocsr()
dA = la.ParallelCSRMatrix.from_scipy(A)
b = rng.uniform(-1, 1, n)
db = la.DeviceVector.from_numpy(b)
res = {}
for chain in ("0", "1", "2"):
    os.environ["ELAG_HIP_GMRES_CHAIN"] = chain
    K = la.KrylovSolver(dA, None, {"Solver name": "GMRES", "Relative tolerance": 1e-300, "Absolute tolerance": 0.0, "Maximum iterations": m, "Restart size": m})
    dx = la.DeviceVector(n, 0.0)
    for _ in range(3):
        K.Mult(db, dx)
    la.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); K.Mult(db, dx); la.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[chain] = (np.median(ts), dx.to_numpy(), K.GetNumIterations(), K.launch_mode())
    print("ELAG_HIP_GMRES_CHAIN=%s: %d steps in %.3f ms (%s)" % (chain, res[chain][2], res[chain][0], res[chain][3]), flush=True)
links = m * (m + 1) // 2
print("n = %d, %d links: %.2f us per link saved by the chain; the three runs' iterates are the same bits: %s" %
      (n, links, (res["0"][0] - res["1"][0]) * 1e3 / links, np.array_equal(res["0"][1], res["1"][1]) and np.array_equal(res["0"][1], res["2"][1])))
