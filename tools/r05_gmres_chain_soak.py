#!/usr/bin/env python3
"""Round 5 soak: GMRES with a step's Gram-Schmidt chain in one launch (gmres_chain_kernel) solving the same system N times -- the arithmetic is
deterministic, so every solve must return the first one's iterate bit for bit with the same iteration count (a stale or torn hand-off would show
as a different vector, a late workgroup as a give-up and a failed call); both register forms; then the launch-per-link chain once: the same bits.   python tools/r05_gmres_chain_soak.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
from parelag_amd import linalg as la
rng = np.random.default_rng(4)
for n, restart, iters in ((706000, 50, 60), (40000, 20, 60), (1300000, 30, 40)):
    off = sp.diags([rng.uniform(-1, 1, n - 1), rng.uniform(-1, 1, n - 7), rng.uniform(-1, 1, n - 300)], [1, -7, 300])
    A = (off + sp.diags(np.full(n, 3.2))).tocsr()
    dA = la.ParallelCSRMatrix.from_scipy(A)
    b = rng.uniform(-1, 1, n)
    db = la.DeviceVector.from_numpy(b)
    params = {"Solver name": "GMRES", "Relative tolerance": 1e-12, "Maximum iterations": iters, "Restart size": restart}
    ref = None
    for chain, reps in (("1", N), ("2", max(1, N // 4)), ("0", 1)):
        os.environ["ELAG_HIP_GMRES_CHAIN"] = chain
        K = la.KrylovSolver(dA, None, params)
        t0 = time.time(); bad = 0
        for i in range(reps):
            dx = la.DeviceVector(n, 0.0)
            K.Mult(db, dx)
            if i % 20 == 0 or i == reps - 1:
                x = dx.to_numpy()
                if ref is None:
                    ref = (x.copy(), K.GetNumIterations())
                elif K.GetNumIterations() != ref[1] or not np.array_equal(x, ref[0]):
                    bad += 1
        print("n = %d, GMRES(%d), ELAG_HIP_GMRES_CHAIN=%s (%s): %d solves of %d iterations in %.1f s, %d of the checked ones differ from the first solve" %
              (n, restart, chain, K.launch_mode(), reps, ref[1], time.time() - t0, bad), flush=True)
        assert bad == 0
print("rss MB", int(open("/proc/self/statm").read().split()[1]) * 4096 // 2**20)
