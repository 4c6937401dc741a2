#!/usr/bin/env python3
"""H(div) + Hiptmair cube456 x<k>: cycle rate and kernels per cycle (for A/B runs under environment switches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen, _lib
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
form = sys.argv[2] if len(sys.argv) > 2 else "hdiv"
A, Ps, Ds, _ = (meshgen.hdiv_problem if form == "hdiv" else meshgen.hcurl_problem)(k)
dA = la.ParallelCSRMatrix.from_scipy(A); dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]; dDs = [la.ParallelCSRMatrix.from_scipy(D) for D in Ds[:len(Ps)]]
for rep in range(2):
    la.synchronize(); t0 = time.perf_counter()
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi", Ds=dDs, aux_smoother="L1 Jacobi")
    la.synchronize(); setup = time.perf_counter() - t0
n = A.shape[0]
rhs, sol = la.DeviceVector(n, 2.0), la.DeviceVector(n, 1.0)
for _ in range(50): H.Mult(rhs, sol)
ms = H.time_apply(rhs, sol, 300)
x = sol.to_numpy()
print("%s x%d: %.4f ms per cycle = %.0f V-cycles/s; %s; setup %.2f ms %s; checksum %.15e" % (form, k, ms, 1e3 / ms, H.launch_mode(), setup * 1e3,
      {k_: round(v, 2) for k_, v in H.setup_breakdown().items()}, float(np.linalg.norm(x))))
