#!/usr/bin/env python3
"""Wall time of the blocked (Darcy, cube456 x k) hierarchy build, repeated: python tools/r04_darcy_setup_probe.py [k] [reps].
Under `rocprofv3 --kernel-trace --stats -- python3 tools/r04_darcy_setup_probe.py` the kernel statistics are this build's."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dar = meshgen.darcy_problem(k)
Ps = dar["Ps"]
n0, n1 = dar["M"].shape[0], dar["B"].shape[0]
bop = la.MfemBlockOperator([0, n0, n0 + n1])
for (i, j), M in {(0, 0): dar["M"], (0, 1): dar["Bt"], (1, 0): dar["B"]}.items():
    bop.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(M))
dPs = [[la.ParallelCSRMatrix.from_scipy(P) for P in Pl] for Pl in Ps]
la.synchronize()
for r in range(reps):
    t0 = time.perf_counter()
    H = la.buildBlockedHierarchyFromDeRhamSequence(bop, dPs, block_smoother="Block Jacobi", a00_inverse="L1 Jacobi", a11_inverse="L1 Jacobi",
                                                   use_negative_s=True, use_graph=False)
    la.synchronize()
    print("build %d: %.2f ms" % (r, (time.perf_counter() - t0) * 1e3), flush=True)
    del H
