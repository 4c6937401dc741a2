#!/usr/bin/env python3
"""Round 5: the Gauss-Seidel dataflow sweep (sweep.hip, sweep_dataflow_kernel) against the launch-per-level sweeps and the oracle's sequential
loop on the level operators of cube456 x k:  python tools/r05_dataflow_probe.py [k] [type]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rtype = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ.setdefault("ELAG_HIP_SWEEP_VERBOSE", "1")
from parelag_amd import linalg as la, meshgen
from oracle import oracle as orc
A, Ps, _ = meshgen.h1_problem(k)
from tests.helpers import oracle_h1_hierarchy
ops = [orc.diag_first(orc.csr(M)) for M in oracle_h1_hierarchy(A, Ps)["As"][:3]]
rng = np.random.default_rng(3)
for lvl, Ad in enumerate(ops):
    n = Ad.shape[0]
    dA = la.ParallelCSRMatrix.from_scipy(Ad)
    f, u0 = rng.standard_normal(n), rng.standard_normal(n)
    res = {}
    for mode in ("0", "2"):
        os.environ["ELAG_HIP_SWEEP_DATAFLOW"] = mode
        t0 = time.perf_counter()
        S = la.HypreSmootherWrapper(dA, rtype, {"Sweeps": 1})
        la.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        S.iterative_mode = True
        u = u0.copy(); S.Mult(f, u)
        df, du = la.DeviceVector.from_numpy(f), la.DeviceVector.from_numpy(u0)
        for _ in range(5):
            S.Mult(df, du)
        la.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            S.Mult(df, du)
        la.synchronize()
        res[mode] = (u, (time.perf_counter() - t0) / 50 * 1e6, setup_ms)
    ref = orc.OSmoother(orc.OParCSR(Ad), type=rtype, sweeps=1); ref.iterative_mode = True
    ur = u0.copy(); ref.Mult(f, ur)
    print("level %d n=%d nnz=%d: level launches %.1f us (setup %.2f ms), dataflow %.1f us (setup %.2f ms); equal to the oracle: %s %s; to each other: %s" %
          (lvl, n, Ad.nnz, res["0"][1], res["0"][2], res["2"][1], res["2"][2], np.array_equal(res["0"][0], ur), np.array_equal(res["2"][0], ur),
           np.array_equal(res["0"][0], res["2"][0])), flush=True)
