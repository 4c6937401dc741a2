#!/usr/bin/env python3
"""The N-rank setup path on ONE rank, phase by phase: sharded generator -> ParCSR pieces -> device hierarchy (cube456 x<k>)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
from parelag_amd import distributed, linalg as la, meshgen_sharded as msh
dist = distributed.init(0, 1)
comm = msh.TorchComm(dist._td, 0, 1)
t0 = time.time(); shards = msh.sharded_h1_problem(comm, k, order="tree"); print("generate %.2f s" % (time.time() - t0))
sizes = [int(st[-1]) for st in shards["starts"]]
R = max(1, sum(1 for v in sizes[:-1] if v > (1 << 17)))
for rep in range(2):
    t0 = time.time()
    H, n, shapes, s = distributed.build_hierarchy_sharded(dist, comm, shards, "L1 Jacobi", replicate_from=R)
    print("build %d: %.3f s (R = %d, sizes %s)" % (rep, s, R, sizes))
    del H
pr = cProfile.Profile(); pr.enable()
H, n, shapes, s = distributed.build_hierarchy_sharded(dist, comm, shards, "L1 Jacobi", replicate_from=R)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
rhs, sol = la.DeviceVector(n, 2.0), la.DeviceVector(n, 1.0)
for _ in range(20): H.Mult(rhs, sol)
print("cycle %.4f ms" % H.time_apply(rhs, sol, 100))
