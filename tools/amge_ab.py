#!/usr/bin/env python3
"""Wall time of one finest-level coarsening for several host-thread counts, in ONE process (box-to-box noise is +-20 %):
fast = two-form path on the SPE10-size hex grid, tet = all forms on cube456 x3, hex = forms 2..3 of the general path on the hex grid.
usage: tools/amge_ab.py [reps] [threads...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import _lib, amge, derham, meshgen
from parelag_amd.amge_level import box_partition, level_from_hex
from parelag_amd.derham_level import seq_level_from_hex, seq_level_from_tets
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
threads = [int(v) for v in sys.argv[2:]] or [1, 4, 16]
n = (60, 220, 85)
cases = {}
lvf = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=np.exp(2 * np.random.default_rng(0).standard_normal(n[0] * n[1] * n[2])))
partf = box_partition(n, (2, 2, 2))
cases["fast hex 60x220x85"] = lambda: amge.coarsen(lvf, partf)
seq = meshgen.Sequence.cube456(3)
lvt, partt = seq_level_from_tets(seq.meshes[0], jstart=0), seq.partition(0)
cases["general tet cube456x3 forms 0-3"] = lambda: derham.coarsen(lvt, partt)
lvh = seq_level_from_hex(n, h=(20.0, 10.0, 2.0), jstart=2)
cases["general hex 60x220x85 forms 2-3"] = lambda: derham.coarsen(lvh, partf)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
lib = _lib.load()
acc = {}
if os.environ.get("AB_NATIVE"):                      # cumulative time inside every native host function, per case and thread count
    for nm in _lib.ABI_SYMBOLS:
        if nm.startswith("elag_hip_host_") and "set_" not in nm:
            def g(*a, _f=getattr(lib, nm), _n=nm):
                t = time.perf_counter(); r = _f(*a); acc[_n] = acc.get(_n, 0.0) + time.perf_counter() - t; return r
            setattr(lib, nm, g)
for name, run in cases.items():
    run()
    out = []
    for t in threads:
        lib.elag_hip_host_set_threads(t)
        best = 1e9
        acc.clear()
        for _ in range(reps):
            t0 = time.perf_counter(); run(); best = min(best, time.perf_counter() - t0)
        out.append("%d threads %.3f s" % (t, best))
        if acc:
            print("   [%s, %d threads] native ms per run: %s" % (name, t, "  ".join("%s %.1f" % (k[14:], 1e3 * v / reps) for k, v in sorted(acc.items(), key=lambda kv: -kv[1]))))
    lib.elag_hip_host_set_threads(0)
    print("%-36s %s" % (name, "   ".join(out)), flush=True)
