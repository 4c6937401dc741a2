"""Wall time spent inside the two coarsening ABI calls (uploads, kernels, downloads) against their kernel time:
    python tools/amge_abi_time.py 3 0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from parelag_amd import _lib, amge, derham, meshgen
from parelag_amd.amge_level import level_from_hex
from parelag_amd.derham_level import seq_level_from_tets

k, js = int(sys.argv[1]), int(sys.argv[2])
seq = meshgen.Sequence.cube456(k)
lv, part = seq_level_from_tets(seq.meshes[0], jstart=js), seq.partition(0)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
L = _lib.init()
acc = {}
calls = []


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t = time.time(); r = self.fn(*a); dt = time.time() - t
        acc[self.name] = acc.get(self.name, 0.0) + dt
        if self.name in ("elag_hip_coarsen_traces", "elag_hip_coarsen_extension"):
            calls.append((self.name[9:], round(dt * 1e3, 2)))
        return r


class Proxy:
    def __getattr__(self, name):
        fn = getattr(L, name)
        return Timed(name, fn) if name.startswith("elag_hip_") else fn


proxy = Proxy()
derham._lib.init = lambda *a, **kw: proxy
derham._lib.load = lambda *a, **kw: proxy
amge._lib.load = lambda *a, **kw: proxy
for rep in range(3):
    acc.clear(); calls.clear(); tm = {}
    t = time.time(); derham.coarsen(lv, part, timings=tm); wall = time.time() - t
print("wall %.3f s, kernels %.1f ms" % (wall, sum(tm.values())))
print("device calls (ms):", calls, {k: round(v, 2) for k, v in tm.items()})
for name, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-44s %.4f s" % (name, v))
