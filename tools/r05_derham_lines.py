#!/usr/bin/env python3
"""Per-line wall time inside derham._coarsen_impl (all forms, cube456 x k level 0): python tools/r05_derham_lines.py [k] [function]"""
import os, sys, time, collections, linecache
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, derham, meshgen, derham_level
from parelag_amd.amge_level import level_from_hex
from parelag_amd.derham_level import seq_level_from_tets
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
fname = sys.argv[2] if len(sys.argv) > 2 else "_coarsen_impl"
target = getattr(derham, fname, None) or getattr(derham_level, fname, None) or getattr(amge, fname)
seq = meshgen.Sequence.cube456(k)
lv, part = seq_level_from_tets(seq.meshes[0], jstart=0), seq.partition(0)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
derham.coarsen(lv, part)
acc = collections.defaultdict(float)
state = {"line": None, "t": 0.0}
code = target.__code__
def local(frame, event, arg):
    now = time.perf_counter()
    if state["line"] is not None:
        acc[state["line"]] += now - state["t"]
    state["line"] = frame.f_lineno if event != "return" else None
    state["t"] = time.perf_counter()
    return local
def tracer(frame, event, arg):
    return local if frame.f_code is code else None
sys.settrace(tracer); t0 = time.perf_counter(); derham.coarsen(lv, part); wall = time.perf_counter() - t0; sys.settrace(None)
print("wall %.3f s (traced)" % wall)
for ln, v in sorted(acc.items(), key=lambda kv: -kv[1])[:32]:
    print("%6.1f ms  %4d  %s" % (v * 1e3, ln, linecache.getline(code.co_filename, ln).strip()[:150]))
