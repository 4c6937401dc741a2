#!/usr/bin/env python3
"""Per-line wall time inside amge.coarsen (two-form fast path) on the SPE10-size hex grid: a trace function on that one frame
charges the time between two line events to the line that was running (tuning aid; prints the 30 most expensive lines)."""
import os, sys, time, collections, linecache
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge
from parelag_amd.amge_level import box_partition, level_from_hex
n = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (60, 220, 85)
target = getattr(amge, sys.argv[4]) if len(sys.argv) > 4 else amge.coarsen
rng = np.random.default_rng(0)
lv = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=np.exp(2 * rng.standard_normal(n[0] * n[1] * n[2])))
part = box_partition(n, (2, 2, 2))
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
amge.coarsen(lv, part)
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
sys.settrace(tracer); t0 = time.perf_counter(); amge.coarsen(lv, part); wall = time.perf_counter() - t0; sys.settrace(None)
print("wall %.3f s (traced)" % wall)
for ln, v in sorted(acc.items(), key=lambda kv: -kv[1])[:30]:
    print("%6.1f ms  %4d  %s" % (v * 1e3, ln, linecache.getline(code.co_filename, ln).strip()[:130]))
