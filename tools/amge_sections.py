#!/usr/bin/env python3
"""Wall time of amge.coarsen (two-form fast path) section by section on the SPE10-size hex grid: wraps the helper functions and
prints cumulative time per helper plus the remainder that is inline numpy in coarsen itself (tuning aid)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, amge_level, derham_level
from parelag_amd.amge_level import box_partition, level_from_hex
n = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (60, 220, 85)
rng = np.random.default_rng(0)
lv = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=np.exp(2 * rng.standard_normal(n[0] * n[1] * n[2])))
part = box_partition(n, (2, 2, 2))
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
amge.coarsen(lv, part)
acc = collections.defaultdict(float)
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(mod, name, g)
for name in ("coarsen_topology", "expand_ranges", "_assembly", "_blocks_to_csr", "_pack_cols", "_take_leading", "_take_leading_rows", "_join_cols",
             "_project", "_spmm_device", "_first_dof_indicator", "ptr_from_counts", "_agglomerate_tables", "_agglomerate_tables_numpy", "_blocks_to_csr_sets"):
    wrap(amge, name)
orig_tab = amge_level.AmgeLevel.elem_dof2_table
def tab(self):
    t = time.perf_counter(); r = orig_tab(self); acc["elem_dof2_table"] += time.perf_counter() - t; return r
amge_level.AmgeLevel.elem_dof2_table = tab
L = amge._lib.init()
for nm in ("elag_hip_coarsen_traces", "elag_hip_coarsen_extension", "elag_hip_host_blocks_to_csr", "elag_hip_host_table_product",
           "elag_hip_host_minimal_intersection_sets", "elag_hip_host_pack_cols", "elag_hip_parcsr_create", "elag_hip_parcsr_spmv"):
    f = getattr(L, nm)
    def g(*a, _f=f, _n=nm):
        t = time.perf_counter(); r = _f(*a); acc[_n] += time.perf_counter() - t; return r
    setattr(L, nm, g)
tm = {}
t0 = time.perf_counter(); amge.coarsen(lv, part, timings=tm); wall = time.perf_counter() - t0
print("wall %.3f s" % wall)
tot = 0.0
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %.3f s" % (k, v)); tot += v
print("  %-28s %.3f s  (inline numpy in coarsen; nested helper time is counted in its parent too)" % ("remainder", wall - tot))
print("stages:", "  ".join("%s %.3f" % (k[5:-2], v) for k, v in tm.items() if k.startswith("host_")))
