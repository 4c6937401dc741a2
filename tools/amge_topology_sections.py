"""Wall time of the pieces of amge.coarsen_topology on the SPE10-size hex grid (host only; tuning aid)."""
import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import amge, _lib
from parelag_amd.amge_level import box_partition, level_from_hex, ptr_from_counts
from parelag_amd.derham_level import _product
n=(60,220,85)
t=time.perf_counter(); lv = level_from_hex(n, h=(20.0, 10.0, 2.0)); print("level_from_hex", time.perf_counter()-t)
part = box_partition(n, (2, 2, 2))
for rep in range(3):
    T=[time.perf_counter()]
    part = np.asarray(part, dtype=np.int64)
    nAE = int(part.max()) + 1
    order = np.argsort(part, kind="stable"); T.append(time.perf_counter())
    ae_ptr = ptr_from_counts(np.bincount(part, minlength=nAE)); T.append(time.perf_counter())
    fc_ptr, f, s = _product(ae_ptr, order, lv.ef_ptr, lv.ef_idx, np.ones(order.size), lv.ef_sgn); T.append(time.perf_counter())
    fcp, fc, sc = np.ascontiguousarray(fc_ptr, dtype=np.int64), np.ascontiguousarray(f, dtype=np.int64), np.ascontiguousarray(s, dtype=np.float64)
    attr32 = np.ascontiguousarray(lv.facet_attr, dtype=np.int32)
    mis_of, ori, nset = np.empty(lv.nF, dtype=np.int64), np.empty(lv.nF), np.zeros(1, dtype=np.int64)
    _lib.check(_lib.load().elag_hip_host_minimal_intersection_sets(nAE, amge._ptr(fcp), amge._ptr(fc), amge._ptr(sc), lv.nF, amge._ptr(attr32), amge._ptr(mis_of), amge._ptr(ori), amge._ptr(nset))); T.append(time.perf_counter())
    nAF = int(nset[0])
    fid = np.nonzero(mis_of >= 0)[0]
    o2 = np.argsort(mis_of[fid], kind="stable")
    af_ptr = ptr_from_counts(np.bincount(mis_of[fid], minlength=nAF))
    af_idx = fid[o2]; T.append(time.perf_counter())
    member = mis_of >= 0
    B_ptr, bc, bs = _product(fc_ptr, f, ptr_from_counts(member.astype(np.int64)), mis_of[member], s, ori[member]); T.append(time.perf_counter())
    print(["%.3f"%(b-a) for a,b in zip(T,T[1:])])
t=time.perf_counter(); amge.coarsen_topology(lv, part); print("topology", time.perf_counter()-t)
