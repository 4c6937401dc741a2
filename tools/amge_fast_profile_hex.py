import cProfile, pstats, sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from parelag_amd import amge
from parelag_amd.amge_level import box_partition, level_from_hex
n = (60, 220, 85)
rng = np.random.default_rng(0)
lv = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=np.exp(2 * rng.standard_normal(n[0] * n[1] * n[2])))
part = box_partition(n, (2, 2, 2))
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
amge.coarsen(lv, part)
pr = cProfile.Profile(); pr.enable(); t = time.time(); amge.coarsen(lv, part); w = time.time() - t; pr.disable()
print("wall %.3f s" % w)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
