#!/usr/bin/env python3
"""cProfile of derham_dist.coarsen_allgather (one rank = the whole level) on the SPE10-size hex grid: restriction of the level to the rank's
agglomerates, Coarsen on the piece, renumbering, merge (tuning aid).  usage: tools/amge_dist_profile.py [nx ny nz]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, derham_dist
from parelag_amd.amge_level import box_partition, level_from_hex
from parelag_amd.derham_level import seq_level_from_hex
n = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (60, 220, 85)
lv = seq_level_from_hex(n, h=(20.0, 10.0, 2.0), jstart=2)
part = np.asarray(box_partition(n, (2, 2, 2)), dtype=np.int64)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
import torch.distributed as td
from parelag_amd.distributed import free_port
td.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % free_port(), rank=0, world_size=1)
run = lambda: derham_dist.coarsen_allgather(lv, part, 0, 1, owner=np.zeros(int(part.max()) + 1, dtype=np.int64))
run()
pr = cProfile.Profile(); pr.enable(); t = time.time(); run(); w = time.time() - t; pr.disable()
print("wall %.3f s" % w)
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
td.destroy_process_group()
