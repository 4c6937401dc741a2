"""cProfile of one full-sequence coarsening (host tables + device calls): python tools/amge_host_profile.py 3 0"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, derham, meshgen
from parelag_amd.amge_level import level_from_hex
from parelag_amd.derham_level import seq_level_from_tets
k, js = int(sys.argv[1]), int(sys.argv[2])
seq = meshgen.Sequence.cube456(k)
lv = seq_level_from_tets(seq.meshes[0], jstart=js)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
derham.coarsen(lv, seq.partition(0))
pr = cProfile.Profile(); pr.enable(); t = time.time(); derham.coarsen(lv, seq.partition(0)); w = time.time() - t; pr.disable()
print("wall %.3f s" % w)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
