"""cProfile of one full-sequence coarsening (host tables + device calls):
   python tools/amge_host_profile.py tet 3 0      |      python tools/amge_host_profile.py hex 60 220 85 0
   python tools/amge_host_profile.py fast 60 220 85     (the two-form fast path, parelag_amd/amge.py)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, derham, meshgen
from parelag_amd.amge_level import box_partition, level_from_hex
from parelag_amd.derham_level import seq_level_from_hex, seq_level_from_tets
if sys.argv[1] == "tet":
    k, js = int(sys.argv[2]), int(sys.argv[3])
    seq = meshgen.Sequence.cube456(k)
    lv, part = seq_level_from_tets(seq.meshes[0], jstart=js), seq.partition(0)
elif sys.argv[1] == "fast":
    n = tuple(int(v) for v in sys.argv[2:5])
    lv, part = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=np.exp(2 * np.random.default_rng(0).standard_normal(n[0] * n[1] * n[2]))), box_partition(n, (2, 2, 2))
else:
    n, js = tuple(int(v) for v in sys.argv[2:5]), int(sys.argv[5])
    lv, part = seq_level_from_hex(n, h=(20.0, 10.0, 2.0), jstart=js), box_partition(n, (2, 2, 2))
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
run = amge.coarsen if sys.argv[1] == "fast" else derham.coarsen
run(lv, part)
pr = cProfile.Profile(); pr.enable(); t = time.time(); run(lv, part); w = time.time() - t; pr.disable()
print("wall %.3f s" % w)
pstats.Stats(pr).sort_stats("tottime").print_stats(40)
