import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, meshgen
from parelag_amd.amge_level import level_from_tets, level_from_hex, box_partition
if sys.argv[1] == "tet":
    seq = meshgen.Sequence.cube456(3); lv = level_from_tets(seq.meshes[0]); part = seq.partition(0)
else:
    n = (30, 110, 42); lv = level_from_hex(n, h=(20., 10., 2.), kinv=np.exp(2 * np.random.default_rng(0).standard_normal(n[0]*n[1]*n[2]))); part = box_partition(n, (2, 2, 2))
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
for stop in (1, 2, 3, 4, 5, 0):
    os.environ["ELAG_HIP_AMGE_STOP"] = str(stop)
    tm = {}
    try:
        amge.coarsen(lv, part, timings=tm)
    except Exception as ex:
        pass
    print(sys.argv[1], "stop", stop, "extension_ms", tm.get("extension_ms"), flush=True)
