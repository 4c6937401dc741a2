"""Where the time of one AMGe coarsening goes (host tables vs device kernels).  python tools/amge_time.py tet 3 | hex 60 220 85"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parelag_amd import amge, meshgen
from parelag_amd.amge_level import level_from_tets, level_from_hex, box_partition

kind = sys.argv[1]
t0 = time.time()
if kind == "seq":
    # whole sequence (forms jstart..3) on cube456 x k:  python tools/amge_time.py seq 3 0
    from parelag_amd import derham
    from parelag_amd.derham_level import seq_level_from_tets
    k, js = int(sys.argv[2]), int(sys.argv[3])
    seq = meshgen.Sequence.cube456(k)
    lv = seq_level_from_tets(seq.meshes[0], jstart=js)
    print("generate %.2fs  n=%s dofs=%s" % (time.time() - t0, lv.n, [lv.ndof(j) for j in range(js, 4)]), flush=True)
    amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
    for rep in range(2):
        cur = lv
        for l in range(k):
            tm = {}
            t0 = time.time()
            res = derham.coarsen(cur, seq.partition(l), timings=tm)
            wall = time.time() - t0
            c = res["coarse"]
            print("rep %d level %d: n %s -> %s dofs -> %s  wall %.3fs  kernel ms total %.3f  %s" %
                  (rep, l, cur.n, c.n, [c.ndof(j) for j in range(js, 4)], wall, sum(tm.values()), {a: round(b, 3) for a, b in tm.items()}), flush=True)
            cur = c
    sys.exit(0)
if kind == "tet":
    k = int(sys.argv[2])
    seq = meshgen.Sequence.cube456(k)
    lv = level_from_tets(seq.meshes[0])
    parts = [seq.partition(l) for l in range(k)]
else:
    n = tuple(int(v) for v in sys.argv[2:5])
    rng = np.random.default_rng(0)
    lv = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=np.exp(2 * rng.standard_normal(n[0] * n[1] * n[2])))
    parts, m = [], n
    while min(m) >= 2 and len(parts) < 4:
        parts.append(box_partition(m, (2, 2, 2)))
        m = tuple(-(-v // 2) for v in m)
print("generate %.2fs  nE=%d ndof2=%d" % (time.time() - t0, lv.nE, lv.ndof2), flush=True)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))    # warm-up (library init)
for rep in range(2):
    cur = lv
    for l, part in enumerate(parts):
        tm = {}
        t0 = time.time()
        res = amge.coarsen(cur, part, timings=tm)
        wall = time.time() - t0
        c = res["coarse"]
        print("rep %d level %d: nAE=%d nAF=%d dof2 %d->%d  wall %.3fs  kernels: traces3 %.3f ms traces2 %.3f ms extension %.3f ms  null=%d" %
              (rep, l, c.nE, c.nF, cur.ndof2, c.ndof2, wall, tm["traces3_ms"], tm["traces2_ms"], tm["extension_ms"], res["n_null"].sum()), flush=True)
        cur = c
