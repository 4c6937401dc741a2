"""Wall time per rank of the sharded coarsening (parelag_amd/derham_dist.py) on cube456 x k, all forms, for N = 1, 2, 4, 8
ranks run one after the other on this GPU (so the numbers are per-rank costs, not a multi-GPU measurement; the exchange is the
in-process stand-in -- in a real run it is one MAX all-reduce of ~n_coarse_entities int64).
    python tools/amge_dist_time.py 3 0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import amge, derham, derham_dist as dd, meshgen, partition
from parelag_amd.amge_level import level_from_hex
from parelag_amd.derham_level import seq_level_from_tets

k, js = int(sys.argv[1]), int(sys.argv[2])
seq = meshgen.Sequence.cube456(k)
lv = seq_level_from_tets(seq.meshes[0], jstart=js)
part = seq.partition(0)
amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
t = time.time(); single = derham.coarsen(lv, part); t_single = time.time() - t
t = time.time(); single = derham.coarsen(lv, part); t_single = min(t_single, time.time() - t)
t = time.time(); topo = derham.coarsen_topology(lv, part); t_topo = time.time() - t
print("cube456 x%d forms %d..3: %d elements, %d agglomerates; single rank %.3f s (topology %.3f s of it)" % (k, js, lv.n[0], topo["n"][0], t_single, t_topo), flush=True)
for N in (2, 4, 8):
    eparts = partition.element_partitions(seq, N)
    owner = np.zeros(topo["n"][0], dtype=np.int64)
    owner[part] = eparts[0]                                   # derefinement agglomerates are rank-local
    reported, walls, sizes = [], [], []
    for r in range(N):
        dd.coarsen_rank(lv, part, r, N, owner=owner, topo=topo, exchange=lambda a: (reported.append(a.copy()), a)[1])
    glob = np.max(np.stack(reported), axis=0)
    pieces = []
    for r in range(N):
        t = time.time()
        R = dd.restrict_level(lv, topo, np.nonzero(owner == r)[0]); t_r = time.time() - t
        p = dd.coarsen_rank(lv, part, r, N, owner=owner, topo=topo, exchange=lambda a: glob)
        walls.append(time.time() - t - t_r); sizes.append((R.level.n[0], round(t_r, 3)))
        pieces.append(p)
    merged = dd.merge_pieces(lv, topo, pieces, atol=1e-10)
    err = max(abs(merged["P"][j] - single["P"][j]).max() for j in range(js, 4))
    dup = sum(p.csel[1].size for p in pieces) / topo["n"][1]
    print("N=%d: per-rank wall (restriction + numeric + renumbering) max %.3f s mean %.3f s  -> %.2fx of single;  + replicated topology %.3f s;"
          "  facets held %.3fx;  |P_merged - P_single| %.2e;  (elements, restriction s) %s" %
          (N, max(walls), np.mean(walls), t_single / max(walls), t_topo, dup, err, sizes), flush=True)

# ---- the whole hierarchy per rank: every further level goes on from the coarse level the rank already holds ----
print("\nall %d coarsenings per rank (held coarse levels, integer topology replicated):" % k, flush=True)
t = time.time(); cur = lv
for l in range(k):
    cur = derham.coarsen(cur, seq.partition(l))["coarse"]
t_single_all = time.time() - t
for N in (2, 4, 8):
    eparts = partition.element_partitions(seq, N)
    per_rank = np.zeros(N)
    held = [None] * N
    level_int, topo_time = lv, 0.0
    for l in range(k):
        part_l = seq.partition(l)
        t = time.time(); topo_l = derham.coarsen_topology(level_int, part_l); topo_time += time.time() - t
        owner = np.zeros(topo_l["n"][0], dtype=np.int64); owner[part_l] = eparts[l]
        reported = []
        for r in range(N):
            dd.coarsen_rank(level_int, part_l, r, N, owner=owner, topo=topo_l, exchange=lambda a: (reported.append(a.copy()), a)[1], held=held[r])
        glob = np.max(np.stack(reported), axis=0)
        new = []
        for r in range(N):
            t = time.time()
            new.append(dd.coarsen_rank(level_int, part_l, r, N, owner=owner, topo=topo_l, exchange=lambda a: glob, held=held[r]))
            per_rank[r] += time.time() - t
        held = new
        level_int = dd.integer_level(topo_l, lv.jstart)
    print("N=%d: numeric per rank max %.3f s mean %.3f s + replicated topologies %.3f s = %.3f s  (single rank, all levels: %.3f s -> %.2fx)" %
          (N, per_rank.max(), per_rank.mean(), topo_time, per_rank.max() + topo_time, t_single_all, t_single_all / (per_rank.max() + topo_time)), flush=True)
