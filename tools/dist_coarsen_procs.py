"""The sharded coarsening with REAL processes: N ranks (default 2) on this box's one GPU, gloo for the integer exchange and the
all-gather of the pieces, every rank's numeric stages on the device.  The parent never touches the GPU; it only starts the
ranks as child processes and collects their exit codes.
    python tools/dist_coarsen_procs.py [nranks] [refines]"""
import os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, k):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from parelag_amd import _lib, amge, derham, derham_dist as dd, meshgen, partition
    from parelag_amd.amge_level import level_from_hex
    from parelag_amd.derham_level import seq_level_from_tets
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _lib.init(0)
    seq = meshgen.Sequence.cube456(k)
    lv, part = seq_level_from_tets(seq.meshes[0], jstart=0), seq.partition(0)
    amge.coarsen(level_from_hex((2, 2, 2)), np.zeros(8, np.int32))
    owner = np.zeros(int(part.max()) + 1, dtype=np.int64)
    owner[part] = partition.element_partitions(seq, world)[0]
    dist.barrier()
    t = time.time()
    full = dd.coarsen_allgather(lv, part, rank, world, owner=owner)
    dist.barrier()
    t_dist = time.time() - t
    t = time.time()
    single = derham.coarsen(lv, part)
    t_single = time.time() - t
    err = max(abs(full["P"][j] - single["P"][j]).max() for j in range(4))
    derr = max(abs(full["coarse"].D[j] - single["coarse"].D[j]).max() for j in range(3))
    merr = max(np.abs(full["coarse"].M[j][0][1] - single["coarse"].M[j][0][1]).max() for j in range(4))
    print("rank %d/%d: cube456 x%d, %d of %d agglomerates; sharded + all-gather + merge %.3f s, single-rank %.3f s (both with %d processes "
          "sharing one GPU);  |P - P_single| %.2e  |D_c - D_c,single| %.2e  |M_c - M_c,single| %.2e" %
          (rank, world, k, full["piece"].agglomerates.size, full["topo"]["n"][0], t_dist, t_single, world, err, derr, merr), flush=True)
    # ---- every level, pieces kept distributed: the rank goes on from the coarse level it holds; only integers are exchanged ----
    eparts = partition.element_partitions(seq, world)
    dist.barrier()
    t = time.time()
    level_int, held = lv, None
    for l in range(k):
        part_l = seq.partition(l)
        topo_l = derham.coarsen_topology(level_int, part_l)
        own = np.zeros(topo_l["n"][0], dtype=np.int64)
        own[part_l] = eparts[l]
        held = dd.coarsen_rank(level_int, part_l, rank, world, owner=own, topo=topo_l, held=held)
        level_int = dd.integer_level(topo_l, lv.jstart)
    dist.barrier()
    t_chain = time.time() - t
    t = time.time()
    cur = single["coarse"]
    for l in range(1, k):
        last = derham.coarsen(cur, seq.partition(l))
        cur = last["coarse"]
    t_rest = time.time() - t
    Pl = last["P"][2].tocsr()[held.fine_dofs[2]] if k > 1 else single["P"][2].tocsr()[held.fine_dofs[2]]
    cerr = abs(held.P[2] - Pl).max()
    print("rank %d/%d: all %d levels with held coarse levels %.3f s (single rank: %.3f s);  |P_2 rows on the last level - single| %.2e" %
          (rank, world, k, t_chain, t_single + t_rest, cerr), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if max(err, derr, merr, cerr) < 1e-10 else 1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), str(k)], env=env) for r in range(world)]
    codes = [p.wait(timeout=900) for p in procs]
    print("exit codes", codes, flush=True)
    sys.exit(max(codes))
