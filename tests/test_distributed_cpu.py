"""N > 1 path on CPU: partition / comm-package logic in one process, and a world_size-2 gloo run of the worker."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from parelag_amd import meshgen, partition
from parelag_amd.distributed import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nparts", [1, 2, 3, 8])
def test_split_rows_reassembles_products(h1_k2, nparts):
    A, Ps, seq = h1_k2
    d = partition.distributed_h1_hierarchy(A, Ps, seq, nparts)
    rng = np.random.default_rng(0)
    for l, lev in enumerate(d["levels"]):
        Ag = d["A_global"][l]
        W = orc.InProcessWorld(lev["A_parts"])
        x = rng.standard_normal(Ag.shape[1])
        assert np.allclose(W.matvec(x), Ag @ x, rtol=1e-12, atol=1e-13)
        assert np.allclose(W.matvec(x, trans=True), Ag.T @ x, rtol=1e-12, atol=1e-13)
        # block numbering is a permutation of the natural one and keeps the operator
        o = lev["order"]
        assert np.array_equal(np.sort(o), np.arange(Ag.shape[0]))
        if l == 0:
            assert abs(Ag - sp.csr_matrix(A)[o][:, o]).max() == 0
        if l > 0:
            Pg = d["P_global"][l - 1]
            Wp = orc.InProcessWorld(lev["P_parts"])
            xc, xf = rng.standard_normal(Pg.shape[1]), rng.standard_normal(Pg.shape[0])
            assert np.allclose(Wp.matvec(xc), Pg @ xc, rtol=1e-12, atol=1e-13)
            assert np.allclose(Wp.matvec(xf, trans=True), Pg.T @ xf, rtol=1e-12, atol=1e-13)
        for p in lev["A_parts"]:
            # comm package invariants of hypre_MatvecCommPkgCreate
            assert np.all(np.diff(p["col_map_offd"]) > 0)
            assert p["recv_starts"][-1] == p["offd"].shape[1]
            assert len(p["send_elmts"]) == p["send_starts"][-1]
            assert np.all(p["send_elmts"] < p["diag"].shape[1])


def test_elements_stay_rank_local_under_derefinement(h1_k2):
    _, _, seq = h1_k2
    parts = partition.element_partitions(seq, 4)
    for l in range(len(parts) - 1):
        fine, coarse = parts[l], parts[l + 1]
        assert np.array_equal(fine, np.repeat(coarse, 8))          # p[j] = j/8 agglomerates never straddle ranks
    assert set(np.unique(parts[0])) == {0, 1, 2, 3}


def test_rcb_is_balanced():
    pts = np.random.default_rng(1).uniform(size=(456, 3))
    for n in (2, 3, 8):
        c = np.bincount(partition.rcb_partition(pts, n), minlength=n)
        assert c.max() - c.min() <= 2


def test_world_size_2_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_WORKER_OK 2" in r.stdout


@pytest.mark.parametrize("nparts", [2, 3, 5])
def test_distributed_darcy_blocks_reassemble(nparts):
    """faces / elements are owned per rank, rectangular blocks (B: elements x faces) carry separate row / column
    partitions; every block and interpolator piece reproduces the global product through the oracle's exchange"""
    from parelag_amd import meshgen
    dar = meshgen.darcy_problem(0, seq=meshgen.Sequence.kuhn_cube(2, 1))
    dd = partition.distributed_darcy(dar, nparts)
    rng = np.random.default_rng(1)
    for k, parts in dd["blocks"].items():
        G = dd["global_blocks"][k]
        W = orc.InProcessWorld(parts)
        x = rng.standard_normal(G.shape[1]); y = rng.standard_normal(G.shape[0])
        assert np.allclose(W.matvec(x), G @ x, rtol=1e-12, atol=1e-13)
        assert np.allclose(W.matvec(y, trans=True), G.T @ y, rtol=1e-12, atol=1e-13)
    (o2, o3), (c2, c3) = dd["order"][0], dd["order"][1]
    for parts, P, fo, co in ((dd["P"][0][0], dar["Ps"][0][0], o2, c2), (dd["P"][0][1], dar["Ps"][0][1], o3, c3)):
        Pg = sp.csr_matrix(P)[fo][:, co]
        xc = rng.standard_normal(Pg.shape[1])
        assert np.allclose(orc.InProcessWorld(parts).matvec(xc), Pg @ xc, rtol=1e-12, atol=1e-13)
    # ownership: an element's rank owns it; a face belongs to the lowest rank among its elements
    ep = partition.element_partitions(dar["seq"], nparts)[0]
    own2 = partition.entity_owners(dar["seq"].meshes[0], ep, 2)
    mesh = dar["seq"].meshes[0]
    assert np.all(own2[mesh.tet_faces] <= ep[:, None])
    assert np.array_equal(partition.entity_owners(mesh, ep, 3), ep)


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher around it (WORLD_SIZE unset) must start two child ranks itself -- before
    anything touches a GPU -- and rank 0 must report n_gpus = 2 (the driver's N=1 invocation style applied to N > 1)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"], capture_output=True, text=True,
                       env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                     # stdout carries exactly the JSON line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["launch_dry_run"] is True
    assert sorted(c["rank"] for c in rec["ranks"]) == [0, 1]
    assert all(c["world_size"] == 2 for c in rec["ranks"])
    assert len({c["pid"] for c in rec["ranks"]}) == 2                  # two child processes, neither of them this launcher
    assert len({c["ppid"] for c in rec["ranks"]}) == 1


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-dry-run"], capture_output=True, text=True,
                       env=env, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.parametrize("nparts", [2, 3, 4])
def test_hex_grid_row_blocks(nparts):
    """BASELINE configs[4] on several ranks: the Cartesian grid of MultigridTestSPE10.cpp:194-208 cut into row blocks of whole
    coarsest boxes; AMGe coarse spaces with several dofs per agglomerated entity (heterogeneous coefficient: null-space dofs,
    element-interior dofs of form 2).  Ownership comes from the levels' own tables, every block / interpolator piece reproduces
    the global product through the oracle's exchange, and no agglomerate of any level straddles two ranks."""
    from oracle import derham as od
    from parelag_amd.amge_level import box_partition
    from parelag_amd.derham_level import seq_level_from_hex
    n = (8, 8, 4)
    kinv = np.exp(1.5 * np.random.default_rng(5).standard_normal(n[0] * n[1] * n[2]))
    lv = seq_level_from_hex(n, h=(20.0, 10.0, 2.0), jstart=2, kinv=kinv)
    parts = [box_partition(n, (2, 2, 2)).astype(np.int64), box_partition((4, 4, 2), (2, 2, 2)).astype(np.int64)]
    levels, Ps = [lv], []
    for part in parts:
        res = od.coarsen(levels[-1], part)
        Ps.append((res["P"][2].tocsr(), res["P"][3].tocsr()))
        levels.append(res["coarse"])
    assert levels[1].ndof(2) > levels[1].n[1]                       # more than one dof per coarse facet / interior dofs: not mesh-countable
    M, W = lv.mass(2), lv.mass(3)
    B = (W @ lv.D[2]).tocsr()
    dar = dict(M=M, B=B, Bt=B.T.tocsr(), Ps=Ps)
    eown = partition.hex_element_owners(n, parts, nparts)
    assert [len(e) for e in eown] == [L.n[0] for L in levels]
    for l, part in enumerate(parts):
        assert np.array_equal(eown[l], eown[l + 1][part])            # agglomerates are rank-local, level after level
    assert set(np.unique(eown[-1])) == set(range(nparts))
    dd = partition.distributed_darcy_levels(dar, levels, eown, nparts)
    rng = np.random.default_rng(1)
    for k, pieces in dd["blocks"].items():
        G = dd["global_blocks"][k]
        W_ = orc.InProcessWorld(pieces)
        x = rng.standard_normal(G.shape[1]); y = rng.standard_normal(G.shape[0])
        assert np.allclose(W_.matvec(x), G @ x, rtol=1e-12, atol=1e-13)
        assert np.allclose(W_.matvec(y, trans=True), G.T @ y, rtol=1e-12, atol=1e-13)
    for l in range(len(Ps)):
        (f2, f3), (c2, c3) = dd["order"][l], dd["order"][l + 1]
        for pieces, P, fo, co in ((dd["P"][l][0], Ps[l][0], f2, c2), (dd["P"][l][1], Ps[l][1], f3, c3)):
            Pg = sp.csr_matrix(P)[fo][:, co]
            xc = rng.standard_normal(Pg.shape[1])
            assert np.allclose(orc.InProcessWorld(pieces).matvec(xc), Pg @ xc, rtol=1e-12, atol=1e-13)
    # ownership: an element's dofs follow the element, a facet's dofs the lowest rank among its elements
    for L, eo in zip(levels, eown):
        own2, own3 = partition.darcy_dof_owners(L, eo)
        assert np.array_equal(own3, np.repeat(eo, np.diff(L.idof[3][0])))
        ef_ptr, ef_idx = L.B[0][0], L.B[0][1]
        e_of = np.repeat(np.arange(L.n[0]), np.diff(ef_ptr))
        first_dof_of_facet = L.idof[2][1][:-1]
        has = np.diff(L.idof[2][1]) > 0
        fo = own2[first_dof_of_facet[has]]
        lowest = np.full(L.n[1], 10 ** 6); np.minimum.at(lowest, ef_idx, eo[e_of])
        assert np.array_equal(fo, lowest[has])
    with pytest.raises(ValueError, match="coarsest boxes"):
        partition.hex_element_owners((4, 4, 4), [box_partition((4, 4, 4), (2, 2, 2)), box_partition((2, 2, 2), (2, 2, 2))], 2)
