"""``DeRhamSequence::Coarsen`` on several ranks (``parelag_amd/derham_dist.py``): replicated integer topology, numeric stages
sharded by agglomerate ownership with the interface entities computed redundantly, one integer exchange for the coarse dof
numbering.  The reference's counterpart is its shared-entity handling (``src/amge/DeRhamSequence.cpp:1820, 1930, 1971, 2024``);
the property tested is the one its parallel ctest runs rely on (``par_form1_np2`` expects the serial numbers,
``testsuite/CMakeLists.txt:127-133``): the ranks' pieces put together are the single-rank result.

CPU half: the oracle's numeric path under the product's restriction / renumbering (in-process ranks), and the count
exchange over a real two-process gloo group.  GPU half: the device path per rank."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import derham as od
from parelag_amd import derham, derham_dist as dd
from parelag_amd.amge_level import box_partition
from parelag_amd.derham_level import seq_level_from_hex

from tests.test_derham_oracle import kuhn_boxes


def in_process(lv, part, nranks, owner, coarsen_fn):
    """All ranks one after the other in this process; the all-reduce is replaced by a two-pass stand-in: every rank first
    reports its counts (``probe``), then runs again against the elementwise maximum."""
    topo = derham.coarsen_topology(lv, part)
    reported = []

    def probe(local):
        reported.append(local.copy())
        return local
    for r in range(nranks):
        dd.coarsen_rank(lv, part, r, nranks, owner=owner, topo=topo, exchange=probe, coarsen_fn=coarsen_fn)
    glob = np.max(np.stack(reported), axis=0)
    for a in reported:
        h = a.size // 2                                            # [counts | -counts where held] (one MAX carries max and min)
        assert np.all((a[:h] == 0) | (a[:h] == glob[:h]))           # ranks sharing an entity agree on its coarse dofs
    pieces = [dd.coarsen_rank(lv, part, r, nranks, owner=owner, topo=topo, exchange=lambda local: glob, coarsen_fn=coarsen_fn)
              for r in range(nranks)]
    return topo, pieces


def same_as_single_rank(lv, merged, single, tol):
    c, cs = merged["coarse"], single["coarse"]
    for j in range(lv.jstart, 4):
        assert [np.array_equal(a, b) for a, b in zip(c.idof[j], cs.idof[j])] == [True] * (lv.cb(j) + 1)
        d = abs(merged["P"][j] - single["P"][j])
        assert (d.max() if d.nnz else 0.0) <= tol * abs(single["P"][j]).max(), j
        assert np.abs(c.T[j] - cs.T[j]).max() <= tol * max(np.abs(cs.T[j]).max(), 1e-300)
        for k in range(lv.cb(j) + 1):
            assert np.array_equal(c.M[j][k][0], cs.M[j][k][0])
            assert np.abs(c.M[j][k][1] - cs.M[j][k][1]).max() <= tol * np.abs(cs.M[j][k][1]).max()
        if j < 3:
            d = abs(c.D[j] - cs.D[j])
            assert (d.max() if d.nnz else 0.0) <= tol * abs(cs.D[j]).max()
        assert np.array_equal(c.pv[j], cs.pv[j])
    assert np.abs(c.const3 - cs.const3).max() <= tol * np.abs(cs.const3).max()


CASES = {"hex_blocks_2": ("hex", 2, None), "hex_blocks_3": ("hex", 3, None), "hex_checkerboard": ("hex", 2, "checker"),
         "tets_hetero_3": ("tet", 3, None), "coarse_level_2": ("coarse", 2, None)}


def make(case):
    kind, nranks, own = CASES[case]
    if kind == "hex":
        n = (4, 4, 4)
        lv, part = seq_level_from_hex(n, h=(0.25, 0.5, 1.0), jstart=0), box_partition(n, (2, 2, 2)).astype(np.int64)
    elif kind == "coarse":
        # an algebraic level (null-space dofs, RangeT bubbles, several dofs per coarse entity) coarsened again: 27 agglomerates of
        # 48 heterogeneous tets -> 2 x 1 x 1 slabs ... here 3 slabs of 9 agglomerates
        lv0, part0 = kuhn_boxes(n=6)
        lv = od.coarsen(lv0, part0)["coarse"]
        part = (np.arange(27) // 9).astype(np.int64)
    else:
        lv, part = kuhn_boxes()
    owner = None if own is None else (np.arange(int(part.max()) + 1) * 7 // 3) % nranks
    return lv, part, nranks, owner


@pytest.mark.parametrize("case", sorted(CASES))
def test_rank_pieces_reproduce_the_single_rank_coarsening_oracle_numerics(case):
    lv, part, nranks, owner = make(case)
    topo, pieces = in_process(lv, part, nranks, owner, od.coarsen)
    assert sum(p.agglomerates.size for p in pieces) == topo["n"][0]
    shared = sum(p.csel[1].size for p in pieces) - topo["n"][1]
    assert shared > 0                                                   # some agglomerated facets are held by two ranks
    merged = dd.merge_pieces(lv, topo, pieces, atol=1e-14)
    same_as_single_rank(lv, merged, od.coarsen(lv, part), tol=1e-14)


def test_a_rank_without_agglomerates_takes_part_and_contributes_nothing():
    lv, part, _n, _o = make("hex_blocks_2")
    owner = np.zeros(8, dtype=np.int64)
    owner[4:] = 2                                                    # rank 1 of 3 owns nothing
    topo, pieces = in_process(lv, part, 3, owner, od.coarsen)
    assert pieces[1].agglomerates.size == 0 and pieces[1].P[2].shape == (0, pieces[0].P[2].shape[1])
    same_as_single_rank(lv, dd.merge_pieces(lv, topo, pieces, atol=1e-14), od.coarsen(lv, part), tol=1e-14)


def _two_level_chain(coarsen_fn, tol):
    """Two coarsenings where the ranks never see the numeric data of the intermediate level as a whole: the second call gets
    the piece of the first (``held``) and only the integer topology of the coarse level is global."""
    n = (8, 8, 8)
    lv = seq_level_from_hex(n, h=(0.125, 0.25, 0.5), jstart=0)
    part0, part1 = box_partition(n, (2, 2, 2)).astype(np.int64), box_partition((4, 4, 4), (2, 2, 2)).astype(np.int64)
    owner1 = np.array([0, 0, 0, 1, 1, 2, 2, 2])                     # three ranks; nested: a level-0 agglomerate follows its parent
    owner0 = owner1[part1]
    topo0, pieces0 = in_process(lv, part0, 3, owner0, coarsen_fn)
    lv1_int = dd.integer_level(topo0, lv.jstart)
    topo1 = derham.coarsen_topology(lv1_int, part1)
    reported = []
    for r in range(3):
        dd.coarsen_rank(lv1_int, part1, r, 3, owner=owner1, topo=topo1, exchange=lambda a: (reported.append(a.copy()), a)[1],
                        coarsen_fn=coarsen_fn, held=pieces0[r])
    glob = np.max(np.stack(reported), axis=0)
    pieces1 = [dd.coarsen_rank(lv1_int, part1, r, 3, owner=owner1, topo=topo1, exchange=lambda a: glob, coarsen_fn=coarsen_fn, held=pieces0[r])
               for r in range(3)]
    single0 = coarsen_fn(lv, part0)
    single1 = coarsen_fn(single0["coarse"], part1)
    same_as_single_rank(single0["coarse"], dd.merge_pieces(single0["coarse"], topo1, pieces1, atol=max(tol, 1e-300)), single1, tol)
    # ownership that is not nested is refused
    with pytest.raises(ValueError, match="not nested"):
        dd.coarsen_rank(lv1_int, part1, 0, 3, owner=np.array([0, 1, 0, 1, 1, 2, 2, 2]), topo=topo1, exchange=lambda a: glob, coarsen_fn=coarsen_fn,
                        held=pieces0[0])


def test_two_level_chain_without_a_global_coarse_level_oracle_numerics():
    _two_level_chain(od.coarsen, tol=1e-13)


@pytest.mark.gpu
def test_two_level_chain_without_a_global_coarse_level_on_device():
    _two_level_chain(derham.coarsen, tol=0.0)


def test_restriction_keeps_the_sequence_exact():
    lv, part, _n, _o = make("tets_hetero_3")
    topo = derham.coarsen_topology(lv, part)
    R = dd.restrict_level(lv, topo, np.array([1, 6]))
    s = R.level
    assert s.n[0] == 96 and all(abs(s.D[j + 1] @ s.D[j]).max() == 0.0 for j in range(2))
    for c in range(2):
        Bc = [sp.csr_matrix((s.B[k][2], s.B[k][1], s.B[k][0]), shape=(s.n[k], s.n[k + 1])) for k in (c, c + 1)]
        assert abs(Bc[0] @ Bc[1]).max() == 0.0
    for j in range(4):
        full = lv.mass(j, 0, scaling=np.isin(np.arange(lv.n[0]), R.sel[0]) * 1.0)       # mass of the kept elements only
        assert abs(s.mass(j) - full[R.dofs[j]][:, R.dofs[j]]).max() < 1e-15
    with pytest.raises(ValueError):
        dd._restrict_rows(*lv.B[0][:2], R.sel[0], np.full(lv.n[1], -1))


def _gloo_worker(rank, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        lv, part, _n, _o = make("hex_blocks_2")
        piece = dd.coarsen_rank(lv, part, rank, 2, coarsen_fn=od.coarsen)              # the real all-reduce of the counts
        full = dd.coarsen_allgather(lv, part, rank, 2, coarsen_fn=od.coarsen)          # + the all-gather of the pieces
        single = od.coarsen(lv, part)
        same_as_single_rank(lv, full, single, tol=1e-14)
        q.put((rank, {k: v.tolist() for k, v in piece.counts.items()}, [p.shape for p in piece.P], piece.coarse_dofs[2].tolist()))
    finally:
        dist.destroy_process_group()


def test_count_exchange_over_gloo_world_size_2():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lv, part, _n, _o = make("hex_blocks_2")
    single = od.coarsen(lv, part)
    assert got[0][1] == got[1][1]                                                       # both ranks hold the global counts
    for (j, c), cnt in got[0][1].items():
        assert cnt == np.diff(single["coarse"].idof[j][c]).tolist()
    assert got[0][2] == got[1][2] and [s[1] for s in got[0][2]] == [single["coarse"].ndof(j) for j in range(4)]
    assert set(got[0][3]) | set(got[1][3]) == set(range(single["coarse"].ndof(2)))


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_rank_pieces_reproduce_the_single_rank_coarsening_on_device(case):
    lv, part, nranks, owner = make(case)
    topo, pieces = in_process(lv, part, nranks, owner, derham.coarsen)
    # bit for bit: workspace placement and workgroup size are properties of the entity, not of the launch it is part of
    # (csrc/amge.hip, Placement), so an interface entity computed by two ranks, or in a smaller batch, comes out identical
    merged = dd.merge_pieces(lv, topo, pieces, atol=1e-300)
    single = derham.coarsen(lv, part)
    same_as_single_rank(lv, merged, single, tol=0.0)
    # the merged result is a valid coarsening on its own: the reference's invariants
    from tests.test_derham_oracle import invariants
    merged["Pi"] = {j: derham.projector(single, j) for j in range(4)}
    inv = invariants(lv, merged, Pi=merged["Pi"])
    assert max(inv.values()) < 1e-9, inv


def _two_rank_form1_errors(coarsen_fn):
    """``par_form1_np2`` (``testsuite/CMakeLists.txt:127-133``): ``UpscalingGeneralForm --form 1 --nref_parallel 1`` on two ranks must
    print the serial numbers."""
    from tests import test_upscaling_golden as ug
    lv, part = ug.fine_level()
    topo, pieces = in_process(lv, part, 2, None, coarsen_fn)
    merged = dd.merge_pieces(lv, topo, pieces)
    fine, coarse = derham.DeRhamSequence(lv), derham.DeRhamSequence(merged["coarse"])
    l2, energy = ug.upscaling_errors(1, fine, coarse, merged["P"][1].tocsr())
    return ("%.4e" % l2, "%.4e" % energy), ug.GOLDEN[1]


def test_two_ranks_print_the_serial_upscaling_errors_oracle_numerics():
    got, want = _two_rank_form1_errors(od.coarsen)
    assert got == want


@pytest.mark.gpu
def test_two_ranks_print_the_serial_upscaling_errors_on_device():
    got, want = _two_rank_form1_errors(derham.coarsen)
    assert got == want


def test_disagreement_on_a_shared_entity_raises_on_every_rank():
    """the count exchange carries max AND min over the holders of an entity, so the rank with the larger count notices too and
    no rank walks on into the next collective alone"""
    n = (4, 4, 2)
    lv = seq_level_from_hex(n, jstart=1)
    part = box_partition(n, (2, 2, 2)).astype(np.int64)
    owner = np.array([0, 1, 0, 1])
    topo = derham.coarsen_topology(lv, part)
    reported = []
    for r in range(2):
        dd.coarsen_rank(lv, part, r, 2, owner=owner, topo=topo, exchange=lambda a: (reported.append(a.copy()), a)[1], coarsen_fn=od.coarsen)
    h = reported[0].size // 2
    shared = np.flatnonzero((reported[0][:h] > 0) & (reported[1][:h] > 0))
    assert shared.size > 0
    tampered = [a.copy() for a in reported]
    tampered[1][shared[0]] += 1                                    # rank 1 believes the shared entity has one more coarse dof
    tampered[1][h + shared[0]] -= 1
    glob = np.max(np.stack(tampered), axis=0)
    for r in range(2):
        with pytest.raises(RuntimeError, match="ranks disagree"):
            dd.coarsen_rank(lv, part, r, 2, owner=owner, topo=topo, exchange=lambda a: glob, coarsen_fn=od.coarsen)
