"""The sharded generator (parelag_amd/meshgen_sharded.py) against the serial one: every rank's shard is the row block
partition.split_rows cuts out of meshgen.h1_problem's matrices -- same numbering, same pattern, same comm package."""
import threading
import traceback

import numpy as np
import pytest
import scipy.sparse as sp

from parelag_amd import meshgen, meshgen_sharded as ms, partition


def run_threads(world, fn):
    group = ms.ThreadComm.Group(world)
    out, errs = [None] * world, []

    def target(r):
        try:
            out[r] = fn(ms.ThreadComm(group, r))
        except Exception:
            errs.append(traceback.format_exc())
            group.barrier.abort()
    ts = [threading.Thread(target=target, args=(r,)) for r in range(world)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=600)
    assert not errs, "\n".join(errs)
    return out


def same_csr(a, b, exact=True, tol=0.0):
    a, b = partition.canon(a), partition.canon(b)
    assert a.shape == b.shape
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)
    if exact:
        assert np.array_equal(a.data, b.data)
    else:
        assert np.max(np.abs(a.data - b.data), initial=0.0) <= tol * max(np.max(np.abs(b.data), initial=0.0), 1e-300)


@pytest.fixture(scope="module")
def serial():
    return meshgen.h1_problem(2)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_shards_are_the_row_blocks_of_the_serial_problem(serial, world):
    A, Ps, seq = serial
    d = partition.distributed_h1_hierarchy(A, Ps, seq, world, galerkin=False)
    shards = run_threads(world, lambda comm: ms.sharded_h1_problem(comm, 2))
    for r, sh in enumerate(shards):
        for l in range(len(Ps) + 1):
            assert np.array_equal(sh["starts"][l], d["levels"][l]["starts"])
        s = d["levels"][0]["starts"]
        same_csr(sh["A_rows"], d["A_global"][0][int(s[r]):int(s[r + 1])], exact=False, tol=1e-13)
        # explicit zeros of the eliminated rows / columns are part of the pattern
        assert sh["A_rows"].nnz == d["A_global"][0][int(s[r]):int(s[r + 1])].nnz
        for l, P in enumerate(d["P_global"]):
            sf = d["levels"][l]["starts"]
            same_csr(sh["P_rows"][l], P[int(sf[r]):int(sf[r + 1])])


@pytest.mark.parametrize("world", [2, 4])
def test_parcsr_pieces_carry_the_comm_package_of_split_rows(serial, world):
    A, Ps, seq = serial
    d = partition.distributed_h1_hierarchy(A, Ps, seq, world, galerkin=False)

    def body(comm):
        sh = ms.sharded_h1_problem(comm, 2)
        pa = ms.parcsr_piece(comm, sh["A_rows"], sh["starts"][0], sh["starts"][0])
        pp = ms.parcsr_piece(comm, sh["P_rows"][0], sh["starts"][0], sh["starts"][1])
        return pa, pp
    res = run_threads(world, body)
    for r, (pa, pp) in enumerate(res):
        for got, ref, exact in ((pa, d["levels"][0]["A_parts"][r], False), (pp, d["levels"][1]["P_parts"][r], True)):
            same_csr(got["diag"], ref["diag"], exact=exact, tol=1e-13)
            same_csr(got["offd"], ref["offd"], exact=exact, tol=1e-13)
            for k in ("col_map_offd", "recv_procs", "recv_starts", "send_procs", "send_starts", "send_elmts"):
                assert np.array_equal(got[k], ref[k]), k
            assert got["first_row"] == ref["first_row"] and got["first_col"] == ref["first_col"]
            assert tuple(got["global_shape"]) == tuple(ref["global_shape"])


@pytest.mark.parametrize("world", [1, 3, 8])
def test_tree_order_is_a_permutation_inside_every_rank_s_block(world):
    """order="tree" numbers a rank's dofs by the refinement tree instead of the serial generator's order: same partition, same owners, and
    the problem is the natural one under a symmetric permutation that never leaves a rank's block -- operator entries to rounding (sums over
    ranks), interpolators bit for bit."""
    nat = run_threads(world, lambda comm: ms.sharded_h1_problem(comm, 2, order="natural"))
    tre = run_threads(world, lambda comm: ms.sharded_h1_problem(comm, 2, order="tree"))
    nl = len(nat[0]["starts"])
    perms = []
    for l in range(nl):
        st = nat[0]["starts"][l]
        assert np.array_equal(st, tre[0]["starts"][l])
        perm = np.full(int(st[-1]), -1, dtype=np.int64)                    # tree number -> natural number
        for r in range(world):
            ow = nat[r]["owned"][l]
            assert np.array_equal(ow, tre[r]["owned"][l])
            gt, gn = tre[r]["gid"][l][ow], nat[r]["gid"][l][ow]
            assert gt.min() >= st[r] and gt.max() < st[r + 1]               # never leaves the rank's block
            perm[gt] = gn
        assert np.array_equal(np.sort(perm), np.arange(len(perm)))
        for r in range(world):                                                # ghosts carry their owners' numbers
            ow = nat[r]["owned"][l]
            assert np.array_equal(perm[tre[r]["gid"][l][~ow]], nat[r]["gid"][l][~ow])
        perms.append(perm)
    if world > 1:
        assert any(not np.array_equal(p, np.arange(len(p))) for p in perms)
    An = sp.vstack([s_["A_rows"] for s_ in nat]).tocsr()
    At = sp.vstack([s_["A_rows"] for s_ in tre]).tocsr()
    same_csr(At, An[perms[0]][:, perms[0]], exact=False, tol=1e-13)
    for l in range(nl - 1):
        Pn = sp.vstack([s_["P_rows"][l] for s_ in nat]).tocsr()
        Pt = sp.vstack([s_["P_rows"][l] for s_ in tre]).tocsr()
        same_csr(Pt, Pn[perms[l]][:, perms[l + 1]])
    # and the tree order is the better one for a gather: entries per distinct column of a 2304-entry window
    def reuse(M):
        M = sp.csr_matrix(M); M.sort_indices()
        tot = 0
        for b in range(0, M.nnz, 2304):
            tot += len(np.unique(M.indices[b:b + 2304]))
        return M.nnz / tot
    if world == 1:
        assert reuse(At) > reuse(An)


def test_sharded_generator_on_another_base_mesh_and_with_fewer_levels():
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    base = seq.meshes[-1]
    fine = seq.meshes[0]
    A = meshgen.eliminate_rowcol(fine.M0_P1_stiffness(), fine.boundary_marker_unit_cube())
    Ps = [meshgen.eliminate_cols(meshgen.interp_P0(seq.meshes[l + 1], seq.meshes[l]), seq.meshes[l + 1].boundary_marker_unit_cube()) for l in range(1)]
    seq2 = meshgen.Sequence(seq.meshes[:2])
    # the serial partition deals the base tets out and lets children inherit: with two of three meshes kept the "base" of the
    # partition is still the coarsest mesh
    eparts = partition.element_partitions(seq, 3)
    shards = run_threads(3, lambda comm: ms.sharded_h1_problem(comm, 2, nlevels=2, base=(base.vertices, base.tets)))
    own = partition.vertex_owners(fine, eparts[0])
    order, new_of_old, starts = partition.block_numbering(own, 3)
    Ag = partition.canon(sp.csr_matrix(A)[order][:, order])
    ownc = partition.vertex_owners(seq.meshes[1], eparts[1])
    orderc, _, startsc = partition.block_numbering(ownc, 3)
    Pg = partition.canon(sp.csr_matrix(Ps[0])[order][:, orderc])
    for r, sh in enumerate(shards):
        assert np.array_equal(sh["starts"][0], starts) and np.array_equal(sh["starts"][1], startsc)
        same_csr(sh["A_rows"], Ag[int(starts[r]):int(starts[r + 1])], exact=False, tol=1e-13)
        same_csr(sh["P_rows"][0], Pg[int(starts[r]):int(starts[r + 1])])
        assert len(sh["P_rows"]) == 1
    del seq2


def test_a_failure_on_one_rank_raises_on_every_rank():
    def body(comm):
        def local():
            if comm.rank == 1:
                raise ValueError("boom")
            return 1
        with pytest.raises(RuntimeError, match="rank 1: ValueError: boom"):
            ms.agree(comm, local)
        return True
    assert run_threads(3, body) == [True, True, True]
