"""CPU suite for the oracle of the whole ``DeRhamSequence::Coarsen`` (``oracle/derham.py``, forms jformStart..3): the
reference cannot run here, so the restatement is pinned by the reference's invariants (``DeRhamSequence::CheckInvariants``,
``src/amge/DeRhamSequence.cpp:694-970``): ``D_f P_j = P_{j+1} D_c`` (CheckDP), ``D_c D_c = 0`` (CheckD), ``M_c = P^T M_f P``
(CheckCoarseMassMatrix), ``Pi P = I`` (CheckPi) and the targets lying in the coarse spaces; the product's integer tables
(``parelag_amd/derham.py``) are checked bit-exact against the oracle's entity-by-entity loops."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import amge as oa, derham as od
from parelag_amd import derham, meshgen
from parelag_amd.amge_level import level_from_tets
from parelag_amd.amge_level import box_partition
from parelag_amd.derham_level import seq_level_from_hex, seq_level_from_tets


def invariants(lv, res, Pi=None):
    c, P = res["coarse"], res["P"]
    Pi = res["Pi"] if Pi is None else Pi
    out = {}
    for j in range(lv.jstart, 4):
        M, Mc = lv.mass(j), c.mass(j)
        out[f"mass{j}"] = abs(P[j].T @ M @ P[j] - Mc).max() / abs(Mc).max()
        out[f"pip{j}"] = abs(Pi[j] @ P[j] - sp.identity(P[j].shape[1])).max()
        out[f"targets{j}"] = np.abs(P[j] @ c.T[j] - lv.T[j]).max() / np.abs(lv.T[j]).max()
        if j < 3:
            out[f"commute{j}"] = abs(lv.D[j] @ P[j] - P[j + 1] @ c.D[j]).max() / max(abs(lv.D[j] @ P[j]).max(), 1e-300)
        if j < 2:
            out[f"exact{j}"] = abs(c.D[j + 1] @ c.D[j]).max()
    return out


@pytest.fixture(scope="module")
def seq():
    return meshgen.Sequence.kuhn_cube(1, 2)


def test_finest_level_is_a_complex_and_matches_the_generator(seq):
    m = seq.meshes[0]
    lv = seq_level_from_tets(m, jstart=0)
    assert abs(lv.D[1] @ lv.D[0]).max() == 0.0 and abs(lv.D[2] @ lv.D[1]).max() == 0.0
    S2 = sp.diags(lv.meta["S2"])
    assert abs(m.M1() - lv.mass(1)).max() < 1e-15                          # edges keep the generator's orientation
    assert abs(S2 @ m.M2() @ S2 - lv.mass(2)).max() < 1e-15
    assert abs(S2 @ m.D1() - lv.D[1]).max() == 0.0 and abs(m.D0() - lv.D[0]).max() == 0.0
    assert abs(m.M0_P1_stiffness() - lv.D[0].T @ lv.mass(1) @ lv.D[0]).max() < 1e-12
    # gradients of the linears are the constant fields, curls of constants vanish
    assert np.abs(lv.D[0] @ lv.T[0][:, 1:] - lv.T[1]).max() < 1e-15 and np.abs(lv.D[1] @ lv.T[1]).max() < 1e-15
    # lower-dimensional masses: tangential trace on a face (partition of unity of the Whitney forms' traces), edge, vertex
    one_t = lv.T[1]                                                         # constant fields
    ed = lv.entity_dof(1, 1)
    f = 5
    t = one_t[ed[f]]
    assert np.allclose(t.T @ lv.block(1, 1, f) @ t, (np.eye(3) - np.outer(lv.T[2][f], lv.T[2][f]) / lv.pv[2][f] ** 2) * lv.pv[2][f], atol=1e-13)


@pytest.mark.parametrize("jstart", [2, 1, 0])
def test_reference_invariants_two_levels(seq, jstart):
    lv = seq_level_from_tets(seq.meshes[0], jstart=jstart)
    r0 = od.coarsen(lv, seq.partition(0))
    r1 = od.coarsen(r0["coarse"], seq.partition(1))
    for l, r in ((lv, r0), (r0["coarse"], r1)):
        inv = invariants(l, r)
        assert max(inv.values()) < 1e-12, inv
    for m in seq.meshes:
        m.build_faces()
    # derefined tets with lowest-order targets: one coarse dof per coarse vertex / edge / face / element
    want = lambda m: [m.nv, m.ne, m.nf, m.nt][jstart:]
    assert [r0["coarse"].ndof(j) for j in range(jstart, 4)] == want(seq.meshes[1])
    assert [r1["coarse"].ndof(j) for j in range(jstart, 4)] == want(seq.meshes[2])


def test_general_oracle_equals_the_two_form_oracle(seq):
    lv = seq_level_from_tets(seq.meshes[0], jstart=2)
    r = od.coarsen(lv, seq.partition(0))
    r2 = oa.coarsen(level_from_tets(seq.meshes[0]), seq.partition(0))
    assert abs(r["P"][2] - r2["P2"]).max() == 0.0 and abs(r["P"][3] - r2["P3"]).max() == 0.0
    assert abs(r["coarse"].D[2] - r2["coarse"].D2).max() == 0.0 and abs(r["Pi"][2] - r2["Pi2"]).max() < 1e-15


def test_heterogeneous_coefficient_gives_null_space_dofs(seq):
    rng = np.random.default_rng(3)
    lv = seq_level_from_tets(seq.meshes[0], jstart=1, kinv=np.exp(2 * rng.standard_normal(seq.meshes[0].nt)))
    r0 = od.coarsen(lv, seq.partition(0))
    assert r0["n_null"][2] > 0
    inv = invariants(lv, r0)
    assert max(inv.values()) < 1e-10, inv
    r1 = od.coarsen(r0["coarse"], seq.partition(1))
    inv = invariants(r0["coarse"], r1)
    assert max(inv.values()) < 1e-10, inv


@pytest.mark.parametrize("jstart", [1, 0])
def test_product_topology_and_dof_tables_are_bit_exact(seq, jstart):
    lv = seq_level_from_tets(seq.meshes[0], jstart=jstart)
    a, b = od.coarsen_topology(lv, seq.partition(0)), derham.coarsen_topology(lv, seq.partition(0))
    assert a["n"] == b["n"]
    for c in range(4):
        for k in range(3):
            assert np.array_equal(a["A"][c][k], b["A"][c][k]), ("A", c, k)
            if c < 3:
                assert np.array_equal(a["B"][c][k], b["B"][c][k]), ("B", c, k)
    tb = derham._Tables(lv)
    for j in range(jstart, 4):
        ao, ap = od.agglomerate_dofs(lv, a, j), derham.agglomerate_dofs(lv, tb, b, j)
        for c in range(lv.cb(j) + 1):
            assert np.array_equal(np.concatenate(ao["rows"][c]), ap["rows"][c][1]), (j, c)
            assert np.array_equal(np.array(ao["nint"][c]), ap["nint"][c]), (j, c)
        ep, ei = tb.edof(j, 0)
        assert np.array_equal(np.concatenate(lv.entity_dof(j, 0)), ei)


def kuhn_boxes(n=4, hetero=True, jstart=0):
    """n^3 Kuhn cells (6 tets each), agglomerated in 2 x 2 x 2 boxes of cells: 48 tets per agglomerate, not a derefinement --
    agglomerated facets of 8 triangles, ridges of 2 edges, balls / discs / paths as the lower forms need them."""
    seq = meshgen.Sequence.kuhn_cube(n, 0)
    m = seq.meshes[0]
    cell = np.minimum((m.vertices[m.tets].mean(1) * n).astype(int), n - 1)
    part = ((cell[:, 0] // 2) * (n // 2) + cell[:, 1] // 2) * (n // 2) + cell[:, 2] // 2
    kinv = np.exp(np.random.default_rng(5).standard_normal(m.nt)) if hetero else None
    return seq_level_from_tets(m, jstart=jstart, kinv=kinv), part.astype(np.int64)


def test_box_agglomerates_with_heterogeneous_coefficient():
    lv, part = kuhn_boxes()
    r0 = od.coarsen(lv, part)
    assert r0["coarse"].n == [8, 36, 54, 27]
    assert r0["n_null"][2] == 24 and r0["coarse"].ndof(1) == 54 + 24          # null dofs of form 2 <-> RangeT bubbles of form 1
    inv = invariants(lv, r0)
    assert max(inv.values()) < 1e-11, inv
    r1 = od.coarsen(r0["coarse"], np.zeros(8, dtype=np.int64))                # one agglomerate: the whole cube
    assert r1["coarse"].n == [1, 6, 12, 8]
    inv = invariants(r0["coarse"], r1)
    assert max(inv.values()) < 1e-11, inv


def rich_targets_level(seq, jstart=0):
    """"Upscaling order" 1: linears for the 3-forms, linear vector fields for the 2- and 1-forms (midpoint rule), quadratics
    for the 0-forms -- null-space dofs on every codimension, RangeT bubbles in every extension stage."""
    m = seq.meshes[0]
    lv = seq_level_from_tets(m, jstart=jstart)
    xv = m.vertices
    xe = 0.5 * (xv[m.edges[:, 0]] + xv[m.edges[:, 1]]); xf = xv[m.faces].mean(1); xt = xv[m.tets].mean(1)
    lin = lambda T, xc: np.concatenate([T] + [T[:, [d]] * xc[:, [c]] for d in range(3) for c in range(3)], axis=1)
    lv.T[3] = np.concatenate([lv.T[3], lv.T[3] * xt], axis=1)
    lv.T[2] = lin(lv.T[2], xf)
    if jstart <= 1:
        lv.T[1] = lin(lv.T[1], xe)
    if jstart <= 0:
        x, y, z = xv.T
        lv.T[0] = np.stack([np.ones_like(x), x, y, z, x * x, y * y, z * z, x * y, y * z, x * z], axis=1)
    return lv


def test_richer_targets_on_every_form(seq):
    lv = rich_targets_level(seq)
    r0 = od.coarsen(lv, seq.partition(0))
    assert r0["n_null"][3] == 48 * 3 and r0["n_null"][2] > 0 and r0["n_null"][1] > 0
    inv = invariants(lv, r0)
    assert max(inv.values()) < 1e-11, inv
    r1 = od.coarsen(r0["coarse"], seq.partition(1))
    inv = invariants(r0["coarse"], r1)
    assert max(inv.values()) < 1e-11, inv


def test_hex_sequence_is_exact_and_coarsens(seq):
    """The Cartesian hex builder for all forms (the element type of MultigridTestSPE10): exact complex, constants' energy =
    volume in every mass matrix, gradients of linears = constants; two coarsenings with 2 x 2 x 2 (then 2 x 2 x 1) boxes keep the
    reference's invariants, with and without a heterogeneous coefficient."""
    n, h = (4, 4, 2), (2.0, 1.0, 0.5)
    vol = float(np.prod(n) * np.prod(h))
    for kinv in (None, np.exp(np.random.default_rng(0).standard_normal(32))):
        lv = seq_level_from_hex(n, h, jstart=0, kinv=kinv)
        assert abs(lv.D[1] @ lv.D[0]).max() == 0.0 and abs(lv.D[2] @ lv.D[1]).max() == 0.0
        assert np.abs(lv.D[0] @ lv.T[0][:, 1:] - lv.T[1]).max() == 0.0
        assert np.abs(lv.D[1] @ lv.T[1]).max() == 0.0 and np.abs(lv.D[2] @ lv.T[2]).max() == 0.0
        assert np.allclose((lv.T[1].T @ lv.mass(1) @ lv.T[1]), vol * np.eye(3), rtol=1e-13, atol=1e-13)
        if kinv is None:
            assert np.allclose((lv.T[2].T @ lv.mass(2) @ lv.T[2]), vol * np.eye(3), rtol=1e-13, atol=1e-13)
        one = np.ones(lv.ndof(0))
        assert np.isclose(one @ lv.mass(0) @ one, vol, rtol=1e-13)
        r0 = od.coarsen(lv, box_partition(n, (2, 2, 2)))
        assert r0["coarse"].n == [4, 20, 33, 18]
        inv = invariants(lv, r0)
        assert max(inv.values()) < 1e-11, inv
        r1 = od.coarsen(r0["coarse"], box_partition((2, 2, 1), (2, 2, 1)))
        assert r1["coarse"].n == [1, 6, 12, 8]
        inv = invariants(r0["coarse"], r1)
        assert max(inv.values()) < 1e-11, inv
        if kinv is not None:
            assert r0["n_null"][2] > 0


def test_element_scaled_mass_operator():
    """``ComputeMassOperator(jform, elemMatrixScaling)`` (``DeRhamSequence.cpp:1325-1370``): every element block scaled by its
    factor before the assembly; a vector of the wrong length is refused like in the reference."""
    from parelag_amd import derham
    from parelag_amd.derham_level import seq_level_from_hex
    seq = derham.DeRhamSequence(seq_level_from_hex((2, 3, 2), jstart=0))
    ne = seq.level.n[0]
    s = np.linspace(0.5, 3.0, ne)
    for j in range(4):
        M = seq.ComputeMassOperator(j, s)
        ref = sum(seq.ComputeMassOperator(j, (np.arange(ne) == e) * s[e]) for e in range(ne))
        assert abs(M - ref).max() < 1e-15
        assert abs(seq.ComputeMassOperator(j, np.ones(ne)) - seq.ComputeMassOperator(j)).max() == 0.0
    with pytest.raises(ValueError, match="elemMatrixScaling has the wrong size"):
        seq.ComputeMassOperator(2, np.ones(ne + 1))
    assert seq.GetNumTrueDofs(1) == seq.GetNumberOfDofs(1) and seq.GetL2ConstRepresentation().shape == (ne,)
