"""CPU suite for SURVEY 8f-3: the oracle of ``DeRhamSequence::Coarsen`` (forms 2 and 3) is pinned by the reference's own
invariants (``DeRhamSequence::CheckInvariants``, ``src/amge/DeRhamSequence.cpp:694-970``) -- the reference itself cannot
run here -- and the product's integer tables (agglomerated topology, coarse dof numbering) are checked bit-exact against
the oracle's entity-by-entity restatement."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import amge as oa
from parelag_amd import amge, meshgen
from parelag_amd.amge_level import box_partition, level_from_hex, level_from_tets

TOPO_KEYS = ("ae_ptr", "ae_idx", "af_ptr", "af_idx", "af_sgn", "B_ptr", "B_idx", "B_sgn", "facet_attr")


def invariants(lv, res):
    c, P2, P3, Pi2, Pi3 = res["coarse"], res["P2"], res["P3"], res["Pi2"], res["Pi3"]
    M2, M3, M2c, M3c = lv.mass2(), lv.mass3(), c.mass2(), c.mass3()
    scale = lambda X: max(abs(X).max(), 1e-300)
    return dict(
        commute=abs(lv.D2 @ P2 - P3 @ c.D2).max() / scale(lv.D2 @ P2),                 # CheckDP (:753-779)
        mass2=abs(P2.T @ M2 @ P2 - M2c).max() / scale(M2c),                             # CheckCoarseMassMatrix (:802-829)
        mass3=abs(P3.T @ M3 @ P3 - M3c).max() / scale(M3c),
        pip2=abs(Pi2 @ P2 - sp.identity(c.ndof2)).max(),                                # CheckPi (:900-970)
        pip3=abs(Pi3 @ P3 - sp.identity(c.ndof3)).max(),
        targets2=np.abs(P2 @ c.T2 - lv.T2).max() / np.abs(lv.T2).max(),                 # targets are in the coarse space
        targets3=np.abs(P3 @ c.T3 - lv.T3).max() / np.abs(lv.T3).max(),
        const=np.abs(P3 @ c.const3 - lv.const3).max() / np.abs(lv.const3).max())


@pytest.fixture(scope="module")
def kuhn():
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    lv = level_from_tets(seq.meshes[0])
    r0 = oa.coarsen(lv, seq.partition(0))
    r1 = oa.coarsen(r0["coarse"], seq.partition(1))
    return seq, lv, r0, r1


def test_finest_level_matches_the_generator(kuhn):
    seq, lv, _r0, _r1 = kuhn
    m = seq.meshes[0]
    S2, S3 = sp.diags(lv.meta["S2"]), sp.diags(lv.meta["S3"])
    assert abs(S2 @ m.M2() @ S2 - lv.mass2()).max() < 1e-15
    assert abs(S3 @ m.D2() @ S2 - lv.D2).max() == 0.0
    assert abs(m.M3() - lv.mass3()).max() == 0.0
    # boundary facets point outward; every interior facet is seen once with each sign
    Bsum = np.asarray(sp.csr_matrix((lv.ef_sgn, lv.ef_idx, lv.ef_ptr), shape=(lv.nE, lv.nF)).sum(axis=0)).ravel()
    assert np.all(Bsum[lv.facet_attr > 0] == 1.0) and np.all(Bsum[lv.facet_attr == 0] == 0.0)
    # the divergence of a constant field vanishes, the constant targets have unit normal flux density
    assert np.abs(lv.D2 @ lv.T2).max() < 1e-14


def test_reference_invariants_two_tet_levels(kuhn):
    _seq, lv, r0, r1 = kuhn
    for l, r in ((lv, r0), (r0["coarse"], r1)):
        inv = invariants(l, r)
        assert max(inv.values()) < 1e-12, inv
    # derefined tets: the AMGe spaces have one dof per coarse face / element, like the nested geometric ones
    for m in _seq.meshes:
        m.build_faces()
    assert (r0["coarse"].ndof2, r0["coarse"].ndof3) == (_seq.meshes[1].nf, _seq.meshes[1].nt)
    assert (r1["coarse"].ndof2, r1["coarse"].ndof3) == (_seq.meshes[2].nf, _seq.meshes[2].nt)
    assert r0["n_null"].sum() == 0 and r0["n_rt"].sum() == 0


def test_coarse_derivative_is_the_coarse_incidence_in_value_dofs(kuhn):
    """With one PV dof per entity the coarse D_2 must be  diag(1/|AE|) * incidence * diag(|AF|)  (the coarse 3-form dof is
    the value of a piecewise constant, the coarse 2-form dof multiplies the unit-normal-flux trace)."""
    seq, lv, r0, _r1 = kuhn
    c = r0["coarse"]
    inc = sp.csr_matrix((c.ef_sgn, c.ef_idx, c.ef_ptr), shape=(c.nE, c.nF))
    vol_c = r0["P3"].T @ np.ones(lv.ndof3)                      # P3 column = fine volumes -> sum = |AE|
    t = r0["topo"]
    area_c = np.add.reduceat(lv.pv2[t["af_idx"]], t["af_ptr"][:-1])
    assert abs(sp.diags(1.0 / vol_c) @ inc @ sp.diags(area_c) - c.D2).max() < 1e-12 * abs(c.D2).max()


def test_hex_levels_with_heterogeneous_coefficient_and_null_dofs():
    rng = np.random.default_rng(1)
    hx = level_from_hex((6, 5, 4), h=(20.0, 10.0, 2.0), kinv=np.exp(2 * rng.standard_normal(120)))
    r0 = oa.coarsen(hx, box_partition((6, 5, 4), (3, 2, 2)))
    r1 = oa.coarsen(r0["coarse"], box_partition((2, 3, 2), (2, 3, 2)))
    assert r0["n_null"].sum() > 0                               # the constants are not minimal-energy extensions any more
    for l, r in ((hx, r0), (r0["coarse"], r1)):
        inv = invariants(l, r)
        assert max(inv.values()) < 1e-10, inv


def test_range_t_bubbles_from_richer_l2_targets():
    hx = level_from_hex((4, 4, 4))
    I, J, K = (g.ravel() for g in np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"))
    hx.T3 = np.stack([hx.const3, hx.const3 * I, hx.const3 * J, hx.const3 * K], axis=1)      # "Upscaling order" 1 for form 3
    r = oa.coarsen(hx, box_partition((4, 4, 4), (2, 2, 2)))
    assert r["coarse"].ndof3 == 8 * 4 and r["n_rt"].sum() == 8 * 3
    inv = invariants(hx, r)
    assert max(inv.values()) < 1e-12, inv
    # every RangeT bubble has a unit entry in the coarse derivative (DeRhamSequence.cpp:2441)
    c = r["coarse"]
    sub = c.D2[:, c.nfd2:]
    assert sub.nnz == 24 and np.all(sub.data == 1.0)


@pytest.mark.parametrize("case", ["tet", "tet_no_attr", "hex", "hex_ragged"])
def test_product_topology_tables_are_bit_exact(case):
    if case.startswith("tet"):
        seq = meshgen.Sequence.kuhn_cube(2, 2)
        lv = level_from_tets(seq.meshes[0], bdr_attr=(case == "tet"))
        part = seq.partition(0)
    elif case == "hex":
        lv, part = level_from_hex((6, 5, 4)), box_partition((6, 5, 4), (3, 2, 2))
    else:
        lv, part = level_from_hex((7, 5, 3)), box_partition((7, 5, 3), (2, 2, 2))
    a, b = oa.coarsen_topology(lv, part), amge.coarsen_topology(lv, part)
    assert a["nAE"] == b["nAE"] and a["nAF"] == b["nAF"]
    for k in TOPO_KEYS:
        assert np.array_equal(a[k], b[k]), k
    # an agglomerated facet separates exactly two agglomerates or lies on the boundary with one attribute
    inc = sp.csr_matrix((np.abs(b["B_sgn"]), b["B_idx"], b["B_ptr"]), shape=(b["nAE"], b["nAF"]))
    per_af = np.asarray(inc.sum(axis=0)).ravel()
    assert np.all((per_af == 2) == (b["facet_attr"] == 0)) or case == "tet_no_attr"


def test_second_level_topology_tables_are_bit_exact(kuhn):
    seq, _lv, r0, _r1 = kuhn
    a, b = oa.coarsen_topology(r0["coarse"], seq.partition(1)), amge.coarsen_topology(r0["coarse"], seq.partition(1))
    for k in TOPO_KEYS:
        assert np.array_equal(a[k], b[k]), k


def upscaling_errors(coarsen_sequence):
    """Numerical upscaling (what ParElag is for, ``examples/Upscaling*.cpp``): the mixed Darcy problem with a smooth source is
    solved on every level of the AMGe sequence (Galerkin operators ``P^T A P``) and the prolongated coarse solutions are
    compared with the fine one in the mass norms.  Returns [(u error, p error) per coarse level]."""
    import scipy.sparse.linalg as spl
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    lv = level_from_tets(seq.meshes[0])
    res = coarsen_sequence(lv, [seq.partition(0), seq.partition(1)])
    M, B, Bt = lv.darcy_blocks()
    W = lv.mass3()
    n0, n1 = M.shape[0], B.shape[0]
    A = sp.bmat([[M, Bt], [B, None]]).tocsc()
    xc = seq.meshes[0].vertices[seq.meshes[0].tets].mean(1)
    f = np.prod(np.sin(np.pi * xc), axis=1)
    rhs = np.concatenate([np.zeros(n0), -(W @ (f * lv.const3))])
    xf = spl.spsolve(A, rhs)
    uf, pf = xf[:n0], xf[n0:]
    out = []
    P2, P3 = sp.identity(n0), sp.identity(n1)
    for r in res:
        P2, P3 = P2 @ r["P2"], P3 @ r["P3"]
        P = sp.block_diag([P2, P3]).tocsc()
        xu = P @ spl.spsolve((P.T @ A @ P).tocsc(), P.T @ rhs)
        eu, ep = xu[:n0] - uf, xu[n0:] - pf
        out.append((np.sqrt((eu @ (M @ eu)) / (uf @ (M @ uf))), np.sqrt((ep @ (W @ ep)) / (pf @ (W @ pf)))))
    return out


def test_upscaling_error_decreases_like_h():
    (u1, p1), (u2, p2) = upscaling_errors(oa.coarsen_sequence)
    # lowest-order spaces: O(h) in both variables -- the error roughly doubles with every coarsening
    assert 0.15 < u1 < 0.30 and 0.15 < p1 < 0.30
    assert 1.7 < u2 / u1 < 2.6 and 1.7 < p2 / p1 < 2.6
