"""GPU parity for SURVEY 8f-3: the device's AMGe coarsening of the (H(div), L2) pair (``parelag_amd/amge.py`` over
``elag_hip_coarsen_traces`` / ``elag_hip_coarsen_extension``) against the CPU oracle of ``DeRhamSequence::Coarsen``
(``oracle/amge.py``) on identical inputs.

Tolerances: integer tables bit-exact; fp64 blocks 1e-10 relative to the largest entry of the compared operator (the
device eliminates with partial pivoting, the oracle uses LAPACK; local condition numbers reach 1e4 with the heterogeneous
coefficient).  Columns that come out of an SVD are determined up to sign (in the reference too: dgesvd); they are compared
after aligning the sign of every coarse dof with the oracle's, and the same signs are applied to the coarse operators."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import amge as oa
from parelag_amd import meshgen
from parelag_amd.amge_level import box_partition, level_from_hex, level_from_tets
from tests.test_amge_oracle import TOPO_KEYS, invariants, upscaling_errors

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.fixture(scope="module")
def amge():
    from parelag_amd import amge as m
    return m


def rel(x, ref):
    ref = sp.csr_matrix(ref) if sp.issparse(ref) else np.asarray(ref)
    x = sp.csr_matrix(x) if sp.issparse(x) else np.asarray(x)
    den = max(abs(ref).max() if ref.size or sp.issparse(ref) and ref.nnz else 0.0, 1e-300)
    d = abs(x - ref).max() if (ref.nnz + x.nnz if sp.issparse(ref) else ref.size) else 0.0
    return d / den


def column_signs(P_dev, P_ref):
    """+-1 per coarse dof so that P_dev diag(s) matches P_ref (SVD columns are defined up to sign)."""
    d = np.asarray(P_dev.multiply(P_ref).sum(axis=0)).ravel()
    s = np.where(d < 0, -1.0, 1.0)
    return s


def compare(amge, lv, part, expect_null=None):
    ro = oa.coarsen(lv, part)
    tm = {}
    rd = amge.coarsen(lv, part, timings=tm)
    for k in TOPO_KEYS:
        assert np.array_equal(ro["topo"][k], rd["topo"][k]), k
    co, cd = ro["coarse"], rd["coarse"]
    for k in ("ef_ptr", "ef_idx", "ef_sgn", "e_dof3_ptr", "f_dof2_ptr", "e_idof2_ptr", "M3e_ptr", "M2e_ptr", "M2f_ptr"):
        assert np.array_equal(getattr(co, k), getattr(cd, k)), k
    assert np.array_equal(ro["n_null"], rd["n_null"]) and np.array_equal(ro["n_rt"], rd["n_rt"])
    if expect_null is not None:
        assert (ro["n_null"].sum() > 0) == expect_null
    # Pi = hat1 + hat0 (I - P hat1) composed here from the device's functionals (amge.projector2 does it on the device)
    Pi2_dev = (rd["hat1"] + rd["hat0"] - rd["hat0"] @ (rd["P2"] @ rd["hat1"])).tocsr()
    s2 = column_signs(rd["P2"], ro["P2"]); s3 = column_signs(rd["P3"], ro["P3"])
    S2, S3 = sp.diags(s2), sp.diags(s3)
    errs = dict(
        P2=rel(rd["P2"] @ S2, ro["P2"]), P3=rel(rd["P3"] @ S3, ro["P3"]),
        D2c=rel(S3 @ cd.D2 @ S2, co.D2), M2c=rel(S2 @ cd.mass2() @ S2, co.mass2()), M3c=rel(S3 @ cd.mass3() @ S3, co.mass3()),
        Pi3=rel(S3 @ rd["Pi3"], ro["Pi3"]), Pi2=rel(S2 @ Pi2_dev, ro["Pi2"]),
        T2=rel(S2 @ cd.T2, co.T2), T3=rel(S3 @ cd.T3, co.T3), const3=rel(s3 * cd.const3, co.const3))
    assert max(errs.values()) < TOL, errs
    # facet mass blocks (not part of the assembled element mass)
    F2 = sp.block_diag([cd.block("M2f", f) for f in range(cd.nF)]) if cd.nF else None
    F2o = sp.block_diag([co.block("M2f", f) for f in range(co.nF)])
    Sf = sp.diags(s2[: cd.nfd2])
    assert rel(Sf @ F2 @ Sf, F2o) < TOL
    # the reference's own invariants hold for the device's result as well
    full = dict(rd, Pi2=Pi2_dev)
    inv = invariants(lv, full)
    assert max(inv.values()) < 1e-9, inv
    return ro, rd, tm


def test_tets_two_levels(amge):
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    lv = level_from_tets(seq.meshes[0])
    _ro, rd, tm = compare(amge, lv, seq.partition(0), expect_null=False)
    assert tm["extension_ms"] > 0.0
    Pi2 = (rd["hat1"] + rd["hat0"] - rd["hat0"] @ (rd["P2"] @ rd["hat1"])).tocsr()
    assert rel(amge.projector2(rd), Pi2) < 1e-13            # ComputeProjector with the device SpGEMM / add
    compare(amge, rd["coarse"], seq.partition(1), expect_null=False)


def test_cube456_refined_once(amge):
    seq = meshgen.Sequence.cube456(1)
    lv = level_from_tets(seq.meshes[0])
    _ro, rd, _tm = compare(amge, lv, seq.partition(0), expect_null=False)
    seq.meshes[1].build_faces()
    assert (rd["coarse"].ndof2, rd["coarse"].ndof3) == (seq.meshes[1].nf, seq.meshes[1].nt) == (1015, 456)


def test_tets_without_boundary_attributes(amge):
    seq = meshgen.Sequence.kuhn_cube(1, 2)
    compare(amge, level_from_tets(seq.meshes[0], bdr_attr=False), seq.partition(0))


def test_hex_heterogeneous_null_dofs_two_levels(amge):
    rng = np.random.default_rng(1)
    hx = level_from_hex((6, 5, 4), h=(20.0, 10.0, 2.0), kinv=np.exp(2 * rng.standard_normal(120)))
    _ro, rd, _tm = compare(amge, hx, box_partition((6, 5, 4), (3, 2, 2)), expect_null=True)
    compare(amge, rd["coarse"], box_partition((2, 3, 2), (2, 3, 2)))


def test_hex_ragged_boxes(amge):
    rng = np.random.default_rng(7)
    hx = level_from_hex((7, 5, 3), h=(1.0, 2.0, 0.5), kinv=np.exp(rng.standard_normal(105)))
    compare(amge, hx, box_partition((7, 5, 3), (2, 2, 2)))


def test_range_t_bubbles(amge):
    hx = level_from_hex((4, 4, 4))
    I, J, K = (g.ravel() for g in np.meshgrid(np.arange(4), np.arange(4), np.arange(4), indexing="ij"))
    # distinct singular values (1 : 2 : 3) so that the SVD basis is unique up to sign
    hx.T3 = np.stack([hx.const3, hx.const3 * I, hx.const3 * 2.0 * J, hx.const3 * 3.0 * K], axis=1)
    ro, _rd, _tm = compare(amge, hx, box_partition((4, 4, 4), (2, 2, 2)))
    assert ro["n_rt"].sum() == 24


def test_large_agglomerates_use_global_scratch(amge):
    """4 x 4 x 4 boxes: 240 local 2-form dofs, a saddle point of order 209 -- beyond the LDS budget of a workgroup."""
    rng = np.random.default_rng(3)
    hx = level_from_hex((8, 8, 4), kinv=np.exp(rng.standard_normal(256)))
    compare(amge, hx, box_partition((8, 8, 4), (4, 4, 4)))


def test_no_targets(amge):
    """Without targets the PV vectors alone span the coarse spaces (``ComputeCoarseTracesNoTargets``, ``DeRhamSequence.cpp:1602-1682``)."""
    seq = meshgen.Sequence.kuhn_cube(1, 2)
    lv = level_from_tets(seq.meshes[0])
    lv.T2 = np.zeros((lv.ndof2, 0)); lv.T3 = np.zeros((lv.ndof3, 0))
    ro = oa.coarsen(lv, seq.partition(0))
    rd = amge.coarsen(lv, seq.partition(0))
    assert rel(rd["P2"], ro["P2"]) < TOL and rel(rd["P3"], ro["P3"]) < TOL and rel(rd["coarse"].D2, ro["coarse"].D2) < TOL


def test_disconnected_agglomerate_is_reported(amge):
    """Two elements that do not touch share no interior facet: the local saddle point is singular
    (``ParELAG_SaddlePointSolver.cpp:117-126`` asserts there)."""
    hx = level_from_hex((4, 1, 1))
    part = np.array([0, 1, 0, 1], dtype=np.int32)
    from parelag_amd._lib import ElagHipError
    with pytest.raises(ElagHipError, match="singular"):
        amge.coarsen(hx, part)


def test_amge_spaces_in_the_blocked_darcy_cycle(amge):
    """The device-built interpolators drive the blocked AMGe V-cycle of MultigridTestDarcy: the cycle built from them equals
    the oracle's cycle built from the oracle's interpolators, and GMRES takes the same number of iterations."""
    from parelag_amd import linalg as la
    from oracle import oracle as orc
    from tests.helpers import gmres, oracle_blocked_hierarchy
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    lv = level_from_tets(seq.meshes[0])
    parts = [seq.partition(0), seq.partition(1)]
    ro = oa.coarsen_sequence(lv, parts)
    rd = amge.coarsen_sequence(lv, parts)
    M, B, Bt = lv.darcy_blocks()
    blocks = {(0, 0): M, (0, 1): Bt, (1, 0): B}
    n0, n1 = M.shape[0], B.shape[0]
    Ho, _ = oracle_blocked_hierarchy(blocks, [(r["P2"], r["P3"]) for r in ro], kind="Block GS Lower", use_negative_s=False)
    op = la.MfemBlockOperator([0, n0, n0 + n1])
    for (i, j), X in blocks.items():
        op.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(orc.csr(X)))
    Ps = [(la.ParallelCSRMatrix.from_scipy(orc.csr(r["P2"])), la.ParallelCSRMatrix.from_scipy(orc.csr(r["P3"]))) for r in rd]
    Hd = la.buildBlockedHierarchyFromDeRhamSequence(op, Ps, block_smoother="Block GS", use_negative_s=False)
    rng = np.random.default_rng(11)
    b = rng.uniform(-1, 1, n0 + n1)
    xo = np.zeros_like(b); xd = np.full_like(b, 3.0)
    Ho.Mult(b, xo)
    Hd.Mult(b, xd)
    assert np.max(np.abs(xd - xo)) < 1e-9 * np.max(np.abs(xo))
    A = sp.bmat([[M, Bt], [B, None]]).tocsr()
    rhs = np.concatenate([np.zeros(n0), lv.mass3() @ lv.const3])

    def ora(r, z):
        z[:] = 0.0
        Ho.Mult(r, z)
    x, it = gmres(lambda v: A @ v, lambda r, z: Hd.Mult(r, z), rhs, rtol=1e-8)
    xr, ito = gmres(lambda v: A @ v, ora, rhs, rtol=1e-8)
    assert it == ito and it < 100
    assert np.max(np.abs(x - xr)) < 1e-7 * np.max(np.abs(xr))


def test_upscaling_errors_equal_the_oracle_s(amge):
    """The device-built sequence upscales the Darcy problem exactly like the oracle's (``examples/Upscaling*.cpp``)."""
    dev = upscaling_errors(amge.coarsen_sequence)
    ref = upscaling_errors(oa.coarsen_sequence)
    assert np.allclose(dev, ref, rtol=1e-8)
    assert dev[1][0] > 1.7 * dev[0][0]


def test_many_targets(amge):
    """30 target vectors on the 2-forms (the count of "Upscaling order" 2): more targets than an agglomerated facet has dofs,
    so the SVD keeps at most (dofs - 1) directions next to the PV vector; subspaces compared through the projectors."""
    rng = np.random.default_rng(12)
    hx = level_from_hex((4, 4, 2), h=(1.0, 2.0, 0.5), kinv=np.exp(rng.standard_normal(32)))
    hx.T2 = np.concatenate([hx.T2, rng.standard_normal((hx.ndof2, 27))], axis=1)
    # the targets of the next form must contain the derivatives of these (``targetDerivativesInForm``, DeRhamSequence.cpp:475-570),
    # otherwise the coarse spaces cannot hold them: 28 targets on the 3-forms, RangeT bubbles for their non-constant part
    hx.T3 = np.concatenate([hx.T3, hx.D2 @ hx.T2[:, 3:]], axis=1)
    part = box_partition((4, 4, 2), (2, 2, 2))
    ro, rd = oa.coarsen(hx, part), amge.coarsen(hx, part)
    assert np.array_equal(ro["n_null"], rd["n_null"]) and np.array_equal(ro["n_rt"], rd["n_rt"]) and ro["n_rt"].sum() > 0
    for k in ("f_dof2_ptr", "e_idof2_ptr", "e_dof3_ptr"):
        assert np.array_equal(getattr(ro["coarse"], k), getattr(rd["coarse"], k)), k
    Pi2 = (rd["hat1"] + rd["hat0"] - rd["hat0"] @ (rd["P2"] @ rd["hat1"])).tocsr()
    assert rel((rd["P2"] @ Pi2).toarray(), (ro["P2"] @ ro["Pi2"]).toarray()) < 1e-8
    inv = invariants(hx, dict(rd, Pi2=Pi2))
    assert max(inv.values()) < 1e-9, inv
