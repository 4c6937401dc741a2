"""GPU parity of the full AMGe coarsening (``parelag_amd/derham.py``: all forms jformStart..3, ``hFacetExtension`` and
``hRidgePeakExtension`` on the device) against the CPU oracle of ``DeRhamSequence::Coarsen`` (``oracle/derham.py``) on
identical inputs: integer tables bit-exact, fp64 operators 1e-10 relative after aligning the sign of SVD columns (as in
``tests/test_gpu_amge.py``); the device's spaces satisfy the reference's invariants themselves."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import derham as od
from parelag_amd import meshgen
from parelag_amd.amge_level import box_partition
from parelag_amd.derham_level import seq_level_from_hex, seq_level_from_tets
from tests.test_derham_oracle import invariants, kuhn_boxes, rich_targets_level

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope="module")
def derham():
    from parelag_amd import derham as m
    return m


def rel(x, ref):
    den = max(abs(ref).max(), 1e-300)
    return abs(x - ref).max() / den if x.shape == ref.shape else np.inf


def signs(P_dev, P_ref):
    d = np.asarray(P_dev.multiply(P_ref).sum(axis=0)).ravel()
    return np.where(d < 0, -1.0, 1.0)


def compare(derham, lv, part):
    ro = od.coarsen(lv, part)
    tm = {}
    rd = derham.coarsen(lv, part, timings=tm)
    for c in range(4):
        for k in range(3):
            assert np.array_equal(ro["topo"]["A"][c][k], rd["topo"]["A"][c][k])
            if c < 3:
                assert np.array_equal(ro["topo"]["B"][c][k], rd["topo"]["B"][c][k])
    co, cd = ro["coarse"], rd["coarse"]
    assert ro["n_null"] == rd["n_null"]
    S = {j: sp.diags(signs(rd["P"][j], ro["P"][j])) for j in range(lv.jstart, 4)}
    Pi = {j: derham.projector(rd, j) for j in range(lv.jstart, 4)}
    errs = {}
    for j in range(lv.jstart, 4):
        for c in range(lv.cb(j) + 1):
            assert np.array_equal(co.idof[j][c], cd.idof[j][c]), (j, c)
        errs[f"P{j}"] = rel(rd["P"][j] @ S[j], ro["P"][j])
        errs[f"Pi{j}"] = rel(S[j] @ Pi[j], ro["Pi"][j])
        errs[f"M{j}"] = rel(S[j] @ cd.mass(j) @ S[j], co.mass(j))
        errs[f"T{j}"] = rel(S[j] @ cd.T[j], co.T[j])
        if j < 3:
            errs[f"D{j}"] = rel(S[j + 1] @ cd.D[j] @ S[j], co.D[j])
    assert max(errs.values()) < TOL, errs
    inv = invariants(lv, rd, Pi=Pi)
    assert max(inv.values()) < 1e-9, inv
    return ro, rd, tm


@pytest.mark.parametrize("jstart", [2, 1, 0])
def test_two_tet_levels(derham, jstart):
    seq = meshgen.Sequence.kuhn_cube(1, 2)
    lv = seq_level_from_tets(seq.meshes[0], jstart=jstart)
    _ro, rd, tm = compare(derham, lv, seq.partition(0))
    if jstart < 2:
        assert tm[f"extension{jstart}_0_ms"] > 0.0                       # the ridge / peak stage ran on the device
    compare(derham, rd["coarse"], seq.partition(1))


def test_heterogeneous_coefficient_null_dofs(derham):
    seq = meshgen.Sequence.kuhn_cube(1, 2)
    rng = np.random.default_rng(3)
    lv = seq_level_from_tets(seq.meshes[0], jstart=1, kinv=np.exp(2 * rng.standard_normal(seq.meshes[0].nt)))
    ro, rd, _tm = compare(derham, lv, seq.partition(0))
    assert ro["n_null"][2] > 0
    compare(derham, rd["coarse"], seq.partition(1))


def test_invariants_on_a_larger_mesh(derham):
    """kuhn_cube(2) refined twice (3072 elements, 384 agglomerates), all four forms: the device result is checked through
    the reference's invariants alone (the numpy oracle would take a minute here)."""
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    lv = seq_level_from_tets(seq.meshes[0], jstart=0)
    rd = derham.coarsen(lv, seq.partition(0))
    inv = invariants(lv, rd, Pi={j: derham.projector(rd, j) for j in range(4)})
    assert max(inv.values()) < 1e-9, inv
    for m in seq.meshes:
        m.build_faces()
    c = seq.meshes[1]
    assert [rd["coarse"].ndof(j) for j in range(4)] == [c.nv, c.ne, c.nf, c.nt]


def test_amge_h1_spaces_in_the_pcg_solve(derham):
    """BASELINE configs[1]: the 0-form problem ``A = D0^T M1 D0`` with essential boundary, hierarchy built from the device's
    AMGe interpolators P_0 (essential columns eliminated as ``ComputeTrueP(jform, ess)`` does): PCG converges like with the
    nested geometric interpolators."""
    from parelag_amd import linalg as la
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    lv = seq_level_from_tets(seq.meshes[0], jstart=0)
    Ps, cur = [], lv
    for l in range(2):
        r = derham.coarsen(cur, seq.partition(l))
        Ps.append(r["P"][0])
        cur = r["coarse"]
    m = seq.meshes[0]
    A = meshgen.eliminate_rowcol((lv.D[0].T @ lv.mass(1) @ lv.D[0]).tocsr(), m.boundary_marker_unit_cube())
    marks = [mm.boundary_marker_unit_cube() for mm in seq.meshes]
    # coarse vertices are the agglomerated peaks, numbered like the coarse mesh's vertices on derefined tets
    Pe = [meshgen.eliminate_cols(P, marks[l + 1]) for l, P in enumerate(Ps)]
    Pg = [meshgen.eliminate_cols(meshgen.interp_P0(seq.meshes[l + 1], seq.meshes[l]), marks[l + 1]) for l in range(2)]
    b = meshgen.load_vector_P1(m, marks[0])
    its = []
    for P_list in (Pe, Pg):
        dA = la.ParallelCSRMatrix.from_scipy(A)
        H = la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(P) for P in P_list], "L1 Jacobi")
        K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 100})
        x = np.zeros_like(b)
        K.Mult(b, x)
        assert K.GetConverged()
        assert np.linalg.norm(A @ x - b) < 1e-6 * np.linalg.norm(b)
        its.append(K.GetNumIterations())
    assert abs(its[0] - its[1]) <= 2, its


def test_box_agglomerates_of_48_tets(derham):
    """Non-nested agglomerates (2 x 2 x 2 Kuhn cells), heterogeneous coefficient: null-space dofs, RangeT bubbles of the
    1-forms, local systems beyond the LDS budget (global scratch) in both extension modes; then the whole cube as one
    agglomerate."""
    lv, part = kuhn_boxes()
    ro, rd, _tm = compare(derham, lv, part)
    assert ro["n_null"][2] == 24
    compare(derham, rd["coarse"], np.zeros(8, dtype=np.int64))


def test_full_size_cube456_x3_all_forms(derham):
    """BASELINE's size (cube456 refined three times: 233 472 elements, 29 184 agglomerates on level 0), all four forms, checked
    through the size-independent properties the reference itself checks (``CheckInvariants``): commutation, exactness of the
    coarse complex, Galerkin masses, ``Pi P = I``, targets in the coarse spaces."""
    seq = meshgen.Sequence.cube456(3)
    lv = seq_level_from_tets(seq.meshes[0], jstart=0)
    rd = derham.coarsen(lv, seq.partition(0))
    c = rd["coarse"]
    assert c.n == [29184, 60016, 36570, 5739] and [c.ndof(j) for j in range(4)] == [5739, 36570, 60016, 29184]
    inv = invariants(lv, rd, Pi={j: derham.projector(rd, j) for j in range(4)})
    assert max(inv.values()) < 1e-9, inv


def test_richer_targets_on_every_form(derham):
    """Upscaling order 1 on all four forms: many null-space dofs with clustered singular values, so the bases differ by
    rotations inside the kept subspaces -- compared through what does not depend on the basis: the numbers of coarse dofs
    per entity and the projectors ``P_j Pi_j`` onto the coarse spaces; the device's spaces satisfy the invariants themselves."""
    seq = meshgen.Sequence.kuhn_cube(1, 2)
    lv = rich_targets_level(seq)
    ro = od.coarsen(lv, seq.partition(0))
    rd = derham.coarsen(lv, seq.partition(0))
    assert ro["n_null"] == rd["n_null"]
    Pi = {j: derham.projector(rd, j) for j in range(4)}
    for j in range(4):
        for c in range(lv.cb(j) + 1):
            assert np.array_equal(ro["coarse"].idof[j][c], rd["coarse"].idof[j][c]), (j, c)
        assert rel((rd["P"][j] @ Pi[j]).toarray(), (ro["P"][j] @ ro["Pi"][j]).toarray()) < 1e-8, j
    inv = invariants(lv, rd, Pi=Pi)
    assert max(inv.values()) < 1e-9, inv
    rd1 = derham.coarsen(rd["coarse"], seq.partition(1))
    inv = invariants(rd["coarse"], rd1, Pi={j: derham.projector(rd1, j) for j in range(4)})
    assert max(inv.values()) < 1e-9, inv


def test_derham_sequence_class_reads_like_the_reference_driver(derham):
    """The flow of ``examples/MultigridTest0Form.cpp:214-300`` against the class mirror: SetjformStart / SetTargets /
    SetSVDTol / Coarsen level by level / CheckInvariants / ComputeTrueP with essential labels."""
    from parelag_amd import linalg as la
    seq = meshgen.Sequence.kuhn_cube(2, 2)
    sequence = [derham.DeRhamSequence(seq_level_from_tets(seq.meshes[0], jstart=0))]
    sequence[0].SetjformStart(0)
    sequence[0].SetTargets(sequence[0].level.T)
    for i in range(2):
        sequence[i].SetSVDTol(1e-9)
        sequence.append(sequence[i].Coarsen(seq.partition(i)))
        defects = sequence[i].CheckInvariants()
        assert max(defects.values()) < 1e-9
    for m in seq.meshes:
        m.build_faces()
    assert [sequence[2].GetNumberOfDofs(j) for j in range(4)] == [seq.meshes[2].nv, seq.meshes[2].ne, seq.meshes[2].nf, seq.meshes[2].nt]
    ess = np.ones(6, dtype=int)                              # every boundary attribute essential (MultigridTest0Form.cpp:194-196)
    A = sequence[0].ComputeTrueDerivativeOperator(0).T @ sequence[0].ComputeTrueMassOperator(1) @ sequence[0].ComputeTrueDerivativeOperator(0)
    A = meshgen.eliminate_rowcol(A.tocsr(), sequence[0]._marker(0, ess))
    Ps = [sequence[i].ComputeTrueP(0, ess) for i in range(2)]
    assert abs(Ps[0] - meshgen.eliminate_cols(sequence[0].GetP(0), seq.meshes[1].boundary_marker_unit_cube())).max() == 0.0
    dA = la.ParallelCSRMatrix.from_scipy(A)
    H = la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(P) for P in Ps], "L1 Jacobi")
    K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 60})
    b = meshgen.load_vector_P1(seq.meshes[0], seq.meshes[0].boundary_marker_unit_cube())
    x = np.zeros_like(b)
    K.Mult(b, x)
    assert K.GetConverged() and K.GetNumIterations() < 25
    # selective labels: only the x = 0 side essential
    only_x0 = np.array([1, 0, 0, 0, 0, 0])
    mk = sequence[1]._marker(0, only_x0)
    assert 0 < mk.sum() < sequence[1]._marker(0, ess).sum()
    assert sequence[0].GetM(2, 1).shape == (seq.meshes[0].ne, seq.meshes[0].ne)      # ridge mass of the 1-forms: one dof per edge
    # testsuite/coarseSpaces.cpp:193-198: the targets are interpolated without error on every level, a rough field is not
    for jf in range(4):
        rows = sequence[0].ComputeSpaceInterpolationError(jf, sequence[0].level.T[jf])
        assert [r[0] for r in rows] == [sequence[1].GetNumberOfDofs(jf), sequence[2].GetNumberOfDofs(jf)]
        assert all(r[1].max() < 1e-9 and (r[2] is None or r[2].max() < 1e-9) for r in rows), (jf, rows)
    rough = np.random.default_rng(5).standard_normal((sequence[0].GetNumberOfDofs(0), 2))
    rows = sequence[0].ComputeSpaceInterpolationError(0, rough)
    assert 0.1 < rows[0][1].min() and np.all(rows[1][1] >= rows[0][1] - 1e-12)
    with pytest.raises(RuntimeError, match="Coarsen"):
        sequence[2].GetP(0)
    with pytest.raises(ValueError):
        derham.DeRhamSequence(seq_level_from_tets(seq.meshes[2], jstart=2)).SetjformStart(0)


def test_hex_grid_all_forms(derham):
    """Cartesian hex cells (MultigridTestSPE10's element type), all four forms, heterogeneous coefficient, ragged boxes."""
    n, h = (5, 4, 3), (2.0, 1.0, 0.5)
    lv = seq_level_from_hex(n, h, jstart=0, kinv=np.exp(np.random.default_rng(4).standard_normal(60)))
    _ro, rd, _tm = compare(derham, lv, box_partition(n, (2, 2, 2)))
    compare(derham, rd["coarse"], box_partition((3, 2, 2), (3, 2, 2)))
