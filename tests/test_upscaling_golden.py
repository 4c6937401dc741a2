"""The reference's known answers for the AMGe coarse spaces of every form: ``testsuite/UpscalingGeneralForm.cpp`` run as
ctest runs it (``testsuite/CMakeLists.txt:97-178``: ``--form 0|1|2 --nref_parallel 1``, no mesh file) and the lines ctest greps:

    form 0:  "u l2-like errors: 1.8389e-02 \\nu energy-like errors: 2.1485e-01"
    form 1:  "u l2-like errors: 3.1436e-02 \\nu energy-like errors: 3.2016e-01"
    form 2:  "u l2-like errors: 9.1847e-03 \\nu energy-like errors: 1.2515e-01"

What that run does (``UpscalingGeneralForm.cpp:206-252, 441-510, 515-560, 640-760``): the unit cube in 2 x 2 x 2 hexahedra, refined
once (64 cells); one coarse level of 8 agglomerates (``MFEMRefinedMeshPartitioner``: the 2 x 2 x 2 children of each parent);
``jFormStart = 0``, unit coefficients, ``SetUpscalingTargets(3, 0)`` (1, z, y, x for the 0-forms; the constant vector
fields; the constant), SVD tolerance 1e-9; ``A = M_form + D^T M_{form+1} D`` on both levels with the coarse right-hand side
``P^T b``; essential zero conditions on the four lateral sides (boundary attributes 2..5), natural data -1 on attribute 1
(``BoundaryLFIntegrator`` / ``VectorFEBoundaryTangentLFIntegrator`` of (1,1,1) / ``VectorFEBoundaryFluxLFIntegrator``), nothing
on attribute 6; printed are ``|| P u_H - u_h ||_M`` and ``|| D (P u_H - u_h) ||_{M_{form+1}}``.  Both are independent of how the
mesh entities are numbered and of which pair of opposite sides carries the natural data, so the generator's own numbering
(attributes 1, 2 = the sides x = 0, x = 1) serves.  The linear systems are solved directly here (the reference stops its
PCG at 1e-6, which does not reach the printed digits: its PETSc and hypre runs print the same numbers).

The CPU half pins the *oracle's* coarsening of all four forms to these numbers; the GPU half runs the product's
(``DeRhamSequence.Coarsen`` on the device kernels) through the same driver."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from oracle import derham as od
from parelag_amd import derham
from parelag_amd.amge_level import box_partition
from parelag_amd.derham_level import seq_level_from_hex

N, H = (4, 4, 4), 0.25
GOLDEN = {0: ("1.8389e-02", "2.1485e-01"), 1: ("3.1436e-02", "3.2016e-01"), 2: ("9.1847e-03", "1.2515e-01")}
ESS_ATTR = np.array([0, 0, 1, 1, 1, 1])                      # the four sides around the axis that carries the natural data


def natural_rhs(lv, form):
    """The boundary linear form on the side with attribute 1 (outward normal n = -e_x), data -1 / (1,1,1)."""
    ny, nz = N[1], N[2]
    yz = lambda v: np.array([(v // (nz + 1)) % (ny + 1), v % (nz + 1)])          # generator: vertex (i, j, k) -> (i (ny+1) + j)(nz+1) + k
    b = np.zeros(lv.ndof(form))
    p1, i1, _s1 = lv.B[1]
    p2, i2, s2 = lv.B[2]
    for f in np.nonzero(lv.facet_attr == 1)[0]:
        if form == 2:
            b[f] = -1.0                                    # dof = outward flux: (phi . n) integrates to one
            continue
        for r in i1[p1[f]:p1[f + 1]]:
            v = i2[p2[r]:p2[r + 1]]
            if form == 0:
                b[v] += -H * H / 4 / 2                     # Q1 on a square: a quarter of the area per vertex (met through two edges)
            else:
                tail, head = (v[0], v[1]) if s2[p2[r]] < 0 else (v[1], v[0])
                d = yz(head) - yz(tail)                    # unit tangent in the (y, z) plane; n x (1,1,1) = (0, 1, -1)
                b[r] += (H / 2) * (d[0] - d[1])            # dof = line integral: the edge function integrates to h / 2 over the face
    return b


def upscaling_errors(form, fine, coarse, P):
    b0 = natural_rhs(fine.level, form)
    sol = []
    for seq, b in ((fine, b0), (coarse, P.T @ b0)):
        M, W, D = seq.ComputeMassOperator(form).tocsr(), seq.ComputeMassOperator(form + 1).tocsr(), seq.GetD(form).tocsr()
        A = (M + D.T @ W @ D).tocsr()
        free = np.nonzero(~seq._marker(form, ESS_ATTR))[0]
        u = np.zeros(A.shape[0])
        u[free] = spl.spsolve(A[free][:, free].tocsc(), b[free])
        sol.append(u)
    e = P @ sol[1] - sol[0]
    de = fine.GetD(form) @ e
    return np.sqrt(e @ (fine.ComputeMassOperator(form) @ e)), np.sqrt(de @ (fine.ComputeMassOperator(form + 1) @ de))


def fine_level():
    return seq_level_from_hex(N, h=(H, H, H), jstart=0), box_partition(N, (2, 2, 2)).astype(np.int64)


@pytest.fixture(scope="module")
def oracle_two_levels():
    lv, part = fine_level()
    assert [lv.T[j].shape[1] for j in range(4)] == [4, 3, 3, 1]                  # SetUpscalingTargets(3, 0)
    res = od.coarsen(lv, part, svd_tol=1e-9)
    return derham.DeRhamSequence(lv), derham.DeRhamSequence(res["coarse"]), res


@pytest.mark.parametrize("form", [0, 1, 2])
def test_oracle_reproduces_the_reference_upscaling_errors(oracle_two_levels, form):
    fine, coarse, res = oracle_two_levels
    l2, energy = upscaling_errors(form, fine, coarse, res["P"][form].tocsr())
    assert ("%.4e" % l2, "%.4e" % energy) == GOLDEN[form]


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 1, 2])
def test_device_coarsening_reproduces_the_reference_upscaling_errors(form):
    lv, part = fine_level()
    fine = derham.DeRhamSequence(lv)
    fine.SetjformStart(0)
    fine.SetTargets(lv.T)
    fine.SetSVDTol(1e-9)
    coarse = fine.Coarsen(part)
    l2, energy = upscaling_errors(form, fine, coarse, fine.GetP(form).tocsr())
    assert ("%.4e" % l2, "%.4e" % energy) == GOLDEN[form]
    assert max(fine.CheckInvariants().values()) < 1e-9


def grid_mesh(n):
    """Vertices and element -> vertex table of the generator's Cartesian grid on the unit box."""
    dim = len(n)
    ax = [np.linspace(0.0, 1.0, m + 1) for m in n]
    V = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, dim)
    vid = np.arange(V.shape[0]).reshape([m + 1 for m in n])
    cells = np.stack(np.meshgrid(*[np.arange(m) for m in n], indexing="ij"), -1).reshape(-1, dim)
    corners = np.stack(np.meshgrid(*[[0, 1]] * dim, indexing="ij"), -1).reshape(-1, dim)
    ev = np.stack([vid[tuple((cells + c).T)] for c in corners], 1)
    return V, ev


def test_geometric_box_partitioner_known_answers():
    """``testsuite/CMakeLists.txt:240-245``: 12 x 16 quadrilaterals, 9 partitions -> "mean size: 21.3333"; and the call of
    ``UpscalingGeneralForm --geometric`` (``:176-183``: 64 cells, 4 partitions asked) gives the eight 2 x 2 x 2 boxes, hence the same
    printed errors as the derefinement coarsening."""
    from parelag_amd.partitioning import GeometricBoxPartitioner
    gbp = GeometricBoxPartitioner()
    V, ev = grid_mesh((12, 16))
    part = gbp.doPartition(V, ev, 9)
    sizes = gbp.GetPartitionSizes()
    assert sizes.size == 9 and "%g" % (sizes.sum() / sizes.size) == "21.3333" and part.max() == 8
    V, ev = grid_mesh(N)
    part = gbp.doPartition(V, ev, 64 // 8 // 2)                                    # level_nElements[1] = GetNE() / 2 before refining
    assert gbp.GetPartitionSizes().tolist() == [8] * 8
    ref = box_partition(N, (2, 2, 2))
    pairs = np.unique(np.stack([part, ref], 1), axis=0)
    assert pairs.shape[0] == 8                                                     # the same agglomerates, numbered differently


# ---- examples/Upscaling2FormAMGe.cpp: three levels of the (H(div), L2) pair ----------------------------------------------------

GOLDEN_2FORM = (("1.9010e-02", "3.9570e-03"), ("1.2883e-01", "5.7793e-02"))       # examples/CMakeLists.txt:50-55 (coarsest level first)
N3, H3 = (8, 8, 8), 0.125


def three_level_errors(M, W, D, Ps, ess, b0):
    """``Upscaling2FormAMGe.cpp:222-235, 282-345, 372-410``: ``A = M + D^T W D`` per level, every side essential (zero normal
    flux), coarse right-hand sides ``P^T b``; errors of the interpolated coarse solutions against the fine one."""
    b, sol = [b0], []
    for P in Ps:
        b.append(P.T @ b[-1])
    for k in range(3):
        A = (M[k] + D[k].T @ W[k] @ D[k]).tocsr()
        free = np.nonzero(~ess[k])[0]
        u = np.zeros(A.shape[0])
        u[free] = spl.spsolve(A[free][:, free].tocsc(), b[k][free])
        sol.append(u)
    out = []
    for k in (2, 1):
        u = sol[k]
        for j in range(k, 0, -1):
            u = Ps[j - 1] @ u
        e = u - sol[0]
        de = D[0] @ e
        out.append(("%.4e" % np.sqrt(e @ (M[0] @ e)), "%.4e" % np.sqrt(de @ (W[0] @ de))))
    return tuple(zip(*out))


def three_levels_general(coarsen_fn, jstart):
    """Unit cube, 2 x 2 x 2 hexahedra refined twice (512 cells); derefinement agglomerates: 64 boxes of 2 x 2 x 2 cells, then
    8 boxes of 2 x 2 x 2 of those; right-hand side the domain integral of (0,0,1) . phi = M x (interpolant of the constant field)."""
    lv = seq_level_from_hex(N3, h=(H3, H3, H3), jstart=jstart)
    r0 = coarsen_fn(lv, box_partition(N3, (2, 2, 2)).astype(np.int64))
    r1 = coarsen_fn(r0["coarse"], box_partition((4, 4, 4), (2, 2, 2)).astype(np.int64))
    levels = [lv, r0["coarse"], r1["coarse"]]
    ess = [derham.DeRhamSequence(L)._marker(2, np.ones(6, dtype=np.int64)) for L in levels]
    return three_level_errors([L.mass(2).tocsr() for L in levels], [L.mass(3).tocsr() for L in levels], [L.D[2].tocsr() for L in levels],
                              [r0["P"][2].tocsr(), r1["P"][2].tocsr()], ess, lv.mass(2) @ lv.T[2][:, 2])


def test_oracle_reproduces_the_reference_three_level_hdiv_errors():
    assert three_levels_general(od.coarsen, jstart=2) == GOLDEN_2FORM


@pytest.mark.gpu
def test_device_coarsening_reproduces_the_reference_three_level_hdiv_errors():
    assert three_levels_general(derham.coarsen, jstart=0) == GOLDEN_2FORM          # jFormStart = 0 as the example sets it


@pytest.mark.gpu
def test_two_form_fast_path_reproduces_the_reference_three_level_hdiv_errors():
    """The same numbers through ``parelag_amd/amge.py`` (``jformStart_ = 2``, vectorised tables) on ``AmgeLevel``."""
    from parelag_amd import amge
    from parelag_amd.amge_level import level_from_hex
    lv = level_from_hex(N3, h=(H3, H3, H3))
    rs = amge.coarsen_sequence(lv, [box_partition(N3, (2, 2, 2)), box_partition((4, 4, 4), (2, 2, 2))])
    levels = [lv, rs[0]["coarse"], rs[1]["coarse"]]
    ess = []
    for L in levels:
        mk = np.zeros(L.ndof2, dtype=bool)
        for f in np.nonzero(L.facet_attr > 0)[0]:
            mk[L.f_dof2_ptr[f]:L.f_dof2_ptr[f + 1]] = True
        ess.append(mk)
    got = three_level_errors([L.mass2() for L in levels], [L.mass3() for L in levels], [L.D2.tocsr() for L in levels],
                             [rs[0]["P2"].tocsr(), rs[1]["P2"].tocsr()], ess, lv.mass2() @ lv.T2[:, 2])
    assert got == GOLDEN_2FORM


@pytest.mark.gpu
def test_three_level_hdiv_errors_through_the_device_solver_stack():
    """End to end as ``Upscaling2FormAMGe.cpp:305-410`` runs it: on every level k the system ``M + D^T W D`` (essential rows
    and columns eliminated) is solved by PCG preconditioned with the multilevel Hiptmair hierarchy of the levels k .. 2
    (``MLHiptmairSolver<HdivProblem3D>(seqs, ess_attr)``) -- here ``KrylovSolver`` over ``buildHierarchyFromDeRhamSequence``
    with the device's AMGe interpolators ``ComputeTrueP(2, ess)`` and true curls ``ComputeTrueD(1, ess)`` -- and the printed errors
    come out of the device solutions."""
    from parelag_amd import linalg as la, meshgen
    lv = seq_level_from_hex(N3, h=(H3, H3, H3), jstart=0)
    seqs = [derham.DeRhamSequence(lv)]
    seqs[0].SetTargets(lv.T)
    for part in (box_partition(N3, (2, 2, 2)), box_partition((4, 4, 4), (2, 2, 2))):
        seqs[-1].SetSVDTol(1e-9)
        seqs.append(seqs[-1].Coarsen(part.astype(np.int64)))
    ess_attr = np.ones(6, dtype=np.int64)
    P = [seqs[k].GetP(2).tocsr() for k in range(2)]
    Pe = [seqs[k].ComputeTrueP(2, ess_attr) for k in range(2)]
    Ce = [seqs[k].ComputeTrueD(1, ess_attr) for k in range(3)]
    M = [s.ComputeMassOperator(2).tocsr() for s in seqs]
    W = [s.ComputeMassOperator(3).tocsr() for s in seqs]
    D = [s.GetD(2).tocsr() for s in seqs]
    b = [M[0] @ lv.T[2][:, 2]]
    for k in range(2):
        b.append(P[k].T @ b[k])
    sol, its = [], []
    for k in range(3):
        ess = seqs[k]._marker(2, ess_attr)
        A = meshgen.eliminate_rowcol((M[k] + D[k].T @ W[k] @ D[k]).tocsr(), ess)
        rhs = np.where(ess, 0.0, b[k])
        dA = la.ParallelCSRMatrix.from_scipy(A)
        H = la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(p) for p in Pe[k:]], "L1 Jacobi",
                                                Ds=[la.ParallelCSRMatrix.from_scipy(c) for c in Ce[k:]] if k < 2 else None)
        K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-12, "Absolute tolerance": 0.0, "Maximum iterations": 500})
        x = np.zeros_like(rhs)
        K.Mult(rhs, x)
        assert K.GetConverged()
        sol.append(x)
        its.append(K.GetNumIterations())
    assert its[0] < 40 and its[2] <= 2, its                  # multilevel Hiptmair on 512 cells; the coarsest level is solved directly
    out = []
    for k in (2, 1):
        u = sol[k]
        for j in range(k, 0, -1):
            u = P[j - 1] @ u
        e = u - sol[0]
        de = D[0] @ e
        out.append(("%.4e" % np.sqrt(e @ (M[0] @ e)), "%.4e" % np.sqrt(de @ (W[0] @ de))))
    assert tuple(zip(*out)) == GOLDEN_2FORM


@pytest.mark.gpu
def test_geometric_variant_with_checked_topology_reproduces_the_form1_errors():
    """``UpscalingGeneralForm --form 1 --nref_parallel 1 --geometric`` (``testsuite/CMakeLists.txt:176-183``): the partition comes from
    ``GeometricBoxPartitioner`` (4 parts asked, 8 boxes made) and ``CoarsenLocalPartitioning`` runs with ``check_topology = true``
    (``UpscalingGeneralForm.cpp:386-388``); the boxes are balls with disc facets, so nothing is de-agglomerated and the same two
    numbers are printed."""
    from parelag_amd.partitioning import GeometricBoxPartitioner
    lv, _part = fine_level()
    V, ev = grid_mesh(N)
    part = GeometricBoxPartitioner().doPartition(V, ev, 4)
    fine = derham.DeRhamSequence(lv)
    fine.SetjformStart(0)
    fine.SetTargets(lv.T)
    fine.SetSVDTol(1e-9)
    coarse = fine.Coarsen(part, check_topology=True)
    assert fine.TopologyMessages == [] and coarse.level.n[0] == 8
    l2, energy = upscaling_errors(1, fine, coarse, fine.GetP(1).tocsr())
    assert ("%.4e" % l2, "%.4e" % energy) == GOLDEN[1]


@pytest.mark.gpu
@pytest.mark.parametrize("form", [0, 1, 2])
def test_two_level_errors_through_the_device_solver_stack(form):
    """``UpscalingGeneralForm`` solved as the reference solves it -- PCG with a multilevel preconditioner built on the sequence
    (``UpscalingHypreSolver``: AMG / AMS / ADS there; here the AMGe hierarchy of the two levels, with the Hiptmair smoother through
    ``D_{form-1}`` for the 1- and 2-forms) -- on the device, tolerance tight enough for the printed digits."""
    from parelag_amd import linalg as la, meshgen
    lv, part = fine_level()
    fine = derham.DeRhamSequence(lv)
    fine.SetTargets(lv.T)
    fine.SetSVDTol(1e-9)
    coarse = fine.Coarsen(part)
    P = fine.GetP(form).tocsr()
    b0 = natural_rhs(lv, form)
    sol = []
    for seq, b, Ps in ((fine, b0, [fine.ComputeTrueP(form, ESS_ATTR)]), (coarse, P.T @ b0, [])):
        M, W, D = seq.ComputeMassOperator(form).tocsr(), seq.ComputeMassOperator(form + 1).tocsr(), seq.GetD(form).tocsr()
        ess = seq._marker(form, ESS_ATTR)
        A = meshgen.eliminate_rowcol((M + D.T @ W @ D).tocsr(), ess)
        dA = la.ParallelCSRMatrix.from_scipy(A)
        Ds = None
        if form > 0 and Ps:
            Ds = [la.ParallelCSRMatrix.from_scipy(fine.ComputeTrueD(form - 1, ESS_ATTR))]
        H = la.buildHierarchyFromDeRhamSequence(dA, [la.ParallelCSRMatrix.from_scipy(p) for p in Ps], "L1 Jacobi", Ds=Ds)
        K = la.KrylovSolver(dA, H, {"Solver name": "PCG", "Relative tolerance": 1e-12, "Absolute tolerance": 0.0, "Maximum iterations": 500})
        x = np.zeros(A.shape[0])
        K.Mult(np.where(ess, 0.0, b), x)
        assert K.GetConverged() and K.GetNumIterations() < 60
        sol.append(x)
    e = P @ sol[1] - sol[0]
    de = fine.GetD(form) @ e
    l2, energy = np.sqrt(e @ (fine.ComputeMassOperator(form) @ e)), np.sqrt(de @ (fine.ComputeMassOperator(form + 1) @ de))
    assert ("%.4e" % l2, "%.4e" % energy) == GOLDEN[form]
