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
