"""``examples/LogicalPartitionerDemo.cpp --Nx 12 --Ny 12 --Nz 12`` as ctest runs it (``examples/CMakeLists.txt:108-113``) and the line it
greps:   "u l2-like errors: 4.3386e-03 4.3386e-03 1.1209e-03 \\nu energy-like errors: 5.6634e-02 5.6634e-02 2.8076e-02".

What the run does (``LogicalPartitionerDemo.cpp:143-170, 213-250, 311-345, 385-395, 466-482``): the unit cube in 12^3 hexahedra; five cells of
every z-layer (the four corners and one in the middle) carry a material id of their own; four levels, agglomerates = connected
cells with the same (i/2, j/2, k/2, material) (``LogicalPartitioner`` with ``CoarsenLogicalCartesianOperatorMaterialId``), level after
level; ``CoarsenLocalPartitioning(partitioning, check_topology = 1, preserve_material_interfaces = 1)`` -- the column of single
cells through the middle drills a tunnel through the level-2 agglomerates around it, the check reports "Element .. has 1
tunnels." and splits them; all forms coarsened (``jFormStart = 0``, ``SetUpscalingTargets(3, 0)``, SVD tolerance 1e-9); the 0-form
problem ``M + D^T W D`` with right-hand side 1 and zero essential conditions on the whole boundary on every level.  The printed
numbers do not depend on the numbering, so the generator's own serves.  Exercised here beyond ``test_upscaling_golden``: three
consecutive coarsenings of algebraic levels for every form, agglomerates of irregular shape (boxes with a notch), the
topology check with de-agglomeration in its real use."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import derham as od
from parelag_amd import derham
from parelag_amd.derham_level import seq_level_from_hex
from parelag_amd.partitioning import CartesianIJK, CoarsenLogicalCartesianOperatorMaterialId, LogicalPartitioner

N = 12
GOLDEN = (("4.3386e-03", "4.3386e-03", "1.1209e-03"), ("5.6634e-02", "5.6634e-02", "2.8076e-02"))


def fine_level():
    lv = seq_level_from_hex((N, N, N), h=(1.0 / N,) * 3, jstart=0)
    mat = np.ones((N, N, N), dtype=np.int64)                       # [x, y, z]; the generator numbers (x N + y) N + z
    attr = 2
    for k in range(N):                                             # LogicalPartitionerDemo.cpp:157-165 (element k N0 N1 + j N0 + i)
        for e in (0, N - 1, N * N - 1, N * N - N, N * N // 2 - N // 2):
            mat[e % N, e // N, k] = attr
            attr += 1
    return lv, CartesianIJK.SetupCartesianIJKMaterialId((N, N, N), mat.reshape(-1), order="generator")


def element_element(lv):
    B0 = sp.csr_matrix((np.abs(lv.B[0][2]), lv.B[0][1], lv.B[0][0]), shape=(lv.n[0], lv.n[1]))
    return (B0 @ B0.T).tocsr()


def four_levels(coarsen_level):
    """coarsen_level(level, partitioning) -> (P_0, coarse level, agglomerated topology) with the checked topology."""
    lv, logical = fine_level()
    partitioner, op = LogicalPartitioner(), CoarsenLogicalCartesianOperatorMaterialId([2, 2, 2])
    levels, Ps, msgs, cur = [lv], [], [], lv
    for _l in range(3):
        part = partitioner.Partition(element_element(cur), logical, op)
        P, cur, topo = coarsen_level(cur, part)
        logical = partitioner.ComputeCoarseLogical(op, topo["A"][0], logical)
        Ps.append(P.tocsr()); levels.append(cur); msgs.append(topo["messages"])
    return levels, Ps, msgs


def direct_solves(As, ess, rhs, lift, Ps):
    """every level's system solved directly: essential dofs carry ``lift``, the free ones solve A_ff u_f = b_f - A_fe lift_e"""
    sol = []
    for A, e, b, g in zip(As, ess, rhs, lift):
        free, fix = np.nonzero(~e)[0], np.nonzero(e)[0]
        u = np.zeros(A.shape[0])
        u[fix] = g[fix]
        u[free] = spl.spsolve(A[free][:, free].tocsc(), b[free] - A[free][:, fix] @ g[fix])
        sol.append(u)
    return sol


def printed_errors(levels, Ps, solve_all=direct_solves):
    """``solve_all(As, ess, rhs, lift, Ps) -> solutions``: how the level systems are solved (directly here; on the device through
    Galerkin products, V-cycles and PCG in tests/test_gpu_known_answers.py)"""
    M = [L.mass(0).tocsr() for L in levels]
    W = [L.mass(1).tocsr() for L in levels]
    D = [L.D[0].tocsr() for L in levels]
    b = [M[0] @ np.ones(levels[0].ndof(0))]                        # DomainLFIntegrator(1): the constant is in the space
    for P in Ps:
        b.append(P.T @ b[-1])
    As = [(M[k] + D[k].T @ W[k] @ D[k]).tocsr() for k in range(len(levels))]
    ess = [derham.DeRhamSequence(L)._marker(0, np.ones(6, dtype=np.int64)) for L in levels]
    sol = solve_all(As, ess, b, [np.zeros(L.ndof(0)) for L in levels], Ps)
    l2, en = [], []
    for k in (3, 2, 1):                                            # coarsest first, as OutputUpscalingErrors prints them
        u = sol[k]
        for j in range(k, 0, -1):
            u = Ps[j - 1] @ u
        e = u - sol[0]
        de = D[0] @ e
        l2.append("%.4e" % np.sqrt(e @ (M[0] @ e))); en.append("%.4e" % np.sqrt(de @ (W[0] @ de)))
    return tuple(l2), tuple(en)


def check_structure(levels, msgs):
    assert [L.n[0] for L in levels] == [1728, 276, 108, 99]        # 216 boxes + 60 single cells; then the split agglomerates
    assert msgs[0] == []
    assert sorted(m.strip().split(" has ")[1] for m in msgs[1]) == ["1 tunnels."] * 3
    assert all("tunnels" in m for m in msgs[2]) and len(msgs[2]) == 2


def test_partitioner_on_the_fine_level():
    lv, logical = fine_level()
    part = LogicalPartitioner().Partition(element_element(lv), logical, CoarsenLogicalCartesianOperatorMaterialId([2, 2, 2]))
    sizes = np.bincount(part)
    assert part.max() + 1 == 276 and np.count_nonzero(sizes == 1) == 60 and np.count_nonzero(sizes == 6) == 30 and np.count_nonzero(sizes == 8) == 186
    first = np.array([np.nonzero(part == p)[0][0] for p in range(part.max() + 1)])
    assert np.all(np.diff(first) > 0)                              # numbered in the order of their first element
    xyz = CartesianIJK.SetupCartesianIJK((3, 4, 5))                # XYZ: element k N0 N1 + j N0 + i
    assert xyz[3 * 4 * 2 + 3 * 1 + 2].tolist() == [2, 1, 2]


def test_oracle_reproduces_the_reference_logical_partitioner_errors():
    def coarsen_level(lv, part):
        topo = od.coarsen_topology(lv, part, check=True)
        res = od.coarsen(lv, part, topo=topo)
        return res["P"][0], res["coarse"], topo
    levels, Ps, msgs = four_levels(coarsen_level)
    check_structure(levels, msgs)
    assert printed_errors(levels, Ps) == GOLDEN


@pytest.mark.gpu
def test_device_coarsening_reproduces_the_reference_logical_partitioner_errors():
    seqs = []

    def coarsen_level(lv, part):
        s = derham.DeRhamSequence(lv)
        s.SetSVDTol(1e-9)
        c = s.Coarsen(part, check_topology=True)
        assert max(s.CheckInvariants().values()) < 1e-9
        seqs.append(s)
        return s.GetP(0), c.level, dict(A=s._res["topo"]["A"], messages=s.TopologyMessages)
    levels, Ps, msgs = four_levels(coarsen_level)
    check_structure(levels, msgs)
    assert printed_errors(levels, Ps) == GOLDEN


# ---- examples/EmbeddedMeshPartitionerDemo.cpp --mesh none --par_ref_levels 2 ----------------------------------------------------

import re

from parelag_amd.amge_level import box_partition

EMBEDDED = ("u l2-like errors: 9.361[0-9]e-03 2.554[0-9]e-03 ", "u energy-like errors: 8.60[0-9]*e-02 4.292[0-9]e-02")   # examples/CMakeLists.txt:122-127


def embedded_mesh_lines(coarsen_level, solve_all=direct_solves):
    """``EmbeddedMeshPartitionerDemo.cpp:148-160, 264-300, 385-400, 452-510``: 2 x 2 x 2 hexahedra refined twice, the lower half with its
    own material id; agglomerates = connected cells with the same (derefinement partition, material) -- ``LogicalPartitioner`` with
    ``CoarsenMFEMMaterialId``; ``check_topology = 1``; the 0-form problem with no load and the essential value 1 on the whole
    boundary, carried to the coarse levels by the cochain projector (``GetPi(form)->GetProjectorMatrix().Mult(ess_data)``)."""
    n = (8, 8, 8)
    lv = seq_level_from_hex(n, h=(0.125,) * 3, jstart=0)
    idx = np.arange(512)
    mat = 1 + (idx % 8 < 4)                                        # the generator numbers (x 8 + y) 8 + z: lower half in z
    identity = lambda logical: logical
    partitioner = LogicalPartitioner()
    levels, Ps, Pis, cur, m = [lv], [], [], lv, n
    for _l in range(2):
        logical = np.stack([box_partition(m, (2, 2, 2)).astype(np.int64), mat], 1)          # (MFEM_partition, materialId)
        part = partitioner.Partition(element_element(cur), logical, identity)
        assert part.max() + 1 == logical[:, 0].max() + 1          # the derefinement agglomerates do not straddle the interface
        P, Pi, cur, topo = coarsen_level(cur, part)
        assert topo["messages"] == []
        mat = partitioner.ComputeCoarseLogical(identity, topo["A"][0], logical)[:, 1]
        m = tuple(v // 2 for v in m)
        Ps.append(P.tocsr()); Pis.append(Pi); levels.append(cur)
    ess = [derham.DeRhamSequence(L)._marker(0, np.ones(6, dtype=np.int64)) for L in levels]
    g = [ess[0].astype(np.float64)]                                # lift: ProjectBdrCoefficient(1) on every boundary attribute
    for Pi in Pis:
        g.append(Pi @ g[-1])
    M = [L.mass(0).tocsr() for L in levels]
    W = [L.mass(1).tocsr() for L in levels]
    D = [L.D[0].tocsr() for L in levels]
    As = [(M[k] + D[k].T @ W[k] @ D[k]).tocsr() for k in range(len(levels))]
    sol = solve_all(As, ess, [np.zeros(L.ndof(0)) for L in levels], g, Ps)      # EliminateRowCol(mm, ess_data(mm), rhs)
    l2, en = [], []
    for k in (2, 1):
        u = sol[k]
        for j in range(k, 0, -1):
            u = Ps[j - 1] @ u
        e = u - sol[0]
        de = D[0] @ e
        l2.append("%.4e" % np.sqrt(e @ (M[0] @ e))); en.append("%.4e" % np.sqrt(de @ (W[0] @ de)))
    return "u l2-like errors: " + " ".join(l2) + " ", "u energy-like errors: " + " ".join(en) + " "


def test_oracle_reproduces_the_embedded_mesh_demo_errors():
    def coarsen_level(lv, part):
        topo = od.coarsen_topology(lv, part, check=True)
        res = od.coarsen(lv, part, topo=topo)
        return res["P"][0], res["Pi"][0], res["coarse"], topo
    lines = embedded_mesh_lines(coarsen_level)
    assert re.match(EMBEDDED[0], lines[0]) and re.match(EMBEDDED[1], lines[1]), lines


@pytest.mark.gpu
def test_device_coarsening_reproduces_the_embedded_mesh_demo_errors():
    def coarsen_level(lv, part):
        s = derham.DeRhamSequence(lv)
        c = s.Coarsen(part, check_topology=True)
        return s.GetP(0), s.GetPi(0), c.level, dict(A=s._res["topo"]["A"], messages=s.TopologyMessages)
    lines = embedded_mesh_lines(coarsen_level)
    assert re.match(EMBEDDED[0], lines[0]) and re.match(EMBEDDED[1], lines[1]), lines
