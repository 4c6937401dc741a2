"""All six reproducible known answers of the reference, solved END TO END on the device: the level systems are the device's own
Galerkin products ``P^T A P`` (``elag_hip_parcsr_rap`` + ``FixZeroRows``) of the fine operator through the device-built AMGe
interpolators, each solved by device PCG preconditioned with the device V-cycle of the levels below it -- with l1-Jacobi,
Chebyshev and the reference's own "L1 Gauss-Seidel" smoothers (Hiptmair-wrapped for the 1- and 2-forms) -- and the numbers the
reference's ctest greps come out of those solutions:

    UpscalingGeneralForm --form 0 | 1 | 2        testsuite/CMakeLists.txt:117-176
    Upscaling2FormAMGe                           examples/CMakeLists.txt:50-55
    LogicalPartitionerDemo --Nx 12 ...           examples/CMakeLists.txt:108-113
    EmbeddedMeshPartitionerDemo --mesh none      examples/CMakeLists.txt:122-127

The reference holds no kernel-level vectors for SpMV / l1 norms / Chebyshev / RAP (they live in hypre); these runs pin the whole
device stack -- coarsening kernels, SpGEMM, SpMV forms, smoothers, cycle, Krylov shell -- to the reference's five printed digits:
a wrong Galerkin product, a wrong l1 norm option that broke convergence, or a wrong interpolator changes them."""
import re

import numpy as np
import pytest
import scipy.sparse as sp

from parelag_amd import derham, meshgen
from parelag_amd.amge_level import box_partition
from parelag_amd.derham_level import seq_level_from_hex

from tests import test_logical_partitioner_golden as lpg
from tests import test_upscaling_golden as ups

pytestmark = pytest.mark.gpu

SMOOTHERS = ["L1 Jacobi", "Chebyshev", "L1 Gauss-Seidel"]


def device_solves(smoother, form=0, true_D=None, stats=None):
    """solve_all hook: level k's operator is RAP_device(... RAP_device(A_0, P_0) ..., P_{k-1}) with the essential rows / columns
    eliminated on level 0 and essential coarse columns eliminated in every P (``ComputeTrueP``); checked against the operator
    assembled from the coarse sequence's own matrices (M_c = P^T M P, D_f P = P D_c), then solved by PCG + V-cycle on the device.
    ``true_D[k]``: the derivative D_{form-1} of level k with essential columns eliminated (Hiptmair auxiliary space)."""
    from parelag_amd import linalg as la

    def solve_all(As, ess, rhs, lift, Ps):
        nl = len(As)
        Pe = [meshgen.eliminate_cols(sp.csr_matrix(P), ess[k + 1]) for k, P in enumerate(Ps)]
        A0 = meshgen.eliminate_rowcol(As[0], ess[0])
        ops = [la.ParallelCSRMatrix.from_scipy(A0)]
        dPe = [la.ParallelCSRMatrix.from_scipy(p) for p in Pe]
        dDe = [la.ParallelCSRMatrix.from_scipy(d) for d in true_D] if true_D is not None else None
        for dp in dPe:
            Ac = la.RAP(ops[-1], dp)                                       # the device's P^T A P
            Ac.fix_zero_rows()
            ops.append(Ac)
        sol = []
        for k in range(nl):
            Ak = meshgen.eliminate_rowcol(As[k], ess[k])
            dev = ops[k].to_scipy()
            assert abs(dev - Ak).max() <= 1e-10 * abs(Ak).max(), "device Galerkin operator differs from the coarse sequence's on level %d" % k
            kw = {}
            if dDe is not None and k < nl - 1:
                kw = dict(Ds=dDe[k:nl - 1], aux_smoother=smoother)
            H = la.buildHierarchyFromDeRhamSequence(ops[k], dPe[k:], smoother, **kw)
            K = la.KrylovSolver(ops[k], H, {"Solver name": "PCG", "Relative tolerance": 1e-12, "Absolute tolerance": 0.0, "Maximum iterations": 500})
            free, fix = ~ess[k], ess[k]
            b = np.where(fix, lift[k], rhs[k] - As[k] @ np.where(fix, lift[k], 0.0))    # EliminateRowCol(dof, value, rhs)
            x = np.zeros(As[k].shape[0])
            K.Mult(b, x)
            assert K.GetConverged(), (smoother, k)
            if stats is not None:
                stats.append((k, K.GetNumIterations()))
            sol.append(x)
        return sol
    return solve_all


@pytest.mark.parametrize("smoother", SMOOTHERS)
@pytest.mark.parametrize("form", [0, 1, 2])
def test_upscaling_general_form_on_the_device_stack(form, smoother):
    lv, part = ups.fine_level()
    fine = derham.DeRhamSequence(lv)
    fine.SetTargets(lv.T)
    fine.SetSVDTol(1e-9)
    coarse = fine.Coarsen(part)
    P = fine.GetP(form).tocsr()
    b0 = ups.natural_rhs(lv, form)
    As, ess = [], []
    for seq in (fine, coarse):
        M, W, D = seq.ComputeMassOperator(form).tocsr(), seq.ComputeMassOperator(form + 1).tocsr(), seq.GetD(form).tocsr()
        As.append((M + D.T @ W @ D).tocsr())
        ess.append(seq._marker(form, ups.ESS_ATTR))
    true_D = [fine.ComputeTrueD(form - 1, ups.ESS_ATTR)] if form > 0 else None
    stats = []
    sol = device_solves(smoother, form, true_D, stats)(As, ess, [b0, P.T @ b0], [np.zeros(a.shape[0]) for a in As], [P])
    e = P @ sol[1] - sol[0]
    de = fine.GetD(form) @ e
    l2, energy = np.sqrt(e @ (fine.ComputeMassOperator(form) @ e)), np.sqrt(de @ (fine.ComputeMassOperator(form + 1) @ de))
    assert ("%.4e" % l2, "%.4e" % energy) == ups.GOLDEN[form]
    assert dict(stats)[0] < 80 and dict(stats)[1] <= 2             # two-level cycle on 64 cells; the coarse level is a direct solve


@pytest.mark.parametrize("smoother", SMOOTHERS)
def test_upscaling_2form_amge_three_levels_on_the_device_stack(smoother):
    lv = seq_level_from_hex(ups.N3, h=(ups.H3,) * 3, jstart=0)
    seqs = [derham.DeRhamSequence(lv)]
    seqs[0].SetTargets(lv.T)
    for part in (box_partition(ups.N3, (2, 2, 2)), box_partition((4, 4, 4), (2, 2, 2))):
        seqs[-1].SetSVDTol(1e-9)
        seqs.append(seqs[-1].Coarsen(part.astype(np.int64)))
    ess_attr = np.ones(6, dtype=np.int64)
    P = [seqs[k].GetP(2).tocsr() for k in range(2)]
    M = [s.ComputeMassOperator(2).tocsr() for s in seqs]
    W = [s.ComputeMassOperator(3).tocsr() for s in seqs]
    D = [s.GetD(2).tocsr() for s in seqs]
    As = [(M[k] + D[k].T @ W[k] @ D[k]).tocsr() for k in range(3)]
    ess = [s._marker(2, ess_attr) for s in seqs]
    b = [M[0] @ lv.T[2][:, 2]]
    for k in range(2):
        b.append(P[k].T @ b[k])
    true_D = [seqs[k].ComputeTrueD(1, ess_attr) for k in range(2)]
    sol = device_solves(smoother, 2, true_D)(As, ess, b, [np.zeros(a.shape[0]) for a in As], P)
    out = []
    for k in (2, 1):
        u = sol[k]
        for j in range(k, 0, -1):
            u = P[j - 1] @ u
        e = u - sol[0]
        de = D[0] @ e
        out.append(("%.4e" % np.sqrt(e @ (M[0] @ e)), "%.4e" % np.sqrt(de @ (W[0] @ de))))
    assert tuple(zip(*out)) == ups.GOLDEN_2FORM


@pytest.mark.parametrize("smoother", SMOOTHERS)
def test_logical_partitioner_demo_four_levels_on_the_device_stack(smoother):
    def coarsen_level(lv, part):
        s = derham.DeRhamSequence(lv)
        s.SetSVDTol(1e-9)
        c = s.Coarsen(part, check_topology=True)
        return s.GetP(0), c.level, dict(A=s._res["topo"]["A"], messages=s.TopologyMessages)
    levels, Ps, msgs = lpg.four_levels(coarsen_level)
    lpg.check_structure(levels, msgs)
    assert lpg.printed_errors(levels, Ps, solve_all=device_solves(smoother)) == lpg.GOLDEN


@pytest.mark.parametrize("smoother", SMOOTHERS)
def test_embedded_mesh_partitioner_demo_on_the_device_stack(smoother):
    """inhomogeneous essential data (1 on the whole boundary) carried to the coarse levels by the device's cochain projector"""
    def coarsen_level(lv, part):
        s = derham.DeRhamSequence(lv)
        c = s.Coarsen(part, check_topology=True)
        return s.GetP(0), s.GetPi(0), c.level, dict(A=s._res["topo"]["A"], messages=s.TopologyMessages)
    lines = lpg.embedded_mesh_lines(coarsen_level, solve_all=device_solves(smoother))
    assert re.match(lpg.EMBEDDED[0], lines[0]) and re.match(lpg.EMBEDDED[1], lines[1]), lines


def test_boundary_operators_compose_to_zero_in_device_norms():
    """``testsuite/topology.cpp:109-116``: for the fine and the agglomerated topology B_i B_{i+1} = 0, checked there with the
    max-norm of the product (``BBT->MaxNorm() > 1e-12`` fails the test).  Here the products are device SpGEMMs and all four norms of
    ``hypre_ParCSRMatrixNorm*`` (infinity, one, max-entry, Frobenius: ``elag_hip_parcsr_norm``) are device reductions."""
    from parelag_amd import linalg as la
    lv = seq_level_from_hex((6, 4, 4), h=(0.5, 0.25, 0.25), jstart=0)
    topo = derham.coarsen_topology(lv, box_partition((6, 4, 4), (2, 2, 2)).astype(np.int64))
    for name, n, B in (("fine", lv.n, lv.B), ("agglomerated", topo["n"], topo["B"])):
        mats = [la.ParallelCSRMatrix.from_scipy(sp.csr_matrix((B[c][2], B[c][1], B[c][0]), shape=(n[c], n[c + 1]))) for c in range(3)]
        for c in range(2):
            BB = mats[c].matmul(mats[c + 1])
            assert mats[c].norm("linf") > 0 and mats[c + 1].norm("l1") > 0
            for kind in ("linf", "l1", "max", "frobenius"):
                assert BB.norm(kind) <= 1e-12, (name, c, kind)
