"""Device Krylov shells against the oracle's restatement of mfem::CGSolver / GMRESSolver / FGMRESSolver and
parelag::StationarySolver: same iteration counts, same convergence flags, iterates within rounding."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import oracle as orc
from tests.helpers import oracle_h1_hierarchy, oracle_blocked_hierarchy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    from parelag_amd import linalg
    return linalg


def close(x, ref, tol):
    return np.max(np.abs(np.asarray(x) - np.asarray(ref))) <= tol * max(np.max(np.abs(ref)), 1e-300)


def h1_setup(la, prob, smoother="L1 Jacobi"):
    A, Ps, _ = prob
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, smoother)
    h = oracle_h1_hierarchy(A, Ps, smoother=la.HYPRE_SMOOTHER_TYPES[smoother])
    Hc = orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"])

    def oprec(r, z):
        Hc.apply(r, z)
    return dA, H, oprec


@pytest.mark.parametrize("use_graph", [True, False])
@pytest.mark.parametrize("check_every", [1, 8])
def test_pcg_with_vcycle_matches_mfem_cg(la, h1_k2, use_graph, check_every):
    """MultigridTest0Form's solve: PCG preconditioned by the AMGe cycle (examples/MultigridTest0Form.cpp:513-600)."""
    A = h1_k2[0]
    dA, H, oprec = h1_setup(la, h1_k2)
    b = np.random.default_rng(1).uniform(-1, 1, A.shape[0])
    params = {"Solver name": "PCG", "Relative tolerance": 1e-10, "Absolute tolerance": 1e-14, "Maximum iterations": 100,
              "Use graph": use_graph, "Check every": check_every}
    K = la.KrylovSolver(dA, H, params)
    x = np.full(A.shape[0], 5.0)
    K.Mult(b, x)
    xr = np.zeros_like(b)
    ref = orc.mfem_cg(lambda v: A @ v, oprec, b, xr, rel_tol=1e-10, abs_tol=1e-14, max_iter=100)
    assert K.GetConverged() and ref.converged
    assert K.GetNumIterations() == ref.final_iter and 3 < ref.final_iter < 40
    assert np.isclose(K.GetFinalNorm(), ref.final_norm, rtol=1e-5)
    assert close(x, xr, 1e-9)
    assert np.linalg.norm(A @ x - b) <= 1e-8 * np.linalg.norm(b)
    # device vectors, second solve on the same object, iterative_mode from the previous solution
    db, dx = la.DeviceVector.from_numpy(b), la.DeviceVector.from_numpy(x)
    K.iterative_mode = True
    K.Mult(db, dx)
    xr2 = x.copy()    # (the relative test restarts from the new, tiny initial residual)
    ref2 = orc.mfem_cg(lambda v: A @ v, oprec, b, xr2, iterative_mode=True, rel_tol=1e-10, abs_tol=1e-14, max_iter=100)
    assert K.GetConverged() == bool(ref2.converged) and abs(K.GetNumIterations() - ref2.final_iter) <= 1
    assert close(dx.to_numpy(), xr2, 1e-9)


def test_cg_stops_at_maximum_iterations_and_unpreconditioned(la, h1_k1):
    A = h1_k1[0]
    dA = la.ParallelCSRMatrix.from_scipy(A)
    b = np.random.default_rng(2).uniform(-1, 1, A.shape[0])
    for maxit in (0, 1, 7):
        K = la.KrylovSolver(dA, None, {"Solver name": "CG", "Relative tolerance": 1e-12, "Maximum iterations": maxit})
        x = np.zeros_like(b); xr = np.zeros_like(b)
        K.Mult(b, x)
        ref = orc.mfem_cg(lambda v: A @ v, None, b, xr, rel_tol=1e-12, max_iter=maxit)
        assert (K.GetConverged(), K.GetNumIterations()) == (bool(ref.converged), ref.final_iter)
        assert close(x, xr, 1e-11) and np.isclose(K.GetFinalNorm(), ref.final_norm, rtol=1e-9)
    # zero right-hand side: converged at iteration 0
    K = la.KrylovSolver(dA, None, {"Solver name": "CG", "Relative tolerance": 1e-12, "Maximum iterations": 5})
    x = np.ones_like(b)
    K.Mult(np.zeros_like(b), x)
    assert K.GetConverged() and K.GetNumIterations() == 0 and not x.any()
    # a smoother as preconditioner
    S = la.HypreSmootherWrapper(dA, 1)
    K = la.KrylovSolver(dA, S, {"Solver name": "PCG", "Relative tolerance": 1e-8, "Maximum iterations": 500})
    x = np.zeros_like(b); xr = np.zeros_like(b)
    K.Mult(b, x)
    l1 = orc.l1_norms(orc.diag_first(A), None, 1)
    ref = orc.mfem_cg(lambda v: A @ v, lambda r, z: z.__setitem__(slice(None), r / l1), b, xr, rel_tol=1e-8, max_iter=500)
    assert K.GetNumIterations() == ref.final_iter and close(x, xr, 1e-8)
    with pytest.raises(ValueError, match="Bad solver type"):
        la.KrylovSolver(dA, None, {"Solver name": "TFQMR"})


@pytest.mark.parametrize("name", ["GMRES", "FGMRES"])
@pytest.mark.parametrize("restart", [50, 7])
def test_gmres_on_the_darcy_system_matches_mfem(la, name, restart):
    """MultigridTestDarcy's solve: (F)GMRES preconditioned by the blocked AMGe cycle (darcy_example_parameters.xml)."""
    from parelag_amd import meshgen
    d = meshgen.darcy_problem(0, seq=meshgen.Sequence.kuhn_cube(2, 2))
    blocks = {(0, 0): d["M"], (0, 1): d["Bt"], (1, 0): d["B"]}
    n0, n1 = d["M"].shape[0], d["B"].shape[0]
    op = la.MfemBlockOperator([0, n0, n0 + n1])
    for (i, j), M in blocks.items():
        op.SetBlock(i, j, la.ParallelCSRMatrix.from_scipy(orc.csr(M)))
    Ps = [[la.ParallelCSRMatrix.from_scipy(orc.csr(P)) for P in Pl] for Pl in d["Ps"]]
    H = la.buildBlockedHierarchyFromDeRhamSequence(op, Ps, block_smoother="Block GS", use_negative_s=False)
    Ho, _ = oracle_blocked_hierarchy(blocks, d["Ps"], kind="Block GS Lower", use_negative_s=False)
    A = sp.bmat([[d["M"], d["Bt"]], [d["B"], None]]).tocsr()
    b = np.concatenate([np.zeros(n0), d["seq"].meshes[0].M3() @ np.ones(n1)])

    def oprec(r, z):
        z[:] = 0.0
        Ho.Mult(r, z)
    params = {"Solver name": name, "Relative tolerance": 1e-6, "Absolute tolerance": 1e-12, "Maximum iterations": 300, "Restart size": restart}
    K = la.KrylovSolver(op, H, params)
    x = np.full(n0 + n1, 2.0)
    K.Mult(b, x)
    xr = np.zeros_like(b)
    ref = orc.mfem_gmres(lambda v: A @ v, oprec, b, xr, rel_tol=1e-6, abs_tol=1e-12, max_iter=300, m=restart, flexible=(name == "FGMRES"))
    assert K.GetConverged() and ref.converged
    assert abs(K.GetNumIterations() - ref.final_iter) <= 1 and ref.final_iter < 120
    if K.GetNumIterations() == ref.final_iter:
        assert np.isclose(K.GetFinalNorm(), ref.final_norm, rtol=1e-3)
        assert close(x, xr, 1e-6)
    assert np.linalg.norm(A @ x - b) <= 1e-4 * np.linalg.norm(b)


def test_gmres_unpreconditioned_small_steps_and_max_iter(la):
    rng = np.random.default_rng(3)
    n = 300
    A = orc.csr(sp.random(n, n, density=0.03, random_state=rng) + sp.diags(np.full(n, 3.0)))
    dA = la.ParallelCSRMatrix.from_scipy(A)
    b = rng.uniform(-1, 1, n)
    for name, maxit, m in (("GMRES", 5, 50), ("GMRES", 12, 5), ("FGMRES", 9, 4), ("GMRES", 0, 10)):
        K = la.KrylovSolver(dA, None, {"Solver name": name, "Maximum iterations": maxit, "Restart size": m, "Check every": 3})
        x = np.zeros(n); xr = np.zeros(n)
        K.Mult(b, x)
        ref = orc.mfem_gmres(lambda v: A @ v, None, b, xr, max_iter=maxit, m=m, flexible=(name == "FGMRES"))
        assert (K.GetConverged(), K.GetNumIterations()) == (bool(ref.converged), ref.final_iter)
        assert close(x, xr, 1e-11) if maxit else not x.any()
        assert np.isclose(K.GetFinalNorm(), ref.final_norm, rtol=1e-9)
    # converges in the middle of a cycle; iterative_mode restart from the solution converges at once
    K = la.KrylovSolver(dA, None, {"Solver name": "GMRES", "Relative tolerance": 1e-10, "Maximum iterations": 200, "Restart size": 20})
    x = np.zeros(n); xr = np.zeros(n)
    K.Mult(b, x)
    ref = orc.mfem_gmres(lambda v: A @ v, None, b, xr, rel_tol=1e-10, max_iter=200, m=20)
    assert K.GetConverged() and K.GetNumIterations() == ref.final_iter and close(x, xr, 1e-9)


def test_stationary_solver_matches_reference_loop(la, h1_k1):
    A = h1_k1[0]
    dA, H, oprec = h1_setup(la, h1_k1)
    b = np.random.default_rng(4).uniform(-1, 1, A.shape[0])
    for kw in (dict(max_its=1), dict(max_its=4), dict(rel_tol=1e-6, max_its=100)):
        S = la.StationarySolver(dA, H, **kw)
        x = np.full_like(b, 3.0); xr = np.zeros_like(b)
        S.Mult(b, x)
        ref = orc.stationary_solver(lambda v: A @ v, oprec, b, xr, rel_tol=kw.get("rel_tol", 0.0), max_iter=kw["max_its"])
        assert (S.GetConverged(), S.GetNumIterations()) == (bool(ref.converged), ref.final_iter)
        assert close(x, xr, 1e-10) and np.isclose(S.GetFinalNorm(), ref.final_norm, rtol=1e-6)


# ---- BiCGSTAB / MINRES (KrylovSolver.cpp:39-42) ------------------------------------------------------------------------------

@pytest.mark.parametrize("prec", ["none", "vcycle"])
@pytest.mark.parametrize("check_every", [1, 8])
def test_bicgstab_matches_mfem(la, h1_k2, prec, check_every):
    A = h1_k2[0]
    dA, H, oprec = h1_setup(la, h1_k2)
    b = np.random.default_rng(4).uniform(-1, 1, A.shape[0])
    params = {"Solver name": "BiCGSTAB", "Relative tolerance": 1e-9, "Maximum iterations": 400, "Check every": check_every}
    K = la.KrylovSolver(dA, H if prec == "vcycle" else None, params)
    x = np.full(A.shape[0], -2.0)
    K.Mult(b, x)
    xr = np.zeros_like(b)
    ref = orc.mfem_bicgstab(lambda v: A @ v, oprec if prec == "vcycle" else None, b, xr, rel_tol=1e-9, max_iter=400)
    assert K.GetConverged() and ref.converged
    # the recurrences are run in the same order; BiCGSTAB amplifies the rounding of the dot products (different summation
    # order on the device), so long unpreconditioned runs drift by a few steps -- the short runs below are identical
    assert abs(K.GetNumIterations() - ref.final_iter) <= (0 if prec == "vcycle" else max(2, ref.final_iter // 8))
    assert np.linalg.norm(A @ x - b) <= 1e-8 * np.linalg.norm(b)
    if prec == "vcycle":
        assert close(x, xr, 1e-8) and np.isclose(K.GetFinalNorm(), ref.final_norm, rtol=1e-4)
    # maximum iterations: not converged, final_iter = max_iter, same iterate
    K2 = la.KrylovSolver(dA, H if prec == "vcycle" else None, dict(params, **{"Maximum iterations": 3, "Relative tolerance": 1e-14}))
    x3, xr3 = np.zeros_like(b), np.zeros_like(b)
    K2.Mult(b, x3)
    ref3 = orc.mfem_bicgstab(lambda v: A @ v, oprec if prec == "vcycle" else None, b, xr3, rel_tol=1e-14, max_iter=3)
    assert not K2.GetConverged() and K2.GetNumIterations() == ref3.final_iter == 3
    assert close(x3, xr3, 1e-10) and np.isclose(K2.GetFinalNorm(), ref3.final_norm, rtol=1e-8)


@pytest.mark.parametrize("prec", ["none", "l1-jacobi"])
def test_minres_on_the_darcy_system_matches_mfem(la, prec):
    """The symmetric indefinite system [[M, Bt], [B, 0]] of MultigridTestDarcy; the preconditioner must be SPD for MINRES:
    one l1-Jacobi sweep on the block-diagonal operator blockdiag(M, B diag(M)^-1 Bt)."""
    from parelag_amd import meshgen
    d = meshgen.darcy_problem(0, seq=meshgen.Sequence.kuhn_cube(2, 1))
    M, B, Bt = d["M"], d["B"], d["Bt"]
    A = sp.bmat([[M, Bt], [B, None]]).tocsr()
    n = A.shape[0]
    b = np.random.default_rng(6).uniform(-1, 1, n)
    dA = la.ParallelCSRMatrix.from_scipy(A)
    S = (B @ sp.diags(1.0 / M.diagonal()) @ Bt).tocsr()
    Pd = sp.block_diag([M, S]).tocsr()
    if prec == "none":
        sm, oprec = None, None
    else:
        dP = la.ParallelCSRMatrix.from_scipy(Pd)
        sm = la.HypreSmootherWrapper(dP, 1, {"Sweeps": 1})
        osm = orc.OSmoother(orc.OParCSR(orc.diag_first(Pd)), type=1, sweeps=1)

        def oprec(r, z):
            z[:] = 0.0
            osm.Mult(r, z)
    params = {"Solver name": "MINRES", "Relative tolerance": 1e-8, "Maximum iterations": 2000, "Check every": 8}
    K = la.KrylovSolver(dA, sm, params)
    x = np.zeros(n)
    K.Mult(b, x)
    xr = np.zeros(n)
    ref = orc.mfem_minres(lambda v: A @ v, oprec, b, xr, rel_tol=1e-8, max_iter=2000)
    assert K.GetConverged() and ref.converged
    assert abs(K.GetNumIterations() - ref.final_iter) <= 2 and ref.final_iter < 2000
    # (hundreds of Lanczos steps: the recurrences lose orthogonality at rounding level differently on the two sides; the
    # five-step run below is compared tightly)
    assert np.isclose(K.GetFinalNorm(), ref.final_norm, rtol=0.5)
    assert K.GetFinalNorm() <= 1e-8 * np.linalg.norm(b) * 1.0001
    assert close(x, xr, 1e-6)
    # a few steps only: identical recurrences, identical iterates
    K5 = la.KrylovSolver(dA, sm, dict(params, **{"Maximum iterations": 5, "Relative tolerance": 1e-14}))
    x5, xr5 = np.zeros(n), np.zeros(n)
    K5.Mult(b, x5)
    ref5 = orc.mfem_minres(lambda v: A @ v, oprec, b, xr5, rel_tol=1e-14, max_iter=5)
    assert not K5.GetConverged() and K5.GetNumIterations() == ref5.final_iter == 5
    assert close(x5, xr5, 1e-11) and np.isclose(K5.GetFinalNorm(), ref5.final_norm, rtol=1e-10)
    # iterative_mode: continue from the five-step iterate
    K.iterative_mode = True
    x6 = x5.copy(); xr6 = xr5.copy()
    K.Mult(b, x6)
    orc.mfem_minres(lambda v: A @ v, oprec, b, xr6, iterative_mode=True, rel_tol=1e-8, max_iter=2000)
    assert close(x6, xr6, 1e-6)


def test_krylov_solver_names(la, h1_k1):
    dA = la.ParallelCSRMatrix.from_scipy(h1_k1[0])
    for name in ("CG", "PCG", "GMRES", "FGMRES", "BiCGSTAB", "MINRES"):
        la.KrylovSolver(dA, None, {"Solver name": name})
    with pytest.raises(ValueError, match="Bad solver type"):
        la.KrylovSolver(dA, None, {"Solver name": "QMR"})


@pytest.mark.parametrize("name", ["PCG", "GMRES", "BiCGSTAB"])
def test_solver_adopts_the_numbering_of_a_reordered_hierarchy(la, h1_k2, name):
    """A hierarchy that renumbered its levels (automatic above 2^18 rows, forced here) hands the Krylov shell its level-0 operator
    and permutation: the iteration runs in that numbering -- right-hand side and initial guess permuted once on entry, the
    solution once on exit, no gather / scatter around the V-cycles -- and the caller sees the same iteration as with the
    natural numbering: same counts as the CPU port, same solution, in preconditioner and in iterative mode."""
    A, Ps, _ = h1_k2
    n = A.shape[0]
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi", reorder=1)
    order = H.GetOrdering(0)
    assert not np.array_equal(order, np.arange(n)) and np.array_equal(np.sort(order), np.arange(n))
    h = oracle_h1_hierarchy(A, Ps)
    Hc = orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"])
    oprec = lambda r, z: Hc.apply(r, z)
    rng = np.random.default_rng(8)
    b = rng.uniform(-1, 1, n)
    params = {"Solver name": name, "Relative tolerance": 1e-9, "Absolute tolerance": 0.0, "Maximum iterations": 100, "Restart size": 30}
    K = la.KrylovSolver(dA, H, params)
    x = np.full(n, 3.0)
    K.Mult(b, x)
    xr = np.zeros(n)
    if name == "PCG":
        ref = orc.mfem_cg(lambda v: A @ v, oprec, b, xr, rel_tol=1e-9, max_iter=100)
    elif name == "GMRES":
        ref = orc.mfem_gmres(lambda v: A @ v, oprec, b, xr, rel_tol=1e-9, abs_tol=0.0, max_iter=100, m=30)
    else:
        ref = orc.mfem_bicgstab(lambda v: A @ v, oprec, b, xr, rel_tol=1e-9, abs_tol=0.0, max_iter=100)
    assert K.GetConverged() and ref.converged and K.GetNumIterations() == ref.final_iter
    assert close(x, xr, 1e-8) and np.linalg.norm(A @ x - b) <= 1e-8 * np.linalg.norm(b)
    # iterative mode: the caller's initial guess goes through the permutation as well
    x0 = rng.uniform(-1, 1, n)
    xd = x0.copy()
    K.iterative_mode = True
    K.Mult(b, xd)
    assert K.GetConverged() and np.linalg.norm(A @ xd - b) <= 1e-8 * np.linalg.norm(b - A @ x0)
    # the standalone cycle (natural vectors in and out) is untouched
    y, yr = np.empty(n), np.empty(n)
    H.Mult(b, y); Hc.apply(b, yr)
    assert close(y, yr, 1e-11)
    assert H.time_apply_internal(3) > 0.0


def _krylov_with_env(la, env, *args):
    """a KrylovSolver made under `env` (the switches are read when the solver is made); the variables are put back afterwards"""
    import os
    saved_env = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return la.KrylovSolver(*args)
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_gmres_on_a_nearly_dependent_krylov_basis(la):
    """ADVICE r4 asked for this case when a two-vector Gram-Schmidt chain was the default (it is gone: krylov.hip's header): an operator that
    is an involution up to 1e-8 gives A v_1 = A^2 v_0 ~ v_0, so at step 1 the vector to orthogonalise is v_0 + O(1e-8) -- where classical
    Gram-Schmidt's error is eps |w| and modified Gram-Schmidt's eps |w - h v|.  The device's chain is MFEM's loop one vector at a time: it
    must agree with the oracle's mfem::GMRESSolver on the iteration count, the iterate and the true residual, in one launch per step and with
    a launch per link; the solver says how it issued its steps."""
    rng = np.random.default_rng(5)
    n = 600
    perm = np.arange(n).reshape(-1, 2)[:, ::-1].reshape(-1)            # swaps 2k <-> 2k+1: P^2 = I
    P = sp.csr_matrix((np.ones(n), (np.arange(n), perm)), shape=(n, n))
    S = sp.random(n, n, density=0.02, random_state=rng)
    for eps_op, tol in ((1e-8, 1e-12), (1e-5, 1e-12), (1e-2, 1e-10)):
        A = orc.csr(P + eps_op * S)
        dA = la.ParallelCSRMatrix.from_scipy(A)
        b = rng.uniform(-1, 1, n)
        out = []
        for env in ({"ELAG_HIP_GMRES_CHAIN": "1"}, {"ELAG_HIP_GMRES_CHAIN": "0"}):
            K = _krylov_with_env(la, env, dA, None, {"Solver name": "GMRES", "Relative tolerance": tol, "Maximum iterations": 300, "Restart size": 30})
            x = np.zeros(n)
            K.Mult(b, x)
            assert K.launch_mode().startswith(("graph", "plain"))
            out.append((K.GetConverged(), K.GetNumIterations(), x, K.GetFinalNorm()))
        xr = np.zeros(n)
        ref = orc.mfem_gmres(lambda v: A @ v, None, b, xr, rel_tol=tol, max_iter=300, m=30)
        assert ref.converged and ref.final_iter >= 2
        bn = np.linalg.norm(b)
        true_res = [np.linalg.norm(b - A @ o[2]) / bn for o in out] + [np.linalg.norm(b - A @ xr) / bn]
        print("eps", eps_op, "iterations", [o[1] for o in out], ref.final_iter, "true residuals", true_res)
        for (conv, it, x, fn), tr in zip(out, true_res):
            assert conv and it == ref.final_iter, (eps_op, it, ref.final_iter)
            assert tr <= 10 * true_res[2] + 1e-13 and close(x, xr, 1e-9), (eps_op, true_res)
        assert np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("name", ["GMRES", "FGMRES"])
def test_gmres_chain_in_one_launch_is_the_launch_per_link_bit_for_bit(la, name):
    """gmres_chain_kernel (krylov.hip): a step's whole modified Gram-Schmidt chain, its norm and the new basis vector in ONE launch -- the
    workgroups keep w in registers, the partial sums of every inner product go from workgroup to workgroup as tagged granules.  Same elements
    per virtual block, same order of summation, MFEM's loop one vector at a time: the iterate, the iteration count and the final norm are those
    of the launch-per-link kernels to the last bit -- with and without a preconditioner, across restarts, for both register forms, for sizes
    around the block and wavefront boundaries; and the oracle's mfem::GMRESSolver agrees as before."""
    rng = np.random.default_rng(23)
    for n, restart, with_prec in ((1, 4, False), (63, 5, False), (257, 7, True), (1000, 50, False), (40000, 12, True), (140001, 6, False)):
        A = orc.csr(sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng) + sp.diags(np.full(n, 2.5)))
        dA = la.ParallelCSRMatrix.from_scipy(A)
        b = rng.uniform(-1, 1, n)
        params0 = {"Solver name": name, "Relative tolerance": 1e-10, "Maximum iterations": 60, "Restart size": restart}
        out = []
        # ... and the step's rotations done inside that launch as the coefficients arrive (the default) or by gmres_givens_kernel
        # (ELAG_HIP_GMRES_CHAIN_GIVENS=0), the converged step's update launched once when the host has read `done` (the default) or by a
        # launch in every step (ELAG_HIP_GMRES_UPDATE_EACH_STEP=1): the same bits, whatever the batch the host polls in (check every 1 / 3 steps)
        for env in ({"ELAG_HIP_GMRES_CHAIN": "0"}, {"ELAG_HIP_GMRES_CHAIN": "1"}, {"ELAG_HIP_GMRES_CHAIN": "2"},
                    {"ELAG_HIP_GMRES_CHAIN": "1", "ELAG_HIP_GMRES_CHAIN_GIVENS": "0"}, {"ELAG_HIP_GMRES_CHAIN": "1", "ELAG_HIP_GMRES_UPDATE_EACH_STEP": "1"},
                    {"ELAG_HIP_GMRES_CHAIN": "0", "ELAG_HIP_GMRES_UPDATE_EACH_STEP": "1"}, {"ELAG_HIP_GMRES_CHAIN": "1", "check": 1},
                    {"ELAG_HIP_GMRES_CHAIN": "1", "check": 3}):
            env = dict(env)
            params = dict(params0, **({"Check every": env.pop("check")} if "check" in env else {}))
            S = la.HypreSmootherWrapper(dA, 1, {"Sweeps": 1}) if with_prec else None      # l1-Jacobi
            K = _krylov_with_env(la, env, dA, S, params)
            for trial in range(2):                                                          # (the second solve: a solver's launches are replayed)
                x = np.zeros(n)
                K.Mult(b, x)
            out.append((K.GetConverged(), K.GetNumIterations(), x, K.GetFinalNorm()))
        for o in out[1:]:
            assert o[0] == out[0][0] and o[1] == out[0][1] and o[3] == out[0][3], (n, restart, [(q[0], q[1], q[3]) for q in out])
            assert np.array_equal(o[2], out[0][2]), (n, restart, np.abs(o[2] - out[0][2]).max())
        if name == "GMRES" and not with_prec:
            xr = np.zeros(n)
            ref = orc.mfem_gmres(lambda v: A @ v, None, b, xr, rel_tol=1e-10, max_iter=60, m=restart)
            assert out[1][1] == ref.final_iter and close(out[1][2], xr, 1e-9)


def test_gmres_chain_that_gives_up_is_reported_and_the_solvers_go_back_to_a_launch_per_link(la):
    """Every wait of gmres_chain_kernel is bounded.  A test hook makes one workgroup publish its first partial sums ten time limits late
    (limit 0.2 ms): the others give up, the launch stores code 2 in the context's error word, the call (or the next one) fails with the reason, and from then on the solvers of the process take the launch
    per link -- same answer as before.  Runs in a child process: the fallback is for the life of the process."""
    import subprocess, sys, os
    code = r"""
import numpy as np, scipy.sparse as sp, os, sys
sys.path.insert(0, %r)
from parelag_amd import linalg as la
rng = np.random.default_rng(2)
n = 300000
A = (sp.random(n, n, density=4.0 / n, random_state=rng) + sp.diags(np.full(n, 2.5))).tocsr()
dA = la.ParallelCSRMatrix.from_scipy(A)
b = rng.uniform(-1, 1, n)
params = {"Solver name": "GMRES", "Relative tolerance": 1e-10, "Maximum iterations": 40, "Restart size": 10}
os.environ["ELAG_HIP_GMRES_CHAIN"] = "0"
K0 = la.KrylovSolver(dA, None, params)
x0 = np.zeros(n); K0.Mult(b, x0)
os.environ["ELAG_HIP_GMRES_CHAIN"] = "1"
os.environ["ELAG_HIP_GMRES_CHAIN_MS"] = "0.2"
os.environ["ELAG_HIP_GMRES_CHAIN_STALL"] = "7"
K = la.KrylovSolver(dA, None, params)
failed = ""
try:
    x = np.zeros(n); K.Mult(b, x)
    la.synchronize()
    K.GetNumIterations()
except Exception as e:
    failed = str(e)
print("FAILED:", failed)
del os.environ["ELAG_HIP_GMRES_CHAIN_MS"]
del os.environ["ELAG_HIP_GMRES_CHAIN_STALL"]
K2 = la.KrylovSolver(dA, None, params)
x2 = np.zeros(n); K2.Mult(b, x2)
print("SAME:", bool(np.array_equal(x2, x0)), K2.GetNumIterations() == K0.GetNumIterations(), K2.GetConverged())
x3 = np.zeros(n); K.Mult(b, x3)             # the solver whose chain gave up: its captured steps are dropped, a launch per link from here on
print("AGAIN:", bool(np.array_equal(x3, x0)), K.GetNumIterations() == K0.GetNumIterations(), K.GetConverged())
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = {ln.split(":", 1)[0]: ln.split(":", 1)[1].strip() for ln in r.stdout.splitlines() if ":" in ln}
    assert "Gram-Schmidt chain" in lines["FAILED"], r.stdout + r.stderr
    assert lines["SAME"] == "True True True", r.stdout + r.stderr
    assert lines["AGAIN"] == "True True True", r.stdout + r.stderr


def test_cg_with_the_update_in_the_launch_that_computes_alpha_is_the_same_iteration_bit_for_bit(la):
    """cg_fused_step2_update_kernel (krylov.hip): the update of x and r that mfem::CGSolver makes at the top of a pass is applied by the launch
    that computes its alpha at the end of the pass before (only den == 0 ends the loop between the two, and then nothing is applied): a launch
    a pass less, the same operations in the same order.  Iterate, iteration count, final norm and flags are those of the form with a launch of
    its own (ELAG_HIP_CG_TAIL_UPDATE=0) to the last bit: with and without a preconditioner, converging, running into the iteration limit,
    from an initial guess, on a zero right-hand side, around the block boundaries; and the oracle's mfem::CGSolver agrees as before."""
    rng = np.random.default_rng(31)
    for n, with_prec, max_iter, tol in ((1, False, 10, 1e-10), (255, True, 50, 1e-10), (1025, False, 7, 1e-14), (40000, True, 60, 1e-9), (140001, False, 30, 1e-8)):
        S0 = sp.random(n, n, density=min(1.0, 5.0 / n), random_state=rng)
        A = orc.csr(S0 + S0.T + sp.diags(np.full(n, 12.0)))
        dA = la.ParallelCSRMatrix.from_scipy(A)
        params = {"Solver name": "PCG", "Relative tolerance": tol, "Maximum iterations": max_iter}
        for b, x_init, iterative in ((rng.uniform(-1, 1, n), None, False), (rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), True), (np.zeros(n), None, False)):
            out = []
            for env in ({"ELAG_HIP_CG_TAIL_UPDATE": "0"}, {"ELAG_HIP_CG_TAIL_UPDATE": "1", "ELAG_HIP_CG_LEAN": "0"}):
                P = la.HypreSmootherWrapper(dA, 1, {"Sweeps": 1}) if with_prec else None      # l1-Jacobi
                K = _krylov_with_env(la, env, dA, P, params)
                K.iterative_mode = iterative
                for trial in range(2):                                                          # (the second solve replays the captured iteration)
                    x = np.zeros(n) if x_init is None else x_init.copy()
                    K.Mult(b, x)
                out.append((K.GetConverged(), K.GetNumIterations(), x, K.GetFinalNorm()))
            assert out[0][0] == out[1][0] and out[0][1] == out[1][1] and out[0][3] == out[1][3], (n, with_prec, [(q[0], q[1], q[3]) for q in out])
            assert np.array_equal(out[0][2], out[1][2]), (n, with_prec, np.abs(out[0][2] - out[1][2]).max())
        if not with_prec:
            b = rng.uniform(-1, 1, n)
            K = la.KrylovSolver(dA, None, params)
            x = np.zeros(n); K.Mult(b, x)
            xr = np.zeros(n)
            ref = orc.mfem_cg(lambda v: A @ v, None, b, xr, rel_tol=tol, max_iter=max_iter)
            assert K.GetNumIterations() == ref.final_iter and bool(K.GetConverged()) == bool(ref.converged) and close(x, xr, 1e-9)


def test_cg_pass_with_direction_product_and_inner_product_in_one_launch_is_the_same_iteration_bit_for_bit(la):
    """cg_dir_amul_dot_kernel (krylov.hip): on a launch-bound system (a row-group product, at most 131 072 unknowns) the direction update
    d' = z + beta d, the product A d' and the partial sums of (d', A d') are ONE launch -- the entries of d' are formed where the product needs
    them, by the lanes and in the order of spmv_fused_row, the partial sums come out of the blocks and the tree of gmres_dot0_kernel.  Iterate,
    iteration count, final norm and flags are those of the three launches (ELAG_HIP_CG_LEAN=0) to the last bit: rows of 1 to 40 entries (1 to
    16 lanes per row), sizes around the block boundaries and at the limit (beyond it the three launches run either way), with and without a
    preconditioner, from an initial guess, into the iteration limit, replayed as a graph and as plain launches."""
    rng = np.random.default_rng(77)
    cases = ((1, 1.0, False, 5, 1e-10), (255, 1.0, True, 40, 1e-10), (257, 3.0, False, 40, 1e-12), (1024, 9.0, True, 9, 1e-14), (5000, 20.0, False, 60, 1e-10),
             (42309, 7.0, True, 60, 1e-9), (131072, 2.0, False, 25, 1e-8), (131073, 2.0, False, 25, 1e-8))
    for n, half_row, with_prec, max_iter, tol in cases:
        S0 = sp.random(n, n, density=min(1.0, half_row / n), random_state=rng)
        A = orc.csr(S0 + S0.T + sp.diags(np.full(n, 4.0 * half_row + 2.0)))
        dA = la.ParallelCSRMatrix.from_scipy(A)
        params = {"Solver name": "PCG", "Relative tolerance": tol, "Maximum iterations": max_iter}
        for b, x_init, iterative in ((rng.uniform(-1, 1, n), None, False), (rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), True)):
            out = []
            for env in ({"ELAG_HIP_CG_LEAN": "0"}, {"ELAG_HIP_CG_LEAN": "1"}, {"ELAG_HIP_CG_LEAN": "1", "ELAG_HIP_KRYLOV_GRAPH": "0"}):
                env = dict(env)
                pr = dict(params, **({"Use graph": False} if env.pop("ELAG_HIP_KRYLOV_GRAPH", None) == "0" else {}))
                P = la.HypreSmootherWrapper(dA, 1, {"Sweeps": 1}) if with_prec else None      # l1-Jacobi
                K = _krylov_with_env(la, env, dA, P, pr)
                K.iterative_mode = iterative
                for trial in range(2):
                    x = np.zeros(n) if x_init is None else x_init.copy()
                    K.Mult(b, x)
                out.append((K.GetConverged(), K.GetNumIterations(), x, K.GetFinalNorm()))
            for o in out[1:]:
                assert o[0] == out[0][0] and o[1] == out[0][1] and o[3] == out[0][3], (n, half_row, with_prec, [(q[0], q[1], q[3]) for q in out])
                assert np.array_equal(o[2], out[0][2]), (n, half_row, with_prec, np.abs(o[2] - out[0][2]).max())
            if not iterative and not with_prec:
                xr = np.zeros(n)
                ref = orc.mfem_cg(lambda v: A @ v, None, b, xr, rel_tol=tol, max_iter=max_iter)
                assert out[1][1] == ref.final_iter and bool(out[1][0]) == bool(ref.converged) and close(out[1][2], xr, 1e-9)


def test_pcg_with_vcycle_in_the_lean_pass_is_the_three_launch_pass_bit_for_bit(la, h1_k2):
    """MultigridTest0Form's solve (PCG + the AMGe cycle) through cg_dir_amul_dot_kernel and through the three launches it replaces: the same
    iterate to the last bit, the same count, the same final norm."""
    A = h1_k2[0]
    dA, H, oprec = h1_setup(la, h1_k2)
    b = np.random.default_rng(4).uniform(-1, 1, A.shape[0])
    params = {"Solver name": "PCG", "Relative tolerance": 1e-10, "Absolute tolerance": 1e-14, "Maximum iterations": 100}
    out = []
    for env in ({"ELAG_HIP_CG_LEAN": "0"}, {"ELAG_HIP_CG_LEAN": "1"}):
        K = _krylov_with_env(la, env, dA, H, params)
        x = np.zeros(A.shape[0])
        K.Mult(b, x)
        out.append((K.GetConverged(), K.GetNumIterations(), x, K.GetFinalNorm()))
    assert out[0][0] and out[0][:2] == out[1][:2] and out[0][3] == out[1][3] and np.array_equal(out[0][2], out[1][2])
    xr = np.zeros_like(b)
    ref = orc.mfem_cg(lambda v: A @ v, oprec, b, xr, rel_tol=1e-10, abs_tol=1e-14, max_iter=100)
    assert out[1][1] == ref.final_iter and close(out[1][2], xr, 1e-9)
