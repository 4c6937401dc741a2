"""BASELINE configs[4] ("MultigridTestSPE10 heterogeneous-coeff Darcy, 8 MI355X, full RAP + AMGe hierarchy") on more than one rank:
the Cartesian grid of examples/MultigridTestSPE10.cpp:194-208 in row blocks of whole coarsest boxes
(partition.hex_element_owners), ParElag's own coarse spaces from the device coarsening, the blocked cycle of
distributed.build_darcy_hierarchy.  Ranks are host threads on the one GPU (in-process communicator)."""
import numpy as np
import pytest

from tests.helpers import oracle_blocked_hierarchy
from tests.test_gpu_multirank import run_ranks

pytestmark = pytest.mark.gpu


def _hex_problem(n, sigma, levels):
    from parelag_amd import derham
    from parelag_amd.amge_level import box_partition
    from parelag_amd.derham_level import seq_level_from_hex
    kinv = np.exp(sigma * np.random.default_rng(2025).standard_normal(n[0] * n[1] * n[2]))
    lv = seq_level_from_hex(n, h=(20.0, 10.0, 2.0), jstart=2, kinv=kinv)
    parts, m = [], n
    for _ in range(levels):
        parts.append(box_partition(m, (2, 2, 2)).astype(np.int64))
        m = tuple(-(-v // 2) for v in m)
    lvls, Ps = [lv], []
    for part in parts:
        res = derham.coarsen(lvls[-1], part)                       # what every rank runs on its share (derham_dist.coarsen_rank)
        Ps.append((res["P"][2].tocsr(), res["P"][3].tocsr()))
        lvls.append(res["coarse"])
    M, W = lv.mass(2), lv.mass(3)
    B = (W @ lv.D[2]).tocsr(); B.sort_indices()
    Bt = B.T.tocsr(); Bt.sort_indices()
    return dict(M=M, B=B, Bt=Bt, Ps=Ps, seq=None), lvls, parts


@pytest.mark.parametrize("world", [2, 3, 8])                       # (8: the rank count of BASELINE configs 4 / 5 and of the driver's scaling run)
@pytest.mark.parametrize("kind", ["Block Jacobi", "Block GS"])
def test_hex_darcy_cycle_on_row_blocks_matches_serial_oracle(world, kind):
    """--form spe10 --hex 16 16 16 --gpus N, in process: partition, distributed per-block Galerkin products, replicated blocked
    sub-hierarchy behind one all-reduce; the cycle equals the serial CPU oracle's on the same AMGe spaces"""
    from parelag_amd import distributed, linalg as la, partition
    n = (16, 16, 16)
    dar, lvls, parts = _hex_problem(n, 2.0, 2)
    assert lvls[1].ndof(2) > lvls[1].n[1] + 0                      # heterogeneous coefficient: extra (null-space / interior) coarse dofs
    eown = partition.hex_element_owners(n, parts, world)
    dd = partition.distributed_darcy_levels(dar, lvls, eown, world)
    n0, n1 = dar["M"].shape[0], dar["B"].shape[0]
    rng = np.random.default_rng(3)
    b0, b1 = rng.uniform(-1, 1, n0), rng.uniform(-1, 1, n1)

    def local(level, rank, v0, v1):
        (o2, o3), (s2, s3) = dd["order"][level], dd["starts"][level]
        return np.concatenate([v0[o2][s2[rank]:s2[rank + 1]], v1[o3][s3[rank]:s3[rank + 1]]])

    def body(dist):
        H, n_local, shapes, _ = distributed.build_darcy_hierarchy(dist, dar, kind, "L1 Jacobi", use_negative_s=False, dd=dd)
        rhs = la.DeviceVector.from_numpy(local(0, dist.rank, b0, b1))
        sol = la.DeviceVector(n_local, 7.0)
        H.Mult(rhs, sol); H.Mult(rhs, sol)
        la.synchronize()
        x = sol.to_numpy()
        assert shapes[0][0] == n0 + n1 and len(shapes) == len(dar["Ps"]) + 1
        del H, rhs, sol
        return x
    res = run_ranks(world, body)
    Ho, _ = oracle_blocked_hierarchy({(0, 0): dar["M"], (0, 1): dar["Bt"], (1, 0): dar["B"]}, dar["Ps"],
                                     kind="Block GS Lower" if kind == "Block GS" else kind, use_negative_s=False)
    xr = np.zeros(n0 + n1)
    Ho.Mult(np.concatenate([b0, b1]), xr)
    for r, x in enumerate(res):
        assert np.max(np.abs(x - local(0, r, xr[:n0], xr[n0:]))) <= 1e-10 * np.max(np.abs(xr))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_hex_darcy_default_solver_graph_converges_on_row_blocks(world):
    """configs[4] with the reference's SPE10 solver graph (spe10_example_parameters.xml:95-127) on N ranks: GMRES preconditioned by
    Block LDU whose inverses are distributed AMGe V-cycles on M and on B diag(M)^-1 B^T (distributed.build_darcy_ldu_amge).  The solve
    converges, every rank agrees on the iteration count, the assembled solution satisfies the global system, and the count is the
    single-rank solver's (same spaces, same graph) give or take the rounding of the distributed sums."""
    from parelag_amd import distributed, linalg as la, partition
    import scipy.sparse as sp
    n = (16, 16, 16)
    dar, lvls, parts = _hex_problem(n, 2.0, 2)
    eown = partition.hex_element_owners(n, parts, world)
    dd = partition.distributed_darcy_levels(dar, lvls, eown, world)
    n0, n1 = dar["M"].shape[0], dar["B"].shape[0]
    b0, b1 = np.zeros(n0), np.asarray(lvls[0].mass(3) @ lvls[0].const3).ravel()          # f = 1 in the pressure equation (MultigridTestSPE10.cpp)
    kparams = {"Solver name": "GMRES", "Relative tolerance": 1e-6, "Absolute tolerance": 1e-6, "Maximum iterations": 300, "Restart size": 50}

    def local(rank, v0, v1):
        (o2, o3), (s2, s3) = dd["order"][0], dd["starts"][0]
        return np.concatenate([v0[o2][s2[rank]:s2[rank + 1]], v1[o3][s3[rank]:s3[rank + 1]]])

    def body(dist):
        H, n_local, _shapes, _ = distributed.build_darcy_hierarchy(dist, dar, "Block Jacobi", "L1 Jacobi", use_negative_s=False, dd=dd)
        ldu = distributed.build_darcy_ldu_amge(dist, dar, H, "L1 Jacobi")
        K = la.KrylovSolver(H._block_op, ldu, kparams)
        rhs = la.DeviceVector.from_numpy(local(dist.rank, b0, b1))
        x = la.DeviceVector(n_local, 0.0)
        K.Mult(rhs, x)
        la.synchronize()
        out = (x.to_numpy(), K.GetNumIterations(), K.GetConverged())
        del K, ldu, H, rhs, x
        return out
    res = run_ranks(world, body)
    its = {r[1] for r in res}
    assert all(r[2] for r in res) and len(its) == 1, [(r[1], r[2]) for r in res]
    # the global solution from the ranks' pieces, against the global operator
    A = sp.bmat([[dar["M"], dar["Bt"]], [dar["B"], None]]).tocsr()
    x0, x1 = np.zeros(n0), np.zeros(n1)
    (o2, o3), (s2, s3) = dd["order"][0], dd["starts"][0]
    for r, (x, _it, _c) in enumerate(res):
        m0 = int(s2[r + 1] - s2[r])
        x0[o2[s2[r]:s2[r + 1]]] = x[:m0]
        x1[o3[s3[r]:s3[r + 1]]] = x[m0:]
    b = np.concatenate([b0, b1])
    # (MFEM's GMRES stops on the PRECONDITIONED residual: the true one of a converged solve is a few 1e-5 here)
    assert np.linalg.norm(A @ np.concatenate([x0, x1]) - b) <= 5e-4 * np.linalg.norm(b)
    assert its.pop() <= 100


def test_bench_runs_config_5_through_the_multi_rank_path():
    """``bench.py --form spe10 --hex 16 16 16`` through the N-rank branch (sharded coarsening, row-block hierarchy) with one rank:
    everything but the second process; parity of the timed cycle vs the serial oracle is part of the line"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--form", "spe10", "--hex", "16", "16", "16", "--amge-levels", "2",
                        "--force-dist", "--steps", "20", "--warmup", "3", "--no-roofline", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["parity_rel_err_vs_oracle"] <= 1e-10
    assert "hex grid 16x16x16" in rec["config"]["workload"] and "sharded over 1 rank" in rec["amge"]["what"]
