"""Worker of tests/test_distributed_cpu.py: run with torch.distributed.run, world_size >= 2, gloo backend.
Every rank builds its rows of the partitioned cube456 hierarchy and drives the ORACLE's ParCSR kernels with a
torch.distributed halo exchange; results are compared with the serial oracle on the global problem."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch.distributed as td

from oracle import oracle as orc
from parelag_amd import meshgen, partition
from tests.helpers import oracle_h1_hierarchy


def main():
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    ex = orc.TorchDistExchange(td)
    A, Ps, seq = meshgen.h1_problem(2)
    d = partition.distributed_h1_hierarchy(A, Ps, seq, world)
    nl = len(d["levels"])
    rng = np.random.default_rng(3)

    # --- operator products against the global matrices
    for l in range(nl):
        lev = d["levels"][l]
        Ag = d["A_global"][l]
        M = orc.oparcsr_from_part(lev["A_parts"][rank], ex)
        s = lev["starts"]
        x = rng.standard_normal(Ag.shape[0])
        y = np.zeros(M.shape[0])
        M.matvec(1.0, x[s[rank]:s[rank + 1]], 0.0, y)
        ref = (Ag @ x)[s[rank]:s[rank + 1]]
        assert np.allclose(y, ref, rtol=1e-12, atol=1e-13), f"matvec level {l}"
        assert abs(M.norm_linf() - abs(Ag).sum(axis=1).max()) < 1e-12
        assert abs(M.norm_l1() - abs(Ag).sum(axis=0).max()) < 1e-12
        if l > 0:
            Pg = d["P_global"][l - 1]
            P = orc.oparcsr_from_part(lev["P_parts"][rank], ex)
            sf = d["levels"][l - 1]["starts"]
            xf = rng.standard_normal(Pg.shape[0])
            yc = np.zeros(P.shape[1])
            P.matvecT(1.0, xf[sf[rank]:sf[rank + 1]], 0.0, yc)
            assert np.allclose(yc, (Pg.T @ xf)[s[rank]:s[rank + 1]], rtol=1e-12, atol=1e-13), f"matvecT level {l}"

    # --- distributed V-cycle (l1-Jacobi, replicated dense coarse solve) == serial oracle V-cycle
    levels = []
    Lc = orc.chol_factor(d["A_global"][-1].toarray())
    cstarts = d["levels"][-1]["starts"]

    def coarse(rhs, sol):
        full = ex.allgatherv(rhs, cstarts)
        xs = np.empty_like(full)
        orc.chol_solve(Lc, full, xs)
        sol[:] = xs[cstarts[rank]:cstarts[rank + 1]]
    for l in range(nl):
        lev = d["levels"][l]
        Al = orc.oparcsr_from_part(lev["A_parts"][rank], ex)
        entry = dict(A=Al, P=orc.oparcsr_from_part(lev["P_parts"][rank], ex) if l > 0 else None)
        if l < nl - 1:
            entry["pre"] = orc.OSmoother(Al, type=1)
            entry["post"] = orc.OSmoother(Al, type=1)
        else:
            entry["coarse"] = coarse
        levels.append(entry)
    H = orc.OHierarchy(levels)
    s0 = d["levels"][0]["starts"]
    b_nat = rng.uniform(-1, 1, A.shape[0])
    order0 = d["levels"][0]["order"]
    b = b_nat[order0]                                    # block numbering
    x_loc = np.empty(s0[rank + 1] - s0[rank])
    H.Mult(b[s0[rank]:s0[rank + 1]], x_loc)
    h = oracle_h1_hierarchy(A, Ps)
    xs = np.empty_like(b_nat)
    orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"]).apply(b_nat, xs)
    ref = xs[order0][s0[rank]:s0[rank + 1]]
    err = np.max(np.abs(x_loc - ref)) / np.max(np.abs(xs))
    assert err < 1e-11, f"distributed V-cycle differs from the serial one: {err}"

    # --- the layout bench.py uses across GPUs: level 0 distributed, levels >= 1 replicated on every rank; the restriction
    #     produces per-rank partial sums (P with global coarse columns) completed by ONE all-reduce, the prolongation
    #     needs no exchange
    import scipy.sparse as sp
    import torch
    Pg0 = sp.csr_matrix(d["P_global"][0])
    Ploc = orc.csr(Pg0[s0[rank]:s0[rank + 1]])
    hsub = oracle_h1_hierarchy(d["A_global"][1], d["P_global"][1:])
    Hsub = orc.CHierarchy(hsub["As"], hsub["Ps"], hsub["sms"], hsub["chol"])

    def replicated_coarse(rhs_partial, sol):
        t = torch.from_numpy(np.ascontiguousarray(rhs_partial).copy())
        td.all_reduce(t)                       # sum of the ranks' partial restricted residuals
        Hsub.apply(t.numpy(), sol)
    A0 = levels[0]["A"]
    lev2 = [dict(A=A0, P=None, pre=orc.OSmoother(A0, type=1), post=orc.OSmoother(A0, type=1)),
            dict(A=orc.OParCSR(orc.csr(sp.csr_matrix((Pg0.shape[1], Pg0.shape[1])))), P=orc.OParCSR(Ploc), coarse=replicated_coarse)]
    H2 = orc.OHierarchy(lev2)
    x2 = np.empty_like(x_loc)
    H2.Mult(b[s0[rank]:s0[rank + 1]], x2)
    err2 = np.max(np.abs(x2 - ref)) / np.max(np.abs(xs))
    assert err2 < 1e-11, f"replicated-coarse layout differs from the serial cycle: {err2}"

    # Chebyshev eigen-estimate runs through the distributed dot products
    M0 = levels[0]["A"]
    mx, mn = orc.max_eig_estimate_cg(M0, 1, 10)
    assert 0.5 < mx < 2.5 and mn > 0

    # --- the sharded generator over gloo: every rank makes only its rows, and they are the serial problem's row blocks with the
    #     comm package split_rows computes from the global matrix; the oracle's ParCSR product on the shard is the global product
    from parelag_amd import meshgen_sharded as msh
    comm = msh.TorchComm(td, rank, world)
    sh = msh.sharded_h1_problem(comm, 2)
    for l in range(nl):
        assert np.array_equal(sh["starts"][l], d["levels"][l]["starts"]), f"sharded starts level {l}"
    mine = partition.canon(sh["A_rows"])
    want = partition.canon(d["A_global"][0][s0[rank]:s0[rank + 1]])
    assert np.array_equal(mine.indptr, want.indptr) and np.array_equal(mine.indices, want.indices)
    assert np.max(np.abs(mine.data - want.data)) <= 1e-13 * np.max(np.abs(want.data))
    for l in range(nl - 1):
        sf = d["levels"][l]["starts"]
        got, ref_p = partition.canon(sh["P_rows"][l]), partition.canon(d["P_global"][l][sf[rank]:sf[rank + 1]])
        assert np.array_equal(got.indptr, ref_p.indptr) and np.array_equal(got.indices, ref_p.indices) and np.array_equal(got.data, ref_p.data)
    piece = msh.parcsr_piece(comm, sh["A_rows"], sh["starts"][0], sh["starts"][0])
    Msh = orc.oparcsr_from_part(piece, ex)
    xg = rng.standard_normal(A.shape[0])
    ysh = np.zeros(Msh.shape[0])
    Msh.matvec(1.0, xg[s0[rank]:s0[rank + 1]], 0.0, ysh)
    assert np.allclose(ysh, (d["A_global"][0] @ xg)[s0[rank]:s0[rank + 1]], rtol=1e-12, atol=1e-13), "sharded operator product"
    td.barrier()
    if rank == 0:
        print("DIST_WORKER_OK", world)


if __name__ == "__main__":
    main()
