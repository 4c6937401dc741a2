#!/usr/bin/env python3
"""CPU baseline of bench.py in the reference's own layout (one process-rank per core, no OpenMP -- CMakeLists.txt:130-132):
the oracle's V-cycle of MultigridTest0Form on cube456 x<k> run by forked worker processes that own row blocks and share their vectors
(oracle/elag_oracle.c, orc_hier_spmd_bench), for a sweep of worker counts; also the oracle's setup (Galerkin products, l1 norms,
coarse factorisation) on one core.  A program of its own because it forks: bench.py starts it as a child process, which holds no
GPU context.  Test infrastructure (oracle/): it times the checker, never the product.

    python -m oracle.spmd_baseline --refines 3 --seconds 3 [--workers 1,2,4,8,16]
prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """cores this process may use: its affinity mask, cut down to the cgroup's CPU quota where one is set (a GPU box of this pool shows
    all 256 cores of the host in the mask and grants 16 of them)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p_ = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p_))
        except (OSError, ValueError):
            pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--refines", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=3.0, help="per worker count")
    ap.add_argument("--workers", default=None, help="comma list; default: powers of two up to the cores this process may use")
    ap.add_argument("--smoother", type=int, default=1, choices=[1, 2], help="hypre relax type of every level: 1 = l1-Jacobi; 2 = \"L1 Gauss-Seidel\", the "
                    "reference's shipped smoother -- on W processes hybrid Gauss-Seidel on W ranks (each sweeps its own rows, the others frozen)")
    args = ap.parse_args()
    import numpy as np
    from parelag_amd import meshgen
    from oracle import oracle as orc
    from tests.helpers import oracle_h1_hierarchy
    avail = usable_cores()
    workers = [int(w) for w in args.workers.split(",")] if args.workers else [w for w in (1, 2, 4, 8, 16, 32, 64) if w <= min(avail, 64)]
    t0 = time.perf_counter()
    A, Ps, _ = meshgen.h1_problem(args.refines)
    gen_s = time.perf_counter() - t0
    orc.build(native=True)
    t0 = time.perf_counter()
    h = oracle_h1_hierarchy(A, Ps, smoother=args.smoother)
    setup_s = time.perf_counter() - t0                      # RAP chain + FixZeroRows + l1 norms + dense Cholesky, one core
    H = orc.CHierarchy(h["As"], h["Ps"], h["sms"], h["chol"], native=True)
    b = np.full(A.shape[0], 2.0)
    xs = np.empty_like(b)
    H.apply(b, xs)
    sweep, diff = {}, 0.0
    for w in workers:
        rate, x = H.spmd_bench(b, w, args.seconds)
        sweep[w] = round(rate, 3)
        if args.smoother == 1 or w == 1:                     # (hybrid Gauss-Seidel on W > 1 ranks is another cycle than the one-rank one, by definition)
            diff = max(diff, float(np.max(np.abs(x - xs)) / np.max(np.abs(xs))))
    best = max(sweep, key=sweep.get)
    print(json.dumps({"refines": args.refines, "n": int(A.shape[0]), "nnz": int(A.nnz), "levels": len(Ps) + 1, "relax_type": args.smoother,
                      "vcycles_per_s_by_workers": sweep, "best_workers": best, "best_vcycles_per_s": sweep[best],
                      "rel_diff_vs_serial_cycle": diff, "setup_s_1core": round(setup_s, 4), "generate_s": round(gen_s, 2),
                      "cores_available": avail, "cpu_model": cpu_model(), "seconds_per_point": args.seconds}), flush=True)


if __name__ == "__main__":
    main()
