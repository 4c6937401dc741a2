#!/usr/bin/env python3
"""Where the host-side time of the driver's short timed region goes (bench.py --steps 20 --warmup 5).

For each launch mode (ELAG_HIP_GRAPH_MODE=graph | eager, set before the hierarchy is finalised) the headline hierarchy is
built and the driver's pattern is replayed several times -- drain, K x Hierarchy::Mult through the C ABI, drain -- with a host
timestamp after every call, so that the cost of the first launches after a drain, of the steady-state call and of the final
drain can be read separately.

    python tools/headline_probe.py [--steps 20] [--reps 8] [--refines 3]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--refines", type=int, default=3)
    ap.add_argument("--modes", default="graph,eager,auto")
    args = ap.parse_args()
    import numpy as np
    from parelag_amd import linalg as la, meshgen, _lib
    _lib.init()
    A, Ps, _ = meshgen.h1_problem(args.refines)
    dA = la.ParallelCSRMatrix.from_scipy(A)
    dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
    n = A.shape[0]
    for mode in args.modes.split(","):
        if mode == "auto":
            os.environ.pop("ELAG_HIP_GRAPH_MODE", None)
        else:
            os.environ["ELAG_HIP_GRAPH_MODE"] = mode
        t0 = time.perf_counter()
        H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi")
        la.synchronize()
        setup = time.perf_counter() - t0
        rhs, sol = la.DeviceVector(n, 2.0), la.DeviceVector(n, 1.0)
        for _ in range(5):
            H.Mult(rhs, sol)
        tw = time.perf_counter()
        while time.perf_counter() - tw < 0.3:
            for _ in range(200):
                H.Mult(rhs, sol)
            la.synchronize()
        print(f"== mode {mode}: launch_mode {H.launch_mode()} setup {setup*1e3:.1f} ms", flush=True)
        ms_dev = H.time_apply(rhs, sol, 200)
        print(f"   device (HIP events, 200 cycles): {ms_dev*1e3:.2f} us/cycle")
        for pre in ("none", "sleep", "sync_idle", "sync_idle_sleep0.3", "sync_idle_sleep1", "sync_idle_sleep5", "sync_idle_spin2", "burst1000_sleep5"):
            tot = []
            for rep in range(args.reps):
                if pre == "sleep":
                    time.sleep(0.005)
                elif pre == "events":
                    H.time_apply(rhs, sol, 200)
                elif pre == "print":
                    print("   .", flush=True)
                elif pre.startswith("sync_idle") or pre.startswith("burst"):
                    for _ in range(1000 if pre.startswith("burst") else 200):
                        H.Mult(rhs, sol)
                    la.synchronize()
                    if "sleep" in pre:
                        time.sleep(float(pre.split("sleep")[1]) * 1e-3)
                    elif "spin" in pre:
                        tq = time.perf_counter()
                        while time.perf_counter() - tq < 2e-3:
                            pass
                la.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    H.Mult(rhs, sol)
                t1 = time.perf_counter()
                la.synchronize()
                t2 = time.perf_counter()
                tot.append(((t2 - t0) * 1e6 / args.steps, (t1 - t0) * 1e6 / args.steps))
            print(f"   pre={pre}: per-cycle us (total, host issue): " + " ".join("%.1f/%.1f" % v for v in tot), flush=True)
        for K in (args.steps, 200, 2000):
            tot = []
            for rep in range(args.reps):
                la.synchronize()
                ts = np.empty(K + 2)
                ts[0] = time.perf_counter()
                for i in range(K):
                    H.Mult(rhs, sol)
                    ts[i + 1] = time.perf_counter()
                la.synchronize()
                ts[K + 1] = time.perf_counter()
                d = np.diff(ts) * 1e6
                tot.append((ts[-1] - ts[0]) * 1e6 / K)
                if rep < 3 and K == args.steps:
                    print(f"   K={K} rep {rep}: calls us " + " ".join("%.1f" % v for v in d[:K]) + f" | drain {d[K]:.1f} | per cycle {tot[-1]:.2f}")
            print(f"   K={K}: per-cycle us over {args.reps} reps: min {min(tot):.2f} median {sorted(tot)[len(tot)//2]:.2f} max {max(tot):.2f}", flush=True)
        del H


if __name__ == "__main__":
    main()
