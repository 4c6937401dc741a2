"""Builds libelag_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m parelag_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels with the
repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libelag_hip.so")
SOURCES = ["context.hip", "comm.hip", "kernels_spmv.hip", "kernels_setup.hip", "parcsr.hip", "dist_rap.hip", "maps.hip", "smoother.hip", "sweep.hip",
           "hierarchy.hip", "dense.hip", "blocked.hip", "krylov.hip", "amge.hip", "host_tables.hip", "host_amge_tables.hip"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wextra",
         "-Wno-unused-parameter", "-ffp-contract=off", "-I/opt/rocm/include"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "elag_hip.h"))
    return hdrs


def build(force: bool = False, verbose: bool = False) -> str:
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(h) for h in _deps())
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            jobs.append((src, obj))
    hipcc = _hipcc()

    def cc(job):
        cmd = [hipcc, *FLAGS, "-c", job[0], "-o", job[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {job[0]}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for warn in ex.map(cc, jobs):
            if warn and verbose:
                print(warn)
    objs = [os.path.join(objdir, s + ".o") for s in SOURCES]
    if jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
