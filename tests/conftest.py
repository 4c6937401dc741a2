import os
import sys

import pytest

# every rank thread of the in-process multi-rank tests owns a library context: a small first scratch slab each (the default reserves 2 GiB at init)
os.environ.setdefault("ELAG_HIP_ARENA_MB", "128")
# thread pools of the checker's host libraries (numpy's BLAS, the oracle's OpenMP build) at the size of the CPU quota, passive waits: a GPU box of
# the pool shows 256 CPUs and grants 16 cores' worth, and 256 spinning workers get the whole process throttled (bench.py tells the story)
def _granted_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n
for _var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, str(_granted_cores()))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def h1_k1():
    """0-form problem on cube456 refined once (840 dofs, 2 levels down to 141)."""
    from parelag_amd import meshgen
    return meshgen.h1_problem(1)


@pytest.fixture(scope="session")
def h1_k2():
    from parelag_amd import meshgen
    return meshgen.h1_problem(2)
