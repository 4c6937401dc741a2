import os
import sys

import pytest

# every rank thread of the in-process multi-rank tests owns a library context: a small first scratch slab each (the default reserves 2 GiB at init)
os.environ.setdefault("ELAG_HIP_ARENA_MB", "128")

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
