"""CPU-side checks of the drop-in boundary: the library loads and exports exactly the symbols that
include/elag_hip.h declares (no compute calls -- there is no GPU here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "elag_hip.h")).read()
    return sorted(set(re.findall(r"ELAG_HIP_API\s+[\w\s\*]*?\b(elag_hip_\w+)\s*\(", txt)))


def test_header_declares_what_python_binds():
    from parelag_amd import _lib
    assert sorted(_lib.ABI_SYMBOLS) == header_symbols()


def test_library_exports_every_declared_symbol():
    from parelag_amd import _lib, build
    path = build.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T elag_hip_" in l)
    assert exported == header_symbols()
    lib = _lib.load(path)          # dlopen + prototypes; does not touch a GPU
    for name in header_symbols():
        assert hasattr(lib, name)


def test_product_path_fails_loudly_without_device():
    """No CPU fallback: on a box without a GPU init must raise, not silently compute elsewhere."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    from parelag_amd import _lib
    lib = _lib.load()
    rc = lib.elag_hip_init(0)
    assert rc != 0
    assert b"device" in lib.elag_hip_last_error().lower()
    from parelag_amd import linalg
    import numpy as np, scipy.sparse as sp
    with pytest.raises(_lib.ElagHipError):
        linalg.ParallelCSRMatrix.from_scipy(sp.eye(3, format="csr"))
