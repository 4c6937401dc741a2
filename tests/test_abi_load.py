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


def test_header_is_plain_c_and_host_entry_points_run_without_a_device(tmp_path):
    """``include/elag_hip.h`` compiles as C99 with ``-pedantic`` and a C program can call the device-free entry points."""
    from parelag_amd import build
    lib = build.build()
    exe = str(tmp_path / "abi_header_test")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_header_test.c"), "-o", exe, lib, f"-Wl,-rpath,{os.path.dirname(lib)}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "C_ABI_OK" in out.stdout, out.stdout + out.stderr
