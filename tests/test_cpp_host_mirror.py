"""The header-only C++ mirror of the reference interface compiles against the C ABI (CPU) and runs (GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "build", "host_mirror_test")


def compile_test():
    from parelag_amd import build
    lib = build.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", SRC, "-o", EXE, lib, f"-Wl,-rpath,{os.path.dirname(lib)}",
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_host_mirror_compiles_with_plain_gxx():
    """No HIP headers are needed on the host side of the boundary: the mirror is plain C++ over a C ABI."""
    compile_test()


@pytest.mark.gpu
def test_host_mirror_runs():
    exe = compile_test()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "HOST_MIRROR_OK" in r.stdout
