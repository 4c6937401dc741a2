"""The threaded host table code (csrc/host_tables.hip, csrc/host_amge_tables.hip) under ThreadSanitizer and under
AddressSanitizer + UBSan: host-only builds of the two files with a driver (tests/cpp/host_tables_threads.cpp) that runs every threaded
entry point with one thread and with six and compares the tables.  (Sanitizers run on the CPU build only; no device code is involved.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "parelag_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "cpp", "build")


@pytest.mark.parametrize("name,flags,env", [
    ("tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"}),
    ("asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], {"ASAN_OPTIONS": "detect_leaks=0 exitcode=67"}),
])
def test_host_tables_under_sanitizers(name, flags, env):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "host_tables_threads_" + name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-host-only", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", *flags, "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, os.path.join(CSRC, "host_tables.hip"), os.path.join(CSRC, "host_amge_tables.hip"),
           "-x", "c++", os.path.join(ROOT, "tests", "cpp", "host_tables_threads.cpp"), "-o", exe, "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "HOST_TABLES_THREADS_OK" in r.stdout, (r.stdout + r.stderr)[-4000:]
