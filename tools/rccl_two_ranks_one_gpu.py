#!/usr/bin/env python3
"""Experiment: can two RCCL ranks share the single GPU of the test box (so that the real ncclSend/ncclRecv halo exchange
can be exercised there)?  Spawns two processes on device 0; prints the outcome.  Not part of the test suite."""
import ctypes as C, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
UID = "/tmp/elag_rccl_uid.bin"

def child(rank):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    from parelag_amd import _lib
    lib = _lib.init(0)
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        _lib.check(lib.elag_hip_comm_unique_id(buf))
        open(UID + ".tmp", "wb").write(bytes(buf)); os.replace(UID + ".tmp", UID)
    else:
        for _ in range(600):
            if os.path.exists(UID): break
            time.sleep(0.05)
        buf = (C.c_ubyte * 128)(*open(UID, "rb").read())
    _lib.check(lib.elag_hip_comm_init(rank, 2, buf))
    v = C.c_double(rank + 1.0)
    _lib.check(lib.elag_hip_comm_allreduce(C.byref(v), 0))
    print(f"rank {rank}: allreduce sum = {v.value}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1])); sys.exit(0)
    if os.path.exists(UID): os.remove(UID)
    ps = [subprocess.Popen([sys.executable, __file__, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    for r, p in enumerate(ps):
        try:
            out, _ = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            p.kill(); out = "TIMEOUT"
        print(f"--- rank {r} rc={p.returncode}\n{out[-1500:]}")
