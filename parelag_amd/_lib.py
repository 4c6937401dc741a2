"""ctypes binding of libelag_hip.so (the C ABI declared in include/elag_hip.h).

The product path has no CPU fallback: if the HIP library is missing or the device cannot be initialised
the import / init raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libelag_hip.so")

HOST, DEVICE = 0, 1
SMOOTHER_JACOBI, SMOOTHER_L1_JACOBI, SMOOTHER_CHEBYSHEV = 0, 1, 16

_i = C.c_int
_d = C.c_double
_p = C.c_void_p
_i64 = C.c_int64
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_lp = C.POINTER(C.c_int64)


class ParCSRDesc(C.Structure):
    _fields_ = [
        ("num_rows", _i), ("num_cols_diag", _i),
        ("diag_i", _p), ("diag_j", _p), ("diag_a", _p),
        ("num_cols_offd", _i),
        ("offd_i", _p), ("offd_j", _p), ("offd_a", _p),
        ("col_map_offd", _p),
        ("first_row", _i64), ("first_col_diag", _i64), ("global_num_rows", _i64), ("global_num_cols", _i64),
        ("num_sends", _i),
        ("send_procs", _p), ("send_map_starts", _p), ("send_map_elmts", _p),
        ("num_recvs", _i),
        ("recv_procs", _p), ("recv_vec_starts", _p),
    ]


class ParCSRInfo(C.Structure):
    _fields_ = [
        ("num_rows", _i), ("num_cols_diag", _i), ("num_cols_offd", _i),
        ("nnz_diag", _i64), ("nnz_offd", _i64), ("first_row", _i64), ("first_col_diag", _i64),
        ("global_num_rows", _i64), ("global_num_cols", _i64),
        ("num_sends", _i), ("num_recvs", _i), ("send_total", _i),
    ]


class AMGeOptions(C.Structure):
    _fields_ = [("fix_zero_rows", _i), ("relax_type", _i), ("sweeps", _i), ("damping", _d), ("cheby_order", _i),
                ("cheby_fraction", _d), ("coarse_kind", _i), ("coarse_iters", _i), ("use_graph", _i), ("reorder", _i)]


class BlockedOptions(C.Structure):
    _fields_ = [("block_smoother", _i), ("inv_a00", AMGeOptions), ("inv_a11", AMGeOptions), ("s_type", _i), ("alpha", _d),
                ("use_negative_s", _i), ("damping", _d), ("coarse_kind", _i), ("use_graph", _i)]


class KrylovOptions(C.Structure):
    _fields_ = [("solver", _i), ("max_iter", _i), ("rel_tol", _d), ("abs_tol", _d), ("restart", _i), ("check_every", _i),
                ("use_graph", _i)]


# name -> argtypes (restype is int status unless listed in _RESTYPES)
_SIGNATURES = {
    "elag_hip_init": [_i],
    "elag_hip_finalize": [],
    "elag_hip_synchronize": [],
    "elag_hip_trim_memory": [C.POINTER(C.c_size_t)],
    "elag_hip_last_error": [],
    "elag_hip_device_name": [C.c_char_p, _i],
    "elag_hip_comm_unique_id": [_p],
    "elag_hip_comm_init": [_i, _i, _p],
    "elag_hip_comm_init_inprocess": [_i, _i, _i],
    "elag_hip_comm_selftest": [_dp, _dp],
    "elag_hip_comm_rank": [_ip, _ip],
    "elag_hip_comm_rounds": [_p, _p],
    "elag_hip_comm_allreduce": [_dp, _i],
    "elag_hip_malloc": [C.POINTER(_p), C.c_size_t],
    "elag_hip_free": [_p],
    "elag_hip_memcpy": [_p, _p, C.c_size_t, _i],
    "elag_hip_memset": [_p, _i, C.c_size_t],
    "elag_hip_parcsr_create": [C.POINTER(_p), C.POINTER(ParCSRDesc)],
    "elag_hip_parcsr_create_from_global_csr": [C.POINTER(_p), _i, _p, _p, _p, C.c_int64, C.c_int64, C.c_int64, _i, C.c_int64],
    "elag_hip_parcsr_destroy": [_p],
    "elag_hip_parcsr_get_info": [_p, C.POINTER(ParCSRInfo)],
    "elag_hip_parcsr_export": [_p, _i, _p, _p, _p, _p, _p, _p, _p],
    "elag_hip_parcsr_spmv": [_p, _i, _d, _p, _d, _p, _i],
    "elag_hip_parcsr_residual": [_p, _p, _p, _p, _i],
    "elag_hip_parcsr_transpose": [_p, C.POINTER(_p)],
    "elag_hip_parcsr_matmul": [_p, _p, C.POINTER(_p)],
    "elag_hip_parcsr_rap": [_p, _p, _p, C.POINTER(_p)],
    "elag_hip_parcsr_replicate": [_p, C.POINTER(_p), _p],
    "elag_hip_parcsr_rap_ignore_nonlocal_range": [_p, _p, _p, C.POINTER(_p)],
    "elag_hip_parcsr_spmvT_ignore_nonlocal": [_p, _p, _p, _i],
    "elag_hip_parcsr_add": [_d, _p, _d, _p, C.POINTER(_p)],
    "elag_hip_parcsr_rdp": [_p, _p, _i, _p, C.POINTER(_p)],
    "elag_hip_parcsr_fix_zero_rows": [_p, _ip],
    "elag_hip_parcsr_delete_zeros": [_p, _d],
    "elag_hip_parcsr_sign_transform": [_p],
    "elag_hip_parcsr_diagonal": [C.POINTER(_p), _i, _i64, _i64, _p],
    "elag_hip_parcsr_norm": [_p, _i, _dp],
    "elag_hip_parcsr_compare": [_p, _p, _d, _ip],
    "elag_hip_parcsr_get_diagonal": [_p, _p, _i],
    "elag_hip_parcsr_set_spmv_lanes": [_p, _i],
    "elag_hip_parcsr_get_spmv_form": [_p, _ip, _ip],
    "elag_hip_parcsr_spmv_boolint": [_p, _i, _p, _i, _p, _i],
    "elag_hip_parcsr_keep_equal_entries": [_p, _p],
    "elag_hip_partition_refined": [_i, _i, _i, _i, _p],
    "elag_hip_connected_components": [_i, _p, _p, _p, _ip],
    "elag_hip_transpose_partitioning": [_i, _p, _i, _p, _p],
    "elag_hip_smoother_create": [C.POINTER(_p), _p, _i, _i, _d, _d, _i, _d],
    "elag_hip_smoother_get_eig_estimate": [_p, _dp, _dp],
    "elag_hip_smoother_set_eig_estimate": [_p, _d, _d],
    "elag_hip_set_gs_hybrid_blocks": [_i],
    "elag_hip_hiptmair_create": [C.POINTER(_p), _p, _p, _p, _p],
    "elag_hip_smoother_apply": [_p, _p, _p, _i, _i, _i],
    "elag_hip_smoother_destroy": [_p],
    "elag_hip_coarse_create": [C.POINTER(_p), _p, _i, _i],
    "elag_hip_coarse_create_replicated": [C.POINTER(_p), _p, _p, C.c_int],
    "elag_hip_coarse_create_from_hierarchy": [C.POINTER(_p), _p, _i],
    "elag_hip_coarse_apply": [_p, _p, _p, _i],
    "elag_hip_coarse_destroy": [_p],
    "elag_hip_hierarchy_create": [C.POINTER(_p), _i],
    "elag_hip_hierarchy_set_operator": [_p, _i, _p],
    "elag_hip_hierarchy_set_prolongator": [_p, _i, _p],
    "elag_hip_hierarchy_set_smoothers": [_p, _i, _p, _p],
    "elag_hip_hierarchy_set_coarse_solver": [_p, _p],
    "elag_hip_hierarchy_finalize": [_p, _i],
    "elag_hip_hierarchy_launch_mode": [_p, _p, _p, _p, _p],
    "elag_hip_hierarchy_setup_breakdown": [_p, C.c_char_p, _i],
    "elag_hip_hierarchy_apply": [_p, _p, _p, _i, _i],
    "elag_hip_hierarchy_destroy": [_p],
    "elag_hip_amge_default_options": [C.POINTER(AMGeOptions)],
    "elag_hip_hierarchy_build": [C.POINTER(_p), _p, _i, C.POINTER(_p), C.POINTER(AMGeOptions)],
    "elag_hip_hierarchy_build_hiptmair": [C.POINTER(_p), _p, _i, C.POINTER(_p), C.POINTER(_p), C.POINTER(AMGeOptions),
                                          C.POINTER(AMGeOptions)],
    "elag_hip_hierarchy_get_ordering": [_p, _i, _p],
    "elag_hip_hierarchy_get_operator": [_p, _i, C.POINTER(_p)],
    "elag_hip_smoother_from_hierarchy": [C.POINTER(_p), _p, _d],
    "elag_hip_hierarchy_get_smoother": [_p, _i, C.POINTER(_p)],
    "elag_hip_blockop_create": [C.POINTER(_p), _i, _i, _p, _p],
    "elag_hip_blockop_set_block": [_p, _i, _i, _p],
    "elag_hip_blockop_get_block": [_p, _i, _i, C.POINTER(_p)],
    "elag_hip_blockop_monolithic": [_p, C.POINTER(_p)],
    "elag_hip_blockop_mult": [_p, _i, _d, _p, _d, _p, _i],
    "elag_hip_blockop_destroy": [_p],
    "elag_hip_schur_complement": [_p, _i, _d, _i, C.POINTER(_p)],
    "elag_hip_block_smoother_create": [C.POINTER(_p), _p, _i, _i, C.POINTER(_p), _d],
    "elag_hip_blocked_default_options": [C.POINTER(BlockedOptions)],
    "elag_hip_hierarchy_build_blocked": [C.POINTER(_p), _p, _i, C.POINTER(_p), _p, C.POINTER(BlockedOptions)],
    "elag_hip_krylov_default_options": [C.POINTER(KrylovOptions)],
    "elag_hip_krylov_create": [C.POINTER(_p), _p, _i, _p, C.POINTER(KrylovOptions)],
    "elag_hip_krylov_mult": [_p, _p, _p, _i, _i],
    "elag_hip_krylov_get_stats": [_p, _ip, _ip, _dp],
    "elag_hip_krylov_launch_mode": [_p],
    "elag_hip_krylov_destroy": [_p],
    "elag_hip_batched_rank": [_i, _p, _p, _p, _p, _p, _p, _d, _p],
    # host integer tables of the coarsening (no device)
    "elag_hip_host_table_product": [_i64, _p, _p, _p, _i64, _p, _p, _p, _i64, _p, _p, _p, _p],
    "elag_hip_host_lookup_find": [_i64, _p, _p, _i64, _i64, _p, _p, _p],
    "elag_hip_host_gather_dense": [_i64, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p],
    "elag_hip_host_expand_ranges": [_i64, _p, _p, _i64, _p, _p],
    "elag_hip_host_entity_dof_table": [_i64, _i, _p, _p, _p, _p, _p, _i64, _p, _p],
    "elag_hip_host_set_threads": [_i],
    "elag_hip_host_set_min_work": [_i64],
    "elag_hip_host_keep_pages": [_i],
    "elag_hip_host_amge_dof_agglomeration": [_i64, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p],
    "elag_hip_host_amge_boundary_traces": [_i64, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p, _p],
    "elag_hip_host_amge_local_derivative": [_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "elag_hip_host_local_triplets": [_i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "elag_hip_host_amge_elem_dof_table": [_i64, _p, _p, _p, _p, _i64, _p, _p],
    "elag_hip_host_block_families_to_csr": [_i, _p, _i64, _i64, _p, _p, _p, _p],
    "elag_hip_host_blocks_to_csr": [_i64, _p, _p, _p, _p, _p, _p, _p, _i, _i64, _i64, _p, _p, _p, _p],
    "elag_hip_host_blocks_to_coo": [_i64, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "elag_hip_host_pack_cols": [_i64, _p, _p, _i64, _i64, _p, _p],
    "elag_hip_host_take_leading": [_i64, _p, _p, _p, _p],
    "elag_hip_host_sort_rows_by_key": [_i64, _p, _p, _i64, _p],
    "elag_hip_host_minimal_intersection_sets": [_i64, _p, _p, _p, _i64, _p, _p, _p, _p],
    "elag_hip_coarsen_traces": [_p, _p, _dp],
    "elag_hip_coarsen_extension": [_p, _p, _dp],
    "elag_hip_time_kernel": [_p, _p, _i, _i, _p, _p, _dp],
    "elag_hip_time_hierarchy": [_p, _p, _p, _i, _dp],
    "elag_hip_time_hierarchy_internal": [_p, _i, _dp],
}
_RESTYPES = {"elag_hip_last_error": C.c_char_p, "elag_hip_krylov_launch_mode": C.c_char_p, "elag_hip_host_set_min_work": C.c_int64}

ABI_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_tls = __import__("threading").local()


class ElagHipError(RuntimeError):
    """Raised for any non-zero status of the C ABI (like PARELAG_ASSERT_HYPRE_ERROR_FLAG -> exception)."""

    def __init__(self, code, msg):
        super().__init__(f"elag_hip error {code}: {msg}")
        self.code = code


def load(path: str = LIB_PATH):
    """dlopen the HIP library and declare prototypes.  No GPU is touched."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: run `python -m parelag_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback for the product path.")
    # Kernel arguments in device memory: the HIP runtime of this image places the kernarg segment in host memory on some boxes of the
    # pool and in device memory on others; with host-resident kernargs every wave's first scalar loads cross PCIe, which costs a
    # launch-bound cycle ~2 us per kernel (the 3-launch cube456 x3 cycle: 21.6 us against 15.8 us, tools/headline_probe.py run with
    # HIP_FORCE_DEV_KERNARG=0 / 1 on such a box).  The runtime reads the variable when it is first loaded, so it is only a default
    # for processes that have not loaded HIP yet; an application that sets it (either way) keeps its choice.
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(status: int):
    if status != 0:
        raise ElagHipError(status, (load().elag_hip_last_error() or b"").decode())


def init(device: int | None = None):
    """elag_hip_init on LOCAL_RANK's device (one process per GPU)."""
    lib = load()
    if not getattr(_tls, "initialised", False):   # the library context is per host thread (one thread per rank)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        import time
        t0 = time.perf_counter()
        check(lib.elag_hip_init(device))
        global init_seconds
        init_seconds = time.perf_counter() - t0      # HIP runtime start-up, the library's code objects, the scratch arena's first slab (bench.py reports it as init_s)
        _tls.initialised = True
    return lib


init_seconds = 0.0


def trim_memory() -> int:
    """elag_hip_trim_memory: the pool of freed device blocks and the idle scratch slabs go back to the driver; returns the bytes released."""
    n = C.c_size_t(0)
    check(init().elag_hip_trim_memory(C.byref(n)))
    return int(n.value)


def device_name() -> str:
    buf = C.create_string_buffer(256)
    check(init().elag_hip_device_name(buf, 256))
    return buf.value.decode()


import contextlib


@contextlib.contextmanager
def host_pages_kept():
    """For the duration of one coarsening: large host blocks come from the heap and freed pages stay with the process
    (``elag_hip_host_keep_pages``: glibc's ``M_MMAP_MAX = 0``, ``M_TRIM_THRESHOLD = 2 GiB``) -- the table stage allocates and frees arrays of
    hundreds of MB per level and otherwise faults them in again every time (0.76 -> 0.46 s for the finest coarsening of the SPE10-size
    grid).  The embedding process's allocator policy is glibc's own again when the coarsening returns, and the heap is trimmed
    (``malloc_trim``): a library call must not change the application's allocator for good.  ``ELAG_HIP_HOST_KEEP_PAGES=0`` never touches
    it, ``=1`` (the round-2 behaviour) leaves the setting on for the life of the process."""
    mode = os.environ.get("ELAG_HIP_HOST_KEEP_PAGES", "scoped")
    if mode == "0":
        yield
        return
    lib = load()
    check(lib.elag_hip_host_keep_pages(1))
    try:
        yield
    finally:
        if mode != "1":
            check(lib.elag_hip_host_keep_pages(0))


def keep_host_pages():
    """(kept for callers of round 2) the process-wide form of ``host_pages_kept``; only acts when ELAG_HIP_HOST_KEEP_PAGES=1"""
    if os.environ.get("ELAG_HIP_HOST_KEEP_PAGES", "scoped") == "1":
        check(load().elag_hip_host_keep_pages(1))
