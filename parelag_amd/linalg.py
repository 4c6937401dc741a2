"""Host-side mirror of the reference's operator interface for the V-cycle path, over the C ABI.

Class / method names follow the reference so that tests read like its own code:

* ``ParallelCSRMatrix``      <- mfem::HypreParMatrix as used by ParElag (``Mult``, ``MultTranspose``)
* ``HypreSmootherWrapper``   <- src/linalg/solver_ops/ParELAG_HypreSmootherWrapper.{hpp,cpp}
* ``HiptmairSmoother``       <- src/linalg/solver_ops/ParELAG_HiptmairSmoother.{hpp,cpp}
* ``Hierarchy``              <- src/linalg/solver_ops/ParELAG_Hierarchy.{hpp,cpp}
* ``SolverLibrary`` / factories selected by the ``"Type"`` string <- src/linalg/solver_core, src/linalg/factories

Vectors may be numpy arrays (host; staged by the library), ``DeviceVector`` objects, or anything exposing
``data_ptr()`` on the GPU (a ``torch.Tensor`` on ``cuda``).  Everything executes in libelag_hip.so; there is no
CPU path in this module.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import DEVICE, HOST, ParCSRDesc, ParCSRInfo, check

_vp = C.c_void_p


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(_vp)


class DeviceVector:
    """A length-n fp64 vector in HBM owned through elag_hip_malloc."""

    def __init__(self, n: int, fill: float | None = None):
        self.n = int(n)
        self._lib = _lib.init()
        p = _vp()
        check(self._lib.elag_hip_malloc(C.byref(p), self.n * 8))
        self.ptr = p
        if fill is not None:
            self.copy_from(np.full(self.n, float(fill)))

    @classmethod
    def from_numpy(cls, a):
        v = cls(len(a))
        v.copy_from(a)
        return v

    def copy_from(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        check(self._lib.elag_hip_memcpy(self.ptr, _np_ptr(a), self.n * 8, 0))

    def to_numpy(self):
        out = np.empty(self.n)
        check(self._lib.elag_hip_memcpy(_np_ptr(out), self.ptr, self.n * 8, 1))
        return out

    def data_ptr(self):
        return self.ptr.value

    def __del__(self):
        try:
            if self.ptr:
                self._lib.elag_hip_free(self.ptr)
        except Exception:
            pass


def _vec(v, n=None, writable=False):
    """-> (pointer, loc, keepalive)"""
    if isinstance(v, DeviceVector):
        return v.ptr, DEVICE, v
    if isinstance(v, np.ndarray):
        if v.dtype != np.float64 or not v.flags.c_contiguous:
            if writable:
                raise TypeError("output vectors must be C-contiguous float64 numpy arrays")
            v = np.ascontiguousarray(v, dtype=np.float64)
        if n is not None and v.size != n:
            raise ValueError(f"vector has size {v.size}, expected {n}")
        return _np_ptr(v), HOST, v
    if hasattr(v, "data_ptr"):  # torch tensor on the GPU
        if getattr(v, "is_cuda", True) is False:
            raise TypeError("torch tensors must live on the GPU (use numpy arrays for host vectors)")
        return _vp(v.data_ptr()), DEVICE, v
    raise TypeError(f"unsupported vector type {type(v)}")


def _same_loc(*locs):
    if len(set(locs)) != 1:
        raise TypeError("all vectors of one call must live in the same memory space")
    return locs[0]


def set_gs_hybrid_blocks(nblocks: int):
    """Sweep smoothers ("L1 Gauss-Seidel", "Gauss-Seidel", ...) created after this call treat ``nblocks`` contiguous row blocks as ranks:
    hypre's hybrid Gauss-Seidel as the reference runs it on that many MPI ranks (``elag_hip_set_gs_hybrid_blocks``); 1 = the sequential
    loop (default), 0 = back to the default / ``ELAG_HIP_GS_BLOCKS``."""
    check(_lib.init().elag_hip_set_gs_hybrid_blocks(int(nblocks)))


def synchronize():
    check(_lib.init().elag_hip_synchronize())


class ParallelCSRMatrix:
    """Device-resident ParCSR matrix (one rank's rows)."""

    def __init__(self, handle, owner=True):
        self._lib = _lib.init()
        self.h = handle
        self._owner = owner
        info = ParCSRInfo()
        check(self._lib.elag_hip_parcsr_get_info(self.h, C.byref(info)))
        self.info = info

    # ---- construction --------------------------------------------------------------------------
    @classmethod
    def from_scipy(cls, A):
        """Single-rank matrix from a scipy CSR (storage order is preserved on upload)."""
        A = sp.csr_matrix(A)
        I = np.ascontiguousarray(A.indptr, dtype=np.int32)
        J = np.ascontiguousarray(A.indices, dtype=np.int32)
        a = np.ascontiguousarray(A.data, dtype=np.float64)
        d = ParCSRDesc()
        d.num_rows, d.num_cols_diag = A.shape
        d.diag_i, d.diag_j, d.diag_a = _np_ptr(I), _np_ptr(J), _np_ptr(a)
        d.global_num_rows, d.global_num_cols = A.shape
        return cls._create(d)

    @classmethod
    def from_parts(cls, part):
        """From one rank's pieces as produced by ``parelag_amd.partition.split_rows`` (dict)."""
        keep = []

        def arr(x, dt):
            x = np.ascontiguousarray(x, dtype=dt)
            keep.append(x)
            return _np_ptr(x)
        dg, od = sp.csr_matrix(part["diag"]), sp.csr_matrix(part["offd"])
        d = ParCSRDesc()
        d.num_rows, d.num_cols_diag = dg.shape
        d.diag_i, d.diag_j, d.diag_a = arr(dg.indptr, np.int32), arr(dg.indices, np.int32), arr(dg.data, np.float64)
        d.num_cols_offd = od.shape[1]
        if od.shape[1]:
            d.offd_i, d.offd_j, d.offd_a = arr(od.indptr, np.int32), arr(od.indices, np.int32), arr(od.data, np.float64)
            d.col_map_offd = arr(part["col_map_offd"], np.int64)
        d.first_row, d.first_col_diag = int(part["first_row"]), int(part["first_col"])
        d.global_num_rows, d.global_num_cols = (int(x) for x in part["global_shape"])
        d.num_sends = len(part["send_procs"])
        d.send_procs = arr(part["send_procs"], np.int32)
        d.send_map_starts = arr(part["send_starts"], np.int32)
        d.send_map_elmts = arr(part["send_elmts"], np.int32)
        d.num_recvs = len(part["recv_procs"])
        d.recv_procs = arr(part["recv_procs"], np.int32)
        d.recv_vec_starts = arr(part["recv_starts"], np.int32)
        return cls._create(d)

    @classmethod
    def from_global_rows(cls, rows, first_row, global_rows, first_col, ncols_local, global_cols):
        """This rank's rows with GLOBAL column ids (scipy CSR) -> ParCSR: the diag / offd split, ``col_map_offd`` and the comm package are made
        by the library on the device and over its own communicator (``elag_hip_parcsr_create_from_global_csr``; collective across the ranks)."""
        rows = sp.csr_matrix(rows)
        I = np.ascontiguousarray(rows.indptr, dtype=np.int32)
        J = np.ascontiguousarray(rows.indices, dtype=np.int32)
        a = np.ascontiguousarray(rows.data, dtype=np.float64)
        h = _vp()
        check(_lib.init().elag_hip_parcsr_create_from_global_csr(C.byref(h), rows.shape[0], _np_ptr(I), _np_ptr(J), _np_ptr(a), int(first_row), int(global_rows),
                                                                 int(first_col), int(ncols_local), int(global_cols)))
        return cls(h)

    @classmethod
    def _create(cls, desc):
        lib = _lib.init()
        h = _vp()
        check(lib.elag_hip_parcsr_create(C.byref(h), C.byref(desc)))
        return cls(h)

    @classmethod
    def diagonal(cls, d):
        lib = _lib.init()
        d = np.ascontiguousarray(d, dtype=np.float64)
        h = _vp()
        check(lib.elag_hip_parcsr_diagonal(C.byref(h), len(d), 0, len(d), _np_ptr(d)))
        return cls(h)

    def __del__(self):
        try:
            if self._owner and self.h:
                self._lib.elag_hip_parcsr_destroy(self.h)
        except Exception:
            pass

    # ---- shape ---------------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.info.num_rows, self.info.num_cols_diag)

    def Height(self):
        return self.info.num_rows

    def Width(self):
        return self.info.num_cols_diag

    @property
    def nnz(self):
        return self.info.nnz_diag + self.info.nnz_offd

    # ---- operator application (mfem::Operator contract) --------------------------------------------
    def Mult(self, x, y, alpha=1.0, beta=0.0):
        px, lx, _k1 = _vec(x, self.Width())
        py, ly, _k2 = _vec(y, self.Height(), writable=True)
        check(self._lib.elag_hip_parcsr_spmv(self.h, 0, alpha, px, beta, py, _same_loc(lx, ly)))

    def MultTranspose(self, x, y, alpha=1.0, beta=0.0):
        px, lx, _k1 = _vec(x, self.Height())
        py, ly, _k2 = _vec(y, self.Width(), writable=True)
        check(self._lib.elag_hip_parcsr_spmv(self.h, 1, alpha, px, beta, py, _same_loc(lx, ly)))

    def MultBoolInt(self, x, y, alpha=1, beta=0):
        """hypre_ParCSRMatrixMatvecBoolInt: integer y = alpha*pattern(A)*x + beta*y (host int32 arrays)."""
        x = np.ascontiguousarray(x, dtype=np.int32)
        if y.dtype != np.int32 or not y.flags.c_contiguous:
            raise TypeError("y must be a contiguous int32 array")
        check(self._lib.elag_hip_parcsr_spmv_boolint(self.h, int(alpha), _np_ptr(x), int(beta), _np_ptr(y), HOST))

    def keep_equal_entries(self, B):
        """hypre_ParCSRMatrixKeepEqualEntries(A, B): overwrites this matrix."""
        check(self._lib.elag_hip_parcsr_keep_equal_entries(self.h, B.h))
        check(self._lib.elag_hip_parcsr_get_info(self.h, C.byref(self.info)))

    def MultTransposeIgnoreNonLocal(self, x, y):
        """SharingMap::IgnoreNonLocal: y = diag^T x (no exchange)."""
        px, lx, _k1 = _vec(x, self.Height())
        py, ly, _k2 = _vec(y, self.Width(), writable=True)
        check(self._lib.elag_hip_parcsr_spmvT_ignore_nonlocal(self.h, px, py, _same_loc(lx, ly)))

    def residual(self, b, x, r):
        """r = b - A x  (mg_utils::ComputeResidual)."""
        pb, lb, _k1 = _vec(b, self.Height())
        px, lx, _k2 = _vec(x, self.Width())
        pr, lr, _k3 = _vec(r, self.Height(), writable=True)
        check(self._lib.elag_hip_parcsr_residual(self.h, pb, px, pr, _same_loc(lb, lx, lr)))

    # ---- setup-side algebra ----------------------------------------------------------------------
    def _new(self, fn, *args):
        h = _vp()
        check(fn(*args, C.byref(h)))
        return ParallelCSRMatrix(h)

    def Transpose(self):
        return self._new(self._lib.elag_hip_parcsr_transpose, self.h)

    def matmul(self, B):
        return self._new(self._lib.elag_hip_parcsr_matmul, self.h, B.h)

    def Replicate(self):
        """collective: the whole matrix on every rank -> (rank-local ParallelCSRMatrix, row_starts)"""
        nr = C.c_int(0)
        check(self._lib.elag_hip_comm_rank(None, C.byref(nr)))
        starts = np.zeros(nr.value + 1, np.int64)
        h = _vp()
        check(self._lib.elag_hip_parcsr_replicate(self.h, C.byref(h), _np_ptr(starts)))
        return ParallelCSRMatrix(h), starts

    def fix_zero_rows(self) -> int:
        n = C.c_int(0)
        check(self._lib.elag_hip_parcsr_fix_zero_rows(self.h, C.byref(n)))
        return n.value

    def delete_zeros(self, tol):
        check(self._lib.elag_hip_parcsr_delete_zeros(self.h, tol))
        check(self._lib.elag_hip_parcsr_get_info(self.h, C.byref(self.info)))

    def sign_transform(self):
        check(self._lib.elag_hip_parcsr_sign_transform(self.h))

    def norm(self, kind: str) -> float:
        k = {"l1": 0, "linf": 1, "max": 2, "frobenius": 3}[kind]
        v = C.c_double(0)
        check(self._lib.elag_hip_parcsr_norm(self.h, k, C.byref(v)))
        return v.value

    def set_spmv_lanes(self, lanes: int):
        check(self._lib.elag_hip_parcsr_set_spmv_lanes(self.h, int(lanes)))

    SPMV_FORMS = {0: "spmv_fused_kernel", 1: "spmv_stream_kernel", 2: "spmv_xc_kernel"}

    def spmv_kernel(self) -> str:
        """name of the kernel ``Mult`` is dispatched to as of now (the x-cache decision is taken on the first launch)"""
        form, lanes = C.c_int(0), C.c_int(0)
        check(self._lib.elag_hip_parcsr_get_spmv_form(self.h, C.byref(form), C.byref(lanes)))
        return self.SPMV_FORMS[form.value] + ("<%d>" % lanes.value if form.value == 0 else "")

    def spmv_window(self) -> int:
        """entries per workgroup window of the stream / x-cache kernel ``Mult`` is dispatched to (0 for the row-group kernel)"""
        form, lanes = C.c_int(0), C.c_int(0)
        check(self._lib.elag_hip_parcsr_get_spmv_form(self.h, C.byref(form), C.byref(lanes)))
        return 0 if form.value == 0 else lanes.value

    def get_diagonal(self):
        d = np.empty(self.Height())
        check(self._lib.elag_hip_parcsr_get_diagonal(self.h, _np_ptr(d), HOST))
        return d

    def to_scipy(self, diag_first=False):
        """Export the diag block (and offd mapped to global columns when present)."""
        i = self.info
        I = np.empty(i.num_rows + 1, np.int32); J = np.empty(i.nnz_diag, np.int32); a = np.empty(i.nnz_diag)
        oI = np.empty(i.num_rows + 1, np.int32); oJ = np.empty(i.nnz_offd, np.int32); oa = np.empty(i.nnz_offd)
        cm = np.empty(i.num_cols_offd, np.int64)
        check(self._lib.elag_hip_parcsr_export(self.h, int(diag_first), _np_ptr(I), _np_ptr(J), _np_ptr(a),
                                               _np_ptr(oI), _np_ptr(oJ), _np_ptr(oa), _np_ptr(cm)))
        D = sp.csr_matrix((i.num_rows, i.num_cols_diag))
        D.indptr, D.indices, D.data = I, J, a
        if i.num_cols_offd == 0:
            return D
        O = sp.csr_matrix((i.num_rows, i.num_cols_offd))
        O.indptr, O.indices, O.data = oI, oJ, oa
        return D, O, cm


def RAP(a: ParallelCSRMatrix, b: ParallelCSRMatrix, c: ParallelCSRMatrix | None = None):
    """mfem::RAP(A, P) = P^T A P, or mfem::RAP(Rt, A, P) = Rt^T A P -> hypre_BoomerAMGBuildCoarseOperator
    (call sites Hierarchy.cpp:365,511-513; HiptmairSmootherFactory.cpp:160-161)."""
    Rt, A, P = (b, a, b) if c is None else (a, b, c)
    h = _vp()
    check(_lib.init().elag_hip_parcsr_rap(Rt.h, A.h, P.h, C.byref(h)))
    return ParallelCSRMatrix(h)


def IgnoreNonLocalRange(Rt: ParallelCSRMatrix, A: ParallelCSRMatrix, P: ParallelCSRMatrix):
    """IgnoreNonLocalRange(RT, A, P) (MatrixUtils.cpp:878-916): RAP with RT's off-rank columns set aside."""
    h = _vp()
    check(_lib.init().elag_hip_parcsr_rap_ignore_nonlocal_range(Rt.h, A.h, P.h, C.byref(h)))
    return ParallelCSRMatrix(h)


def Assemble(range_e_tE: ParallelCSRMatrix, A_blockdiag: ParallelCSRMatrix, domain_e_tE: ParallelCSRMatrix):
    """Assemble(SharingMap, SparseMatrix, SharingMap) (SharingMap.cpp:975-1011): e_tE^T blockdiag(A) e_tE."""
    return RAP(range_e_tE, A_blockdiag, domain_e_tE)


def Add(a, A: ParallelCSRMatrix, b, B: ParallelCSRMatrix):
    """hypre_ParCSRMatrixAdd2 (src/hypreExtension/parcsr-add.c:199-245)."""
    h = _vp()
    check(_lib.init().elag_hip_parcsr_add(a, A.h, b, B.h, C.byref(h)))
    return ParallelCSRMatrix(h)


def RDP(Rt: ParallelCSRMatrix, d, P: ParallelCSRMatrix):
    """hypre_RDP (src/hypreExtension/par_Tmatmul.c:17-39)."""
    pd, ld, _k = _vec(d, Rt.Height())
    h = _vp()
    check(_lib.init().elag_hip_parcsr_rdp(Rt.h, pd, ld, P.h, C.byref(h)))
    return ParallelCSRMatrix(h)


def Compare(A: ParallelCSRMatrix, B: ParallelCSRMatrix, tol: float) -> int:
    """hypre_ParCSRMatrixCompare bit mask."""
    m = C.c_int(0)
    check(_lib.init().elag_hip_parcsr_compare(A.h, B.h, tol, C.byref(m)))
    return m.value


# ---- smoothers ------------------------------------------------------------------------------------

# string -> hypre relax type, the table of src/linalg/factories/ParELAG_HypreSmootherFactory.cpp:92-109
HYPRE_SMOOTHER_TYPES = {
    "Jacobi": 0, "L1 Jacobi": 1, "L1 Gauss-Seidel": 2, "Kaczmarz": 3, "L1 Gauss-Seidel Truncated": 4,
    "Lumped Jacobi": 5, "Gauss-Seidel": 6, "Chebyshev": 16, "Taubin": 1001, "FIR": 1002, "Hypre Jacobi": 413,
}


class _SolverBase:
    iterative_mode = True

    def IsPreconditioner(self):
        return not self.iterative_mode


class HypreSmootherWrapper(_SolverBase):
    """HypreSmootherWrapper(A, type, params): params uses the reference's ParameterList keys
    ("Sweeps", "Damping Factor", "Omega", "Cheby Poly Order", "Cheby Poly Fraction")."""

    def __init__(self, A: ParallelCSRMatrix, type: int = 1, params: dict | None = None, handle=None, owner=True):
        params = params or {}
        self._lib = _lib.init()
        self.A = A
        self.type = type
        self._owner = owner
        if handle is None:
            handle = _vp()
            check(self._lib.elag_hip_smoother_create(
                C.byref(handle), A.h, int(type), int(params.get("Sweeps", 1)), float(params.get("Damping Factor", 1.0)),
                float(params.get("Omega", 1.0)), int(params.get("Cheby Poly Order", 2)),
                float(params.get("Cheby Poly Fraction", 0.3))))
        self.h = handle
        self.iterative_mode = True

    def eig_estimate(self):
        mx, mn = C.c_double(), C.c_double()
        check(self._lib.elag_hip_smoother_get_eig_estimate(self.h, C.byref(mx), C.byref(mn)))
        return mx.value, mn.value

    def set_eig_estimate(self, max_eig, min_eig):
        check(self._lib.elag_hip_smoother_set_eig_estimate(self.h, max_eig, min_eig))

    def _apply(self, rhs, sol, transpose):
        n = self.A.Height()
        pr, lr, _k1 = _vec(rhs, n)
        ps, ls, _k2 = _vec(sol, n, writable=True)
        check(self._lib.elag_hip_smoother_apply(self.h, pr, ps, int(not self.iterative_mode), int(transpose), _same_loc(lr, ls)))

    def Mult(self, rhs, sol):
        self._apply(rhs, sol, False)

    def MultTranspose(self, rhs, sol):
        self._apply(rhs, sol, True)

    def __del__(self):
        try:
            if self._owner and self.h:
                self._lib.elag_hip_smoother_destroy(self.h)
        except Exception:
            pass


class HierarchySolver(HypreSmootherWrapper):
    """A finalized ``Hierarchy`` as an inner solver object -- what the parameter lists call an AMGe solver in an
    "A00 Inverse" / "S Inverse" entry (``example_parameterlists/spe10_example_parameters.xml:114-127``).  ``scale = -1``
    applies the negative of the cycle (hierarchy built on ``B D^-1 B^T``, used as the inverse of ``-B D^-1 B^T``)."""

    def __init__(self, H: "Hierarchy", scale: float = 1.0):
        lib = _lib.init()
        handle = _vp()
        check(lib.elag_hip_smoother_from_hierarchy(C.byref(handle), H.h, float(scale)))
        A0 = H.GetOperator(0)
        super().__init__(A0, type=-1, handle=handle, owner=True)
        self._keep = [H, A0]
        self.iterative_mode = False


class HiptmairSmoother(HypreSmootherWrapper):
    """HiptmairSmoother(A, A_aux, D, primary, auxiliary) (ParELAG_HiptmairSmoother.cpp:22-46)."""

    def __init__(self, A, D, primary: HypreSmootherWrapper, auxiliary: HypreSmootherWrapper):
        self._lib = _lib.init()
        self.A, self.D, self.primary, self.auxiliary = A, D, primary, auxiliary
        self._owner = True
        h = _vp()
        check(self._lib.elag_hip_hiptmair_create(C.byref(h), A.h, D.h, primary.h, auxiliary.h))
        self.h = h
        self.iterative_mode = True


class CoarseSolver(_SolverBase):
    def __init__(self, A: ParallelCSRMatrix, kind: int = 0, iters: int = 0):
        self._lib = _lib.init()
        self.A = A
        h = _vp()
        check(self._lib.elag_hip_coarse_create(C.byref(h), A.h, kind, iters))
        self.h = h
        self.iterative_mode = False

    def Mult(self, rhs, sol):
        pr, lr, _k1 = _vec(rhs, self.A.Height())
        ps, ls, _k2 = _vec(sol, self.A.Height(), writable=True)
        check(self._lib.elag_hip_coarse_apply(self.h, pr, ps, _same_loc(lr, ls)))

    def __del__(self):
        try:
            if self.h:
                self._lib.elag_hip_coarse_destroy(self.h)
        except Exception:
            pass


# ---- hierarchy --------------------------------------------------------------------------------------

class Hierarchy(_SolverBase):
    """parelag::Hierarchy.  ``iterative_mode`` False <=> IsPreconditioner() (Solver.hpp:57-103)."""

    def __init__(self, num_levels: int | None = None, handle=None):
        self._lib = _lib.init()
        self._keep = []
        if handle is None:
            handle = _vp()
            check(self._lib.elag_hip_hierarchy_create(C.byref(handle), int(num_levels)))
        self.h = handle
        self.iterative_mode = False

    def SetOperator(self, level, A):
        self._keep.append(A)
        check(self._lib.elag_hip_hierarchy_set_operator(self.h, level, A.h))

    def SetProlongator(self, coarse_level, P):
        self._keep.append(P)
        check(self._lib.elag_hip_hierarchy_set_prolongator(self.h, coarse_level, P.h))

    def SetSmoothers(self, level, pre, post):
        self._keep += [pre, post]
        check(self._lib.elag_hip_hierarchy_set_smoothers(self.h, level, pre.h if pre else None, post.h if post else None))

    def SetCoarseSolver(self, solver):
        self._keep.append(solver)
        check(self._lib.elag_hip_hierarchy_set_coarse_solver(self.h, solver.h))

    def Finalize(self, use_graph=True):
        check(self._lib.elag_hip_hierarchy_finalize(self.h, int(use_graph)))

    def Mult(self, rhs, sol):
        if type(rhs) is DeviceVector and type(sol) is DeviceVector:
            # the launch-bound case (a cycle is a handful of 4 us kernels): no per-call work on the host beyond the call itself
            st = self._lib.elag_hip_hierarchy_apply(self.h, rhs.ptr, sol.ptr, 0 if self.iterative_mode else 1, DEVICE)
            if st:
                check(st)
            return
        pr, lr, _k1 = _vec(rhs)
        ps, ls, _k2 = _vec(sol, writable=True)
        check(self._lib.elag_hip_hierarchy_apply(self.h, pr, ps, int(self.IsPreconditioner()), _same_loc(lr, ls)))

    def GetOperator(self, level) -> ParallelCSRMatrix:
        h = _vp()
        check(self._lib.elag_hip_hierarchy_get_operator(self.h, level, C.byref(h)))
        return ParallelCSRMatrix(h, owner=False)

    def GetSmoother(self, level) -> HypreSmootherWrapper:
        h = _vp()
        check(self._lib.elag_hip_hierarchy_get_smoother(self.h, level, C.byref(h)))
        return HypreSmootherWrapper(self.GetOperator(level), handle=h, owner=False)

    def GetOrdering(self, level):
        """new -> old row map of a level (identity unless the hierarchy was built with reordering)."""
        n = self.GetOperator(level).Height()
        o = np.empty(n, np.int32)
        check(self._lib.elag_hip_hierarchy_get_ordering(self.h, level, _np_ptr(o)))
        return o

    def launch_mode(self) -> dict:
        """how Finalize decided to issue a cycle: replayed hipGraph or plain launches, with the two timings when it measured them"""
        g, k = C.c_int(), C.c_int()
        tg, te = C.c_double(), C.c_double()
        check(self._lib.elag_hip_hierarchy_launch_mode(self.h, C.byref(g), C.byref(k), C.byref(tg), C.byref(te)))
        return {"hipgraph": bool(g.value), "kernels_per_cycle": k.value, "graph_us": round(tg.value, 2), "plain_launches_us": round(te.value, 2)}

    def setup_breakdown(self) -> dict:
        """milliseconds of wall time per setup phase (Galerkin products, renumbering, smoothers, coarse solver, fused level operators,
        SpMV forms, dense tail, flattened-cycle operators, launch mode): the reference's "Build Hierarchy" timer by phase"""
        buf = C.create_string_buffer(2048)
        check(self._lib.elag_hip_hierarchy_setup_breakdown(self.h, buf, 2048))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.split("=")
                out[k] = float(v)
        return out

    def time_apply_internal(self, reps) -> float:
        """mean milliseconds per V-cycle on vectors in the hierarchy's own numbering (the cost inside a Krylov solve)"""
        ms = C.c_double(0)
        check(self._lib.elag_hip_time_hierarchy_internal(self.h, int(reps), C.byref(ms)))
        return ms.value

    def time_apply(self, rhs, sol, reps) -> float:
        """mean milliseconds per V-cycle, measured with HIP events on the library's compute stream"""
        pr, lr, _k1 = _vec(rhs)
        ps, ls, _k2 = _vec(sol, writable=True)
        if _same_loc(lr, ls) != DEVICE:
            raise TypeError("time_apply needs device vectors")
        ms = C.c_double(0)
        check(self._lib.elag_hip_time_hierarchy(self.h, pr, ps, int(reps), C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.h:
                self._lib.elag_hip_hierarchy_destroy(self.h)
        except Exception:
            pass


def _amge_options(lib, smoother, params, fix_zero_rows=True, coarse_kind=0, use_graph=True, reorder=-1):
    params = params or {}
    opt = _lib.AMGeOptions()
    check(lib.elag_hip_amge_default_options(C.byref(opt)))
    opt.fix_zero_rows = int(fix_zero_rows)
    opt.relax_type = HYPRE_SMOOTHER_TYPES[smoother] if isinstance(smoother, str) else int(smoother)
    opt.sweeps = int(params.get("Sweeps", 1))
    opt.damping = float(params.get("Damping Factor", 1.0))
    opt.cheby_order = int(params.get("Cheby Poly Order", 2))
    opt.cheby_fraction = float(params.get("Cheby Poly Fraction", 0.3))
    opt.coarse_kind = int(coarse_kind)
    opt.use_graph = int(use_graph)
    opt.reorder = int(reorder)
    return opt


def buildHierarchyFromDeRhamSequence(A: ParallelCSRMatrix, Ps, smoother="L1 Jacobi", params: dict | None = None,
                                     fix_zero_rows=True, coarse_kind=0, use_graph=True, reorder=-1,
                                     Ds=None, aux_smoother="L1 Jacobi", aux_params: dict | None = None) -> Hierarchy:
    """RAP chain + smoothers + coarse solve in one ABI call (Hierarchy.cpp:282-383, AMGeSolverFactory.cpp:26-169).
    ``Ps[l]`` is the true interpolator from level l+1 to level l (DeRhamSequence::ComputeTrueP).  With ``Ds`` (the true
    exterior derivative into the form, one per non-coarsest level) every level gets a HiptmairSmoother whose primary /
    auxiliary relaxations are ``smoother`` / ``aux_smoother`` ("Primary Smoother" / "Auxiliary Smoother")."""
    lib = _lib.init()
    arr = (_vp * max(len(Ps), 1))(*[P.h for P in Ps])
    opt = _amge_options(lib, smoother, params, fix_zero_rows, coarse_kind, use_graph, reorder)
    h = _vp()
    if Ds is None:
        check(lib.elag_hip_hierarchy_build(C.byref(h), A.h, len(Ps), arr, C.byref(opt)))
    else:
        if len(Ds) < len(Ps):
            raise ValueError("need one D per non-coarsest level")
        darr = (_vp * max(len(Ps), 1))(*[D.h for D in Ds[:len(Ps)]])
        aux = _amge_options(lib, aux_smoother, aux_params)
        check(lib.elag_hip_hierarchy_build_hiptmair(C.byref(h), A.h, len(Ps), arr, darr, C.byref(opt), C.byref(aux)))
    H = Hierarchy(handle=h)
    H._keep += [A, *Ps, *(Ds or [])]
    return H


# ---- blocked (saddle-point) layer ---------------------------------------------------------------------------------

class MfemBlockOperator:
    """parelag::MfemBlockOperator (src/linalg/utilities/ParELAG_MfemBlockOperator.hpp:30-135): blocks are
    ParallelCSRMatrix handles over contiguous block vectors with the given local offsets.  On the device the blocks are
    also merged into one ParCSR (``Monolithic()``) so that Mult is a single fused SpMV."""

    def __init__(self, row_offsets, col_offsets=None):
        self._lib = _lib.init()
        self.row_offsets = np.ascontiguousarray(row_offsets, dtype=np.int32)
        self.col_offsets = self.row_offsets if col_offsets is None else np.ascontiguousarray(col_offsets, dtype=np.int32)
        h = _vp()
        check(self._lib.elag_hip_blockop_create(C.byref(h), len(self.row_offsets) - 1, len(self.col_offsets) - 1,
                                                _np_ptr(self.row_offsets), _np_ptr(self.col_offsets)))
        self.h = h
        self._blocks = {}

    def GetNumBlockRows(self):
        return len(self.row_offsets) - 1

    def GetNumBlockCols(self):
        return len(self.col_offsets) - 1

    def Height(self):
        return int(self.row_offsets[-1])

    def Width(self):
        return int(self.col_offsets[-1])

    def SetBlock(self, i, j, op: ParallelCSRMatrix):
        check(self._lib.elag_hip_blockop_set_block(self.h, int(i), int(j), op.h))
        self._blocks[(i, j)] = op

    def IsZeroBlock(self, i, j):
        h = _vp()
        check(self._lib.elag_hip_blockop_get_block(self.h, int(i), int(j), C.byref(h)))
        return not h

    def GetBlock(self, i, j) -> ParallelCSRMatrix:
        return self._blocks[(i, j)]

    def Monolithic(self) -> ParallelCSRMatrix:
        h = _vp()
        check(self._lib.elag_hip_blockop_monolithic(self.h, C.byref(h)))
        M = ParallelCSRMatrix(h, owner=False)
        M._keep = self
        return M

    def _mult(self, trans, x, y, alpha, beta):
        px, lx, _k1 = _vec(x, self.Height() if trans else self.Width())
        py, ly, _k2 = _vec(y, self.Width() if trans else self.Height(), writable=True)
        check(self._lib.elag_hip_blockop_mult(self.h, int(trans), alpha, px, beta, py, _same_loc(lx, ly)))

    def Mult(self, x, y, alpha=1.0, beta=0.0):
        self._mult(False, x, y, alpha, beta)

    def MultTranspose(self, x, y, alpha=1.0, beta=0.0):
        self._mult(True, x, y, alpha, beta)

    def __del__(self):
        try:
            if self.h:
                self._lib.elag_hip_blockop_destroy(self.h)
        except Exception:
            pass


SCHUR_TYPES = {"NONE": 0, "DIAGONAL": 1, "ABSROWSUM": 2}
BLOCK_SMOOTHERS = {"Block Jacobi": 0, "Block GS": 1, "Block GS Lower": 1, "Block GS Upper": 2, "Block LDU": 3}


def SchurComplement(op: MfemBlockOperator, s_type="DIAGONAL", alpha=1.0, negate=False) -> ParallelCSRMatrix:
    """SchurComplementFactory::BuildOperator (ParELAG_SchurComplementFactory.cpp:36-181): A11 - alpha A10 D^-1 A01."""
    h = _vp()
    check(_lib.init().elag_hip_schur_complement(op.h, SCHUR_TYPES[s_type.upper()], float(alpha), int(negate), C.byref(h)))
    return ParallelCSRMatrix(h)


class BlockSolver(HypreSmootherWrapper):
    """BlockDiagonalSolver / BlockTriangularSolver / Block2x2LDUInverseOperator over device smoothers.
    ``inv_ops`` are smoother objects (HypreSmootherWrapper, HiptmairSmoother ...): one per block, or the four LDU inverses
    {A00_1, A00_2, A00_3, S}."""

    def __init__(self, op: MfemBlockOperator, inv_ops, kind="Block Jacobi", damping=1.0):
        self._lib = _lib.init()
        self.op, self.inv_ops = op, list(inv_ops)
        self.A = op                       # Height() for the vector checks
        self._owner = True
        arr = (_vp * len(self.inv_ops))(*[s.h for s in self.inv_ops])
        h = _vp()
        check(self._lib.elag_hip_block_smoother_create(C.byref(h), op.h, BLOCK_SMOOTHERS[kind] if isinstance(kind, str) else int(kind),
                                                       len(self.inv_ops), arr, float(damping)))
        self.h = h
        self.iterative_mode = True


def BlockDiagonalSolver(op, inv_ops):
    return BlockSolver(op, inv_ops, "Block Jacobi")


def BlockTriangularSolver(op, inv_ops, triangle="Lower"):
    return BlockSolver(op, inv_ops, "Block GS Lower" if triangle.lower().startswith("l") else "Block GS Upper")


def Block2x2LDUInverseOperator(op, invA00_1, invA00_2, invA00_3, invS, damping=1.0):
    return BlockSolver(op, [invA00_1, invA00_2, invA00_3, invS], "Block LDU", damping)


def darcy_ldu_amge_preconditioner(op: MfemBlockOperator, M, B, Ps, smoother="L1 Jacobi", params=None, dPs=None):
    """The reference's SPE10 solver graph (``examples/example_parameterlists/spe10_example_parameters.xml:95-127``) for the mixed system
    ``[[M, B^T], [B, 0]]`` on one rank: ``Block2x2LDUInverseOperator`` whose three ``A00`` inverses are AMGe V-cycles on ``M`` (2-form
    interpolators ``Ps[l][0]``) and whose ``S`` inverse is an AMGe V-cycle on ``B diag(M)^-1 B^T`` (3-form interpolators ``Ps[l][1]``),
    negated.  Everything is formed on the device from the blocks ``op`` already holds: ``M = op.GetBlock(0, 0)``, the Schur complement by
    ``SchurComplementFactory``'s own product (``elag_hip_schur_complement``), the interpolators from ``dPs`` (device matrices, ``dPs[l][i]`` =
    block i of level l+1 -> l) when the caller has uploaded them for the blocked hierarchy already -- round 3 formed ``B diag(M)^-1 B^T`` with
    scipy on the host and uploaded ``M``, the product and both interpolator families a second time: 114 ms for a 13 ms solve at cube456 x3.
    ``M``, ``B`` (scipy) are only read when ``op`` lacks a block; ``Ps`` (scipy) only without ``dPs``.  The N-rank form is
    ``distributed.build_darcy_ldu_amge``."""
    try:
        dM = op.GetBlock(0, 0)
        Spos = SchurComplement(op, "DIAGONAL", 1.0, negate=True)                  # + B diag(M)^-1 B^T  (the A11 block is empty)
    except KeyError:
        Mm, Bm = sp.csr_matrix(M), sp.csr_matrix(B)
        Sh = (Bm @ sp.diags(1.0 / Mm.diagonal()) @ Bm.T).tocsr()
        Sh.sort_indices()
        dM, Spos = ParallelCSRMatrix.from_scipy(Mm), ParallelCSRMatrix.from_scipy(Sh)
    if dPs is None:
        dPs = [[ParallelCSRMatrix.from_scipy(sp.csr_matrix(P[0])), ParallelCSRMatrix.from_scipy(sp.csr_matrix(P[1]))] for P in Ps]
    HM = buildHierarchyFromDeRhamSequence(dM, [P[0] for P in dPs], smoother, params)
    HS = buildHierarchyFromDeRhamSequence(Spos, [P[1] for P in dPs], smoother, params)
    iM, iS = HierarchySolver(HM), HierarchySolver(HS, scale=-1.0)
    ldu = Block2x2LDUInverseOperator(op, iM, iM, iM, iS)
    ldu.iterative_mode = False
    ldu._keep_ldu = [HM, HS, iM, iS, dM, Spos, dPs]
    return ldu


def buildBlockedHierarchyFromDeRhamSequence(A: MfemBlockOperator, Ps, block_smoother="Block Jacobi",
                                            a00_inverse="L1 Jacobi", a00_params=None, a11_inverse="L1 Jacobi", a11_params=None,
                                            s_type="DIAGONAL", alpha=1.0, use_negative_s=True, damping=1.0,
                                            fix_zero_rows=None, coarse_kind=1, use_graph=True) -> Hierarchy:
    """buildBlockedHierarchyFromDeRhamSequence (Hierarchy.cpp:400-541) with the block smoother of the
    Block2x2{Jacobi,GaussSeidel,LDU}SolverFactory on every level.  ``Ps[l][i]`` = block i of the interpolator level l+1 -> l."""
    lib = _lib.init()
    nb = A.GetNumBlockRows()
    flat = [P.h for Pl in Ps for P in Pl]
    if any(len(Pl) != nb for Pl in Ps):
        raise ValueError("every level needs one interpolator per block")
    arr = (_vp * max(len(flat), 1))(*flat)
    opt = _lib.BlockedOptions()
    check(lib.elag_hip_blocked_default_options(C.byref(opt)))
    opt.block_smoother = BLOCK_SMOOTHERS[block_smoother]
    opt.inv_a00 = _amge_options(lib, a00_inverse, a00_params)
    opt.inv_a11 = _amge_options(lib, a11_inverse, a11_params)
    opt.s_type = SCHUR_TYPES[s_type.upper()]
    opt.alpha, opt.use_negative_s, opt.damping = float(alpha), int(use_negative_s), float(damping)
    opt.coarse_kind, opt.use_graph = int(coarse_kind), int(use_graph)
    fz = None if fix_zero_rows is None else np.ascontiguousarray(fix_zero_rows, dtype=np.int32)
    h = _vp()
    check(lib.elag_hip_hierarchy_build_blocked(C.byref(h), A.h, len(Ps), arr, _np_ptr(fz), C.byref(opt)))
    H = Hierarchy(handle=h)
    H._keep += [A, *[P for Pl in Ps for P in Pl]]
    return H


# ---- Krylov shells ----------------------------------------------------------------------------------------------------

KRYLOV_SOLVERS = {"CG": 0, "PCG": 0, "GMRES": 1, "FGMRES": 2, "STATIONARY": 3, "BICGSTAB": 4, "MINRES": 5}


class KrylovSolver(_SolverBase):
    """parelag::KrylovSolver(A, Prec, params) (src/linalg/solver_ops/ParELAG_KrylovSolver.cpp:23-96) on the device.
    ``params`` uses the reference's keys: "Solver name" (CG | PCG | GMRES | FGMRES; "Stationary" selects
    parelag::StationarySolver), "Relative tolerance", "Absolute tolerance", "Maximum iterations", "Restart size".
    ``Prec`` is any solver object of this module (smoother, block solver, Hierarchy, CoarseSolver) or None."""

    def __init__(self, A, Prec=None, params: dict | None = None):
        params = params or {}
        self._lib = _lib.init()
        self.A_user = A
        self.A = A.Monolithic() if isinstance(A, MfemBlockOperator) else A
        self.Prec = Prec
        name = str(params.get("Solver name", "CG")).upper()
        if name not in KRYLOV_SOLVERS:
            raise ValueError(f'KrylovSolver::KrylovSolver(...): Bad solver type ("{name}"). Valid choices are "CG", "PCG", "GMRES", '
                             f'"FGMRES", "BiCGSTAB", "MINRES" (KrylovSolver.cpp:44-51) and "Stationary"')
        opt = _lib.KrylovOptions()
        check(self._lib.elag_hip_krylov_default_options(C.byref(opt)))
        opt.solver = KRYLOV_SOLVERS[name]
        opt.max_iter = int(params.get("Maximum iterations", params.get("Maximum Iterations", 1 if name == "STATIONARY" else 10)))
        opt.rel_tol = float(params.get("Relative tolerance", params.get("Relative Tolerance", 0.0)))
        opt.abs_tol = float(params.get("Absolute tolerance", params.get("Absolute Tolerance", 0.0)))
        opt.restart = int(params.get("Restart size", 50))
        opt.check_every = int(params.get("Check every", 8))
        opt.use_graph = int(params.get("Use graph", True))
        if Prec is None:
            kind, ph = 0, None
        elif isinstance(Prec, Hierarchy):
            kind, ph = 2, Prec.h
        elif isinstance(Prec, CoarseSolver):
            kind, ph = 3, Prec.h
        else:
            kind, ph = 1, Prec.h
        h = _vp()
        check(self._lib.elag_hip_krylov_create(C.byref(h), self.A.h, kind, ph, C.byref(opt)))
        self.h = h
        self.iterative_mode = False      # KrylovSolver.cpp:77

    def Mult(self, rhs, sol):
        n = self.A.Height()
        pr, lr, _k1 = _vec(rhs, n)
        ps, ls, _k2 = _vec(sol, n, writable=True)
        check(self._lib.elag_hip_krylov_mult(self.h, pr, ps, int(self.iterative_mode), _same_loc(lr, ls)))

    def _stats(self):
        c, it, nrm = C.c_int(), C.c_int(), C.c_double()
        check(self._lib.elag_hip_krylov_get_stats(self.h, C.byref(c), C.byref(it), C.byref(nrm)))
        return c.value, it.value, nrm.value

    def GetConverged(self):
        return bool(self._stats()[0])

    def GetNumIterations(self):
        return self._stats()[1]

    def GetFinalNorm(self):
        return self._stats()[2]

    def launch_mode(self) -> str:
        """How the last Mult issued its iterations: "graph", "plain" or "plain: <why a capture was given up>" (diagnostic)."""
        return (self._lib.elag_hip_krylov_launch_mode(self.h) or b"").decode()

    def __del__(self):
        try:
            if self.h:
                self._lib.elag_hip_krylov_destroy(self.h)
        except Exception:
            pass


def StationarySolver(A, solver, rel_tol=0.0, abs_tol=0.0, max_its=1):
    """parelag::StationarySolver(op, solver, rel_tol, abs_tol, max_its) (ParELAG_StationarySolver.cpp:23-39); it is
    constructed as a preconditioner (IsPreconditioner() == true), i.e. iterative_mode False."""
    return KrylovSolver(A, solver, {"Solver name": "Stationary", "Relative Tolerance": rel_tol, "Absolute Tolerance": abs_tol,
                                    "Maximum Iterations": max_its})


def time_kernel(A: ParallelCSRMatrix, kernel: str, x, y, reps=20, smoother: HypreSmootherWrapper | None = None) -> float:
    """mean ms per launch of a fine-level kernel ('spmv' | 'residual' | 'jacobi') by HIP events."""
    kid = {"spmv": 0, "residual": 1, "jacobi": 2}[kernel]
    px, lx, _k1 = _vec(x)
    py, ly, _k2 = _vec(y, writable=True)
    if _same_loc(lx, ly) != DEVICE:
        raise TypeError("time_kernel needs device vectors")
    ms = C.c_double(0)
    check(_lib.init().elag_hip_time_kernel(A.h, smoother.h if smoother else None, kid, int(reps), px, py, C.byref(ms)))
    return ms.value


# ---- agglomerate index maps (host-side integer logic of the library; no GPU needed) ------------------------------

def partition_refined(n_elements: int, n_parts: int, ndim: int = 3, legacy_numbering: bool = False):
    """MFEMRefinedMeshPartitioner::Partition (src/partitioning/MFEMRefinedMeshPartitioner.cpp:50-95)."""
    p = np.empty(n_elements, np.int32)
    check(_lib.load().elag_hip_partition_refined(n_elements, n_parts, ndim, int(legacy_numbering), _np_ptr(p)))
    return p


def connected_components(partitioning, conn):
    """connectedComponents (src/structures/connectedComponents.cpp:23-85) -> (number of components, new partitioning)."""
    conn = sp.csr_matrix(conn)
    p = np.ascontiguousarray(partitioning, dtype=np.int32).copy()
    I = np.ascontiguousarray(conn.indptr, dtype=np.int32); J = np.ascontiguousarray(conn.indices, dtype=np.int32)
    n = C.c_int(0)
    check(_lib.load().elag_hip_connected_components(len(p), _np_ptr(p), _np_ptr(I), _np_ptr(J), C.byref(n)))
    return n.value, p


def transpose_partitioning(partitioning, n_parts: int):
    """transposePartitioning (src/structures/transpose.cpp:27-55) -> (i, j) of the part -> elements table."""
    p = np.ascontiguousarray(partitioning, dtype=np.int32)
    I = np.empty(n_parts + 1, np.int32); J = np.empty(int((p >= 0).sum()), np.int32)
    check(_lib.load().elag_hip_transpose_partitioning(len(p), _np_ptr(p), n_parts, _np_ptr(I), _np_ptr(J)))
    return I, J
