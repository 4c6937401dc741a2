"""ctypes front-end of the CPU ORACLE (test infrastructure -- see the header of elag_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED at kernel granularity (hypre/MFEM absent; see elag_oracle.c header and DESIGN.md).

The ParCSR-level routines compose the C kernels exactly where the reference places the halo
exchange (src/hypreExtension/hypre_ParCSRMatrixMatvecBoolInt.c:148-201, :212-345); the exchange
itself is a pluggable object so the same code runs serially, with all ranks in one process, or
across processes (torch.distributed/gloo in tests).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_DIR = os.path.dirname(os.path.abspath(__file__))
_c_int_p = C.POINTER(C.c_int)
_c_dbl_p = C.POINTER(C.c_double)


def build(native: bool = False, omp: bool = False) -> str:
    """native: -march=native (timing); omp: the multi-core timing build (-fopenmp, see the header of elag_oracle.c).
    The parity checks use the default serial, generic build."""
    if omp:
        out, target = "libelag_oracle_native_omp.so", "OMPOUT"
    else:
        out, target = ("libelag_oracle_native.so" if native else "libelag_oracle.so"), "OUT"
    path = os.path.join(_DIR, out)
    src = os.path.join(_DIR, "elag_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        args = ["make", "-C", _DIR, f"{target}={out}", out]
        if native or omp:
            args.append("ARCH=-march=native")
        subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return path


_libs = {}


def lib(native: bool = False, omp: bool = False):
    key = (native, omp)
    if key not in _libs:
        L = C.CDLL(build(native, omp))
        L.orc_csr_max_norm.restype = C.c_double
        L.orc_csr_frobenius_norm.restype = C.c_double
        L.orc_hier_create.restype = C.c_void_p
        _libs[key] = L
    return _libs[key]


def _ip(a):
    return None if a is None else a.ctypes.data_as(_c_int_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(_c_dbl_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def csr(A) -> sp.csr_matrix:
    """Canonical oracle CSR: int32 indices, fp64 values, storage order preserved."""
    A = sp.csr_matrix(A)
    A.indptr = _i32(A.indptr); A.indices = _i32(A.indices); A.data = _f64(A.data)
    return A


def raw_csr(n_rows, n_cols, I, J, a) -> sp.csr_matrix:
    M = sp.csr_matrix((n_rows, n_cols))
    M.indptr, M.indices, M.data = _i32(I), _i32(J), _f64(a)
    return M


def diag_first(A: sp.csr_matrix) -> sp.csr_matrix:
    """Reorder each row of a square CSR so that the diagonal entry is stored first (hypre / MFEM
    convention for square ParCSR diag blocks)."""
    A = csr(A).copy()
    I, J, a = A.indptr, A.indices, A.data
    for i in range(A.shape[0]):
        s, e = I[i], I[i + 1]
        k = np.nonzero(J[s:e] == i)[0]
        if len(k) and k[0] != 0:
            k = s + k[0]
            J[s + 1:k + 1], J[s] = J[s:k].copy(), i
            v = a[k]
            a[s + 1:k + 1] = a[s:k].copy(); a[s] = v
    return A


# ---- CSR kernels --------------------------------------------------------------------------------

def csr_matvec(A, alpha, x, beta, y):
    lib().orc_csr_matvec(C.c_int(A.shape[0]), _ip(A.indptr), _ip(A.indices), _dp(A.data),
                         C.c_double(alpha), _dp(x), C.c_double(beta), _dp(y))


def csr_matvecT(A, alpha, x, beta, y):
    lib().orc_csr_matvecT(C.c_int(A.shape[0]), C.c_int(A.shape[1]), _ip(A.indptr), _ip(A.indices),
                          _dp(A.data), C.c_double(alpha), _dp(x), C.c_double(beta), _dp(y))


def csr_matvec_boolint(A, alpha, x, beta, y):
    lib().orc_csr_matvec_boolint(C.c_int(A.shape[0]), _ip(A.indptr), _ip(A.indices), C.c_int(alpha),
                                 _ip(x), C.c_int(beta), _ip(y))


def transpose(A) -> sp.csr_matrix:
    A = csr(A)
    n, m = A.shape
    TI = np.empty(m + 1, np.int32); TJ = np.empty(A.nnz, np.int32); Ta = np.empty(A.nnz)
    lib().orc_csr_transpose(C.c_int(n), C.c_int(m), _ip(A.indptr), _ip(A.indices), _dp(A.data),
                            _ip(TI), _ip(TJ), _dp(Ta))
    return raw_csr(m, n, TI, TJ, Ta)


def matmul(A, B) -> sp.csr_matrix:
    A, B = csr(A), csr(B)
    n = A.shape[0]
    CI = np.empty(n + 1, np.int32)
    nnz = lib().orc_csr_matmul_count(C.c_int(n), C.c_int(B.shape[1]), _ip(A.indptr), _ip(A.indices),
                                     _ip(B.indptr), _ip(B.indices), _ip(CI))
    CJ = np.empty(nnz, np.int32); Ca = np.empty(nnz)
    lib().orc_csr_matmul_fill(C.c_int(n), C.c_int(B.shape[1]), _ip(A.indptr), _ip(A.indices), _dp(A.data),
                              _ip(B.indptr), _ip(B.indices), _dp(B.data), _ip(CI), _ip(CJ), _dp(Ca))
    return raw_csr(n, B.shape[1], CI, CJ, Ca)


def rap(RT, A, P) -> sp.csr_matrix:
    """hypre_BoomerAMGBuildCoarseOperator(RT, A, P) on one rank (diag blocks)."""
    RT, A, P = csr(RT), csr(A), csr(P)
    R = transpose(RT)
    ncr, ncc, nf = RT.shape[1], P.shape[1], A.shape[1]   # A_marker spans A's columns (rectangular blocks: B = W D)
    sq = int(ncr == ncc)
    CI = np.empty(ncr + 1, np.int32)
    args = (C.c_int(ncr), C.c_int(nf), C.c_int(ncc), C.c_int(sq), _ip(R.indptr), _ip(R.indices), _dp(R.data),
            _ip(A.indptr), _ip(A.indices), _dp(A.data), _ip(P.indptr), _ip(P.indices), _dp(P.data))
    nnz = lib().orc_rap(*args, _ip(CI), None, None)
    CJ = np.empty(nnz, np.int32); Ca = np.empty(nnz)
    lib().orc_rap(*args, _ip(CI), _ip(CJ), _dp(Ca))
    return raw_csr(ncr, ncc, CI, CJ, Ca)


def mfem_rap(Rt, A, P) -> sp.csr_matrix:
    """mfem::RAP(Rt, A, P) on serial SparseMatrix: Transpose(Rt), R*A, then (RA)*P."""
    return matmul(matmul(transpose(Rt), A), P)


def add(a, A, b, B) -> sp.csr_matrix:
    A, B = csr(A), csr(B)
    n, m = A.shape
    CI = np.empty(n + 1, np.int32)
    nnz = lib().orc_csr_add_count(C.c_int(n), C.c_int(m), _ip(A.indptr), _ip(A.indices), _ip(B.indptr),
                                  _ip(B.indices), _ip(CI))
    CJ = np.empty(nnz, np.int32); Ca = np.empty(nnz)
    lib().orc_csr_add_fill(C.c_int(n), C.c_int(m), C.c_double(a), _ip(A.indptr), _ip(A.indices), _dp(A.data),
                           C.c_double(b), _ip(B.indptr), _ip(B.indices), _dp(B.data), _ip(CI), _ip(CJ), _dp(Ca))
    return raw_csr(n, m, CI, CJ, Ca)


def delete_zeros(A, tol) -> sp.csr_matrix:
    A = csr(A)
    OI = np.empty(A.shape[0] + 1, np.int32); OJ = np.empty(A.nnz, np.int32); Oa = np.empty(A.nnz)
    nnz = lib().orc_csr_delete_zeros(C.c_int(A.shape[0]), _ip(A.indptr), _ip(A.indices), _dp(A.data),
                                     C.c_double(tol), _ip(OI), _ip(OJ), _dp(Oa))
    return raw_csr(A.shape[0], A.shape[1], OI, OJ[:nnz].copy(), Oa[:nnz].copy())


def sign_transform(A) -> sp.csr_matrix:
    A = csr(A).copy()
    lib().orc_sign_transform(C.c_int(A.nnz), _dp(A.data))
    return A


def max_norm(A):
    return lib().orc_csr_max_norm(C.c_int(A.nnz), _dp(_f64(A.data)))


def frobenius_norm(A):
    return lib().orc_csr_frobenius_norm(C.c_int(A.nnz), _dp(_f64(A.data)))


def fix_zero_rows(diag, offd=None) -> int:
    return lib().orc_fix_zero_rows(C.c_int(diag.shape[0]), _ip(diag.indptr), _ip(diag.indices), _dp(diag.data),
                                   _ip(offd.indptr) if offd is not None else None,
                                   _dp(offd.data) if offd is not None else None)


def l1_norms(diag, offd=None, option=1) -> np.ndarray:
    l1 = np.empty(diag.shape[0])
    bad = lib().orc_l1_norms(C.c_int(diag.shape[0]), _ip(diag.indptr), _ip(diag.indices), _dp(diag.data),
                             _ip(offd.indptr) if offd is not None else None,
                             _dp(offd.data) if offd is not None else None, C.c_int(option), _dp(l1))
    if bad:
        raise RuntimeError(f"hypre_ParCSRComputeL1Norms: {bad} zero l1 norms")
    return l1


def block_starts(n, nblocks):
    """row blocks of the device's virtual ranks (sweep.hip) and of orc_hier_spmd_bench's type-2 levels: n // nblocks rows each, the first
    n % nblocks blocks one more"""
    size, rest = divmod(int(n), int(nblocks))
    return np.concatenate([[0], np.cumsum([size + (1 if b < rest else 0) for b in range(int(nblocks))])]).astype(np.int64)


def l1_norms_blocks(A, nblocks, option=2) -> np.ndarray:
    """hypre_ParCSRComputeL1Norms (options 1, 2, 4) of a square operator split like a ParCSR matrix on ``nblocks`` ranks with the contiguous row
    blocks of ``block_starts``: the entries of a row outside its own block are its offd part."""
    A = csr(A)
    n = A.shape[0]
    if int(nblocks) <= 1:
        return l1_norms(A, None, option)
    st = block_starts(n, nblocks)
    blk = np.searchsorted(st, np.arange(n), side="right") - 1
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    inside = blk[A.indices] == blk[rows]
    keep = lambda m: sp.csr_matrix((A.data[m], A.indices[m], np.concatenate([[0], np.cumsum(np.bincount(rows[m], minlength=n))]).astype(np.int32)), shape=A.shape)
    return l1_norms(keep(inside), keep(~inside), option)


def cheby_coefs(max_eig, min_eig, fraction, order):
    c = np.zeros(5)
    co = lib().orc_cheby_coefs(C.c_double(max_eig), C.c_double(min_eig), C.c_double(fraction), C.c_int(order), _dp(c))
    return co, c


def hypre_random(n, seed=1):
    v = np.empty(n)
    lib().orc_hypre_random(C.c_int(n), C.c_int(seed), _dp(v))
    return v


def tql1(d, e):
    d = _f64(d).copy(); e = _f64(e).copy()
    err = lib().orc_tql1(C.c_int(len(d)), _dp(d), _dp(e))
    if err:
        raise RuntimeError("tql1 failed")
    return d


# ---- index maps ---------------------------------------------------------------------------------

def partition_refined(n_elements, n_parts):
    p = np.empty(n_elements, np.int32)
    lib().orc_partition_refined(C.c_int(n_elements), C.c_int(n_parts), _ip(p))
    return p


def partition_refined_legacy(n_elements, n_parts, ndim=3):
    p = np.empty(n_elements, np.int32)
    lib().orc_partition_refined_legacy(C.c_int(n_elements), C.c_int(n_parts), C.c_int(ndim), _ip(p))
    return p


def connected_components(partitioning, conn):
    conn = csr(conn)
    p = _i32(partitioning).copy()
    n = lib().orc_connected_components(C.c_int(len(p)), _ip(p), _ip(conn.indptr), _ip(conn.indices))
    return n, p


def transpose_partitioning(partitioning, nparts):
    p = _i32(partitioning)
    I = np.empty(nparts + 1, np.int32); J = np.empty(int((p > -1).sum()), np.int32)
    lib().orc_transpose_partitioning(C.c_int(len(p)), _ip(p), C.c_int(nparts), _ip(I), _ip(J))
    return I, J


# ---- ParCSR with a pluggable halo exchange --------------------------------------------------------

class SerialExchange:
    """One rank: there are no neighbours (hypre's comm handle with num_sends = num_recvs = 0)."""
    rank, size = 0, 1

    def forward(self, pkg, send_buf):
        return np.empty(0)

    def reverse(self, pkg, ghost_buf):
        return np.empty(0)

    def allreduce_sum(self, v):
        return v

    def allreduce_max(self, v):
        return v


class OParCSR:
    """Oracle-side ParCSR: diag + offd CSR, col_map_offd, comm package lists."""

    def __init__(self, diag, offd=None, col_map_offd=None, pkg=None, exch=None,
                 first_row=0, first_col=0, global_shape=None):
        self.diag = csr(diag)
        n = self.diag.shape[0]
        self.offd = csr(offd) if offd is not None else csr(sp.csr_matrix((n, 0)))
        self.col_map_offd = np.asarray(col_map_offd if col_map_offd is not None else [], dtype=np.int64)
        self.pkg = pkg   # dict(send_procs, send_starts, send_elmts, recv_procs, recv_starts)
        self.exch = exch or SerialExchange()
        self.first_row, self.first_col = first_row, first_col
        self.global_shape = global_shape or self.diag.shape

    @property
    def shape(self):
        return self.diag.shape

    def _send_buf(self, x):
        if self.pkg is None or len(self.pkg["send_elmts"]) == 0:
            return np.empty(0)
        el = _i32(self.pkg["send_elmts"])
        buf = np.empty(len(el))
        lib().orc_gather(C.c_int(len(el)), _ip(el), _dp(x), _dp(buf))
        return buf

    def matvec(self, alpha, x, beta, y):
        """hypre_ParCSRMatrixMatvec: pack, start exchange, diag product, wait, offd product (beta=1)."""
        x = _f64(x)
        ghost = self.exch.forward(self.pkg, self._send_buf(x))
        csr_matvec(self.diag, alpha, x, beta, y)
        if self.offd.shape[1]:
            csr_matvec(self.offd, alpha, _f64(ghost), 1.0, y)

    def matvecT(self, alpha, x, beta, y):
        """hypre_ParCSRMatrixMatvecT: offd^T x -> y_tmp, reverse exchange, diag^T x, scatter-add."""
        x = _f64(x)
        ytmp = np.zeros(self.offd.shape[1])
        if self.offd.shape[1]:
            csr_matvecT(self.offd, alpha, x, 0.0, ytmp)
        ybuf = self.exch.reverse(self.pkg, ytmp)
        csr_matvecT(self.diag, alpha, x, beta, y)
        if self.pkg is not None and len(self.pkg["send_elmts"]):
            el = _i32(self.pkg["send_elmts"])
            lib().orc_scatter_add(C.c_int(len(el)), _ip(el), _dp(_f64(ybuf)), _dp(y))

    def ghost(self, x):
        return _f64(self.exch.forward(self.pkg, self._send_buf(_f64(x))))

    def norm_linf(self):
        """hypre_ParCSRMatrixNormlinf (hypre_ParCSRMatrixNorms.c:74-127)."""
        Aabs = OParCSR(abs(self.diag), abs(self.offd), self.col_map_offd, self.pkg, self.exch)
        y = np.zeros(self.shape[0])
        Aabs.matvec(1.0, np.ones(self.shape[1]), 0.0, y)
        return self.exch.allreduce_max(max(0.0, y.max(initial=0.0)))

    def norm_l1(self):
        """hypre_ParCSRMatrixNorml1 (:18-72): max entry of |A|^T 1."""
        Aabs = OParCSR(abs(self.diag), abs(self.offd), self.col_map_offd, self.pkg, self.exch)
        y = np.zeros(self.shape[1])
        Aabs.matvecT(1.0, np.ones(self.shape[0]), 0.0, y)
        return self.exch.allreduce_max(max(0.0, y.max(initial=0.0)))

    def norm_max(self):
        return self.exch.allreduce_max(max(max_norm(self.diag), max_norm(self.offd)))

    def norm_frobenius(self):
        return float(np.sqrt(self.exch.allreduce_sum(frobenius_norm(self.diag) ** 2 + frobenius_norm(self.offd) ** 2)))


def compare(A: OParCSR, B: OParCSR, tol: float) -> int:
    """hypre_ParCSRMatrixCompare (hypre_ParCSRMatrixCompare.c:125-195): bit mask of mismatches."""
    cmp = 0
    if A.global_shape[0] != B.global_shape[0]: cmp |= 1
    if A.global_shape[1] != B.global_shape[1]: cmp |= 2
    if A.first_row != B.first_row: cmp |= 4
    if A.first_row + A.shape[0] != B.first_row + B.shape[0]: cmp |= 8
    if A.first_col != B.first_col: cmp |= 16
    if A.first_col + A.shape[1] != B.first_col + B.shape[1]: cmp |= 32

    def merged(M):
        ncol = M.global_shape[1]
        d = sp.csr_matrix((M.diag.data, M.diag.indices + M.first_col, M.diag.indptr), shape=(M.shape[0], ncol))
        if M.offd.shape[1]:
            o = sp.csr_matrix((M.offd.data, M.col_map_offd[M.offd.indices], M.offd.indptr), shape=(M.shape[0], ncol))
            return add(1.0, d, 1.0, o)
        return csr(d)
    if cmp & 3 == 0 and A.shape[0] == B.shape[0]:
        c = add(1.0, merged(A), -1.0, merged(B))
        diff = A.exch.allreduce_sum(max_norm(c))
        if diff > tol: cmp |= 64
    else:
        cmp |= 64
    return cmp


# ---- smoothers ------------------------------------------------------------------------------------

class OSmoother:
    """mfem::HypreSmoother as driven by HypreSmootherWrapper (ParELAG_HypreSmootherWrapper.cpp:28-34):
    types 1 (l1-Jacobi), 2 / 4 (l1-GS, l1 norms option 2 / 4; hypre's three branches: relax_weight = omega = 1, relax_weight = 1 "SSOR",
    relax_weight != 1 "scaled SSOR"), 6 (hybrid symmetric GS; (1, 1) and the weighted / over-relaxed branch),
    16 (Chebyshev) and 413 ("Hypre Jacobi" = mfem::HypreDiagScale, HypreSmootherFactory.cpp:28-31).  ``iterative_mode`` as in
    mfem::Solver.  Also 3 (Kaczmarz: hypre_ParCSRRelax with the rows' squared 2-norms), 5 (lumped Jacobi) and the two polynomial
    smoothers of mfem::HypreSmoother, 1001 (Taubin: ``taubin_iter`` steps u += c r / max_eig, c alternating lambda, mu) and 1002 (FIR:
    windowed Chebyshev low-pass of order ``order``, pass band ``fraction * max_eig``), with mfem's default options (lambda 0.5, mu -0.5,
    40 steps, Blackman window 0.42 / 0.50 / 0.08; the reference's factory cannot set others: HypreSmootherFactory.cpp:38-40) and
    max_eig = hypre_ParCSRMaxEigEstimateCG(A, scale 0, 10) / 2.  Not restated: 0 (plain Jacobi through BoomerAMGRelax)."""

    def __init__(self, A: OParCSR, type=1, sweeps=1, weight=1.0, omega=1.0, order=2, fraction=0.3,
                 eig=None, taubin=(0.5, -0.5, 40), window=(0.42, 0.50, 0.08)):
        self.A, self.type, self.sweeps, self.weight, self.omega = A, type, sweeps, weight, omega
        self.order, self.fraction = order, fraction
        self.iterative_mode = True
        n = A.shape[0]
        if type in (1, 2, 3, 4):
            self.l1 = l1_norms(A.diag, A.offd, option=type)
        if type == 5:
            # mfem::HypreSmoother::SetOperator: l1_norms = A * 1 (row sums, ghost columns included), then the relaxation of type 1
            self.l1 = np.zeros(n)
            A.matvec(1.0, np.ones(A.shape[1]), 0.0, self.l1)
        if type in (1001, 1002):
            # mfem::HypreSmoother::SetOperator: poly_scale = 0; the Taubin and FIR polynomials are defined on [0, 2]
            self.max_eig, self.min_eig = eig if eig is not None else max_eig_estimate_cg(A, 0, 10)
            self.max_eig /= 2.0
            self.taubin = taubin
            if type == 1002:                                     # SetFIRCoefficients(max_eig)
                i = np.arange(order + 1)
                t = i * np.pi / (order + 1)
                win = window[0] + window[1] * np.cos(t) + window[2] * np.cos(2 * t)
                theta = np.arccos(1.0 - 0.5 * fraction * self.max_eig)
                ch = np.empty(order + 1)
                ch[0] = theta / np.pi
                ch[1:] = 2.0 * np.sin(i[1:] * theta) / (i[1:] * np.pi)
                self.fir = win * ch
        if type == 16:
            self.max_eig, self.min_eig = eig if eig is not None else max_eig_estimate_cg(A, 1, 10)
            self.co, self.coefs = cheby_coefs(self.max_eig, self.min_eig, fraction, order)
            self.ds = np.empty(n)
            lib().orc_inv_sqrt_diag(C.c_int(n), _ip(A.diag.indptr), _dp(A.diag.data), _dp(self.ds))

    def Mult(self, f, u):
        A, n = self.A, self.A.shape[0]
        f = _f64(f)
        if not self.iterative_mode:
            u[:] = 0.0
        L = lib()
        for _ in range(self.sweeps):
            if self.type in (1, 5):
                v = f.copy()
                A.matvec(-self.weight, u, self.weight, v)
                L.orc_jacobi_update(C.c_int(n), _dp(v), _dp(self.l1), _dp(u))
            elif self.type == 6:
                ue = A.ghost(u)
                has = A.offd.shape[1] > 0
                args = (C.c_int(n), _ip(A.diag.indptr), _ip(A.diag.indices), _dp(A.diag.data),
                        _ip(A.offd.indptr) if has else None, _ip(A.offd.indices) if has else None,
                        _dp(A.offd.data) if has else None, _dp(f), _dp(ue))
                if self.weight == 1.0 and self.omega == 1.0:
                    L.orc_gs_sweep(*args, _dp(u))
                else:
                    vt = np.empty(n)
                    L.orc_gs_sweep_sor(*args, C.c_double(self.weight), C.c_double(self.omega), _dp(u), _dp(vt))
            elif self.type == 3:
                ue = A.ghost(u)
                has = A.offd.shape[1] > 0
                L.orc_kaczmarz_sweep(C.c_int(n), _ip(A.diag.indptr), _ip(A.diag.indices), _dp(A.diag.data),
                                     _ip(A.offd.indptr) if has else None, _ip(A.offd.indices) if has else None,
                                     _dp(A.offd.data) if has else None, _dp(f), _dp(self.l1), _dp(ue), C.c_double(self.omega), _dp(u))
            elif self.type == 1001:                              # ParCSRRelax_Taubin
                lam, mu, N = self.taubin
                r = np.empty(n)
                for i in range(N):
                    r[:] = f
                    A.matvec(-1.0, u, 1.0, r)
                    coef = lam if i % 2 == 0 else mu
                    u += coef * r / self.max_eig
            elif self.type == 1002:                              # ParCSRRelax_FIR
                x0 = u.copy()
                x1 = f.copy()
                A.matvec(-1.0, x0, 1.0, x1)
                x1 /= -self.max_eig
                x1 = x0 - x1
                x3 = self.fir[0] * x0 + self.fir[1] * x1
                for k in range(2, self.order + 1):
                    x2 = f.copy()
                    A.matvec(-1.0, x1, 1.0, x2)
                    x2 /= -self.max_eig
                    x2 = (x1 - x0) + (x1 - 2 * x2)
                    x3 += self.fir[k] * x2
                    x0, x1 = x1, x2
                u[:] = x3
            elif self.type == 413:
                L.orc_diag_scale(C.c_int(n), _ip(A.diag.indptr), _dp(A.diag.data), _dp(f), _dp(u))
            elif self.type in (2, 4):
                ue = A.ghost(u)
                has = A.offd.shape[1] > 0
                args = (C.c_int(n), _ip(A.diag.indptr), _ip(A.diag.indices), _dp(A.diag.data),
                        _ip(A.offd.indptr) if has else None, _ip(A.offd.indices) if has else None,
                        _dp(A.offd.data) if has else None, _dp(f), _dp(self.l1), _dp(ue))
                if self.weight != 1.0:
                    vt = np.empty(n)
                    L.orc_l1gs_sweep_scaled(*args, C.c_double(self.weight), C.c_double(self.omega), _dp(u), _dp(vt))
                elif self.omega == 1.0:
                    L.orc_l1gs_sweep(*args, _dp(u))
                else:
                    L.orc_l1gs_sweep_omega(*args, C.c_double(self.omega), _dp(u))
            elif self.type == 16:
                tmp = np.zeros(n); r = np.empty(n); orig = np.empty(n); v = np.zeros(n)
                A.matvec(-1.0, u, 0.0, tmp)
                L.orc_cheby_r(C.c_int(n), _dp(self.ds), _dp(f), _dp(tmp), _dp(r))
                L.orc_cheby_start(C.c_int(n), C.c_double(self.coefs[self.co]), _dp(r), _dp(u), _dp(orig))
                for i in range(self.co - 1, -1, -1):
                    L.orc_vec_mul(C.c_int(n), _dp(self.ds), _dp(u), _dp(tmp))
                    A.matvec(1.0, tmp, 0.0, v)
                    L.orc_cheby_step(C.c_int(n), C.c_double(self.coefs[i]), _dp(r), _dp(self.ds), _dp(v), _dp(u))
                L.orc_cheby_finish(C.c_int(n), _dp(orig), _dp(self.ds), _dp(u))
            else:
                raise NotImplementedError(self.type)


def max_eig_estimate_cg(A: OParCSR, scale=1, max_iter=10, seed=1):
    """hypre_ParCSRMaxEigEstimateCG [3P par_relax_more.c]: ``max_iter`` CG/Lanczos steps on
    D^-1/2 A D^-1/2 from a hypre-RNG start vector; eigenvalues of the tridiagonal via tql1."""
    n = A.shape[0]
    ex = A.exch
    size = int(A.global_shape[0])
    max_iter = min(max_iter, size)
    r = hypre_random(n, seed * (ex.rank + 1))
    ds = np.ones(n)
    if scale:
        lib().orc_inv_sqrt_diag(C.c_int(n), _ip(A.diag.indptr), _dp(A.diag.data), _dp(ds))
    tridiag = np.zeros(max_iter + 1); trioffd = np.zeros(max_iter + 1)
    p = np.zeros(n); s = np.zeros(n); u = np.zeros(n)
    gamma = 1.0
    i = 0
    while i < max_iter:
        s[:] = r
        gamma_old = gamma
        gamma = ex.allreduce_sum(float(r @ s))
        if i == 0:
            beta = 1.0
            p[:] = s
        else:
            beta = gamma / gamma_old
            p[:] = s + beta * p
        if scale:
            u[:] = ds * p
            A.matvec(1.0, u, 0.0, s)
            s *= ds
        else:
            A.matvec(1.0, p, 0.0, s)
        sdotp = ex.allreduce_sum(float(s @ p))
        alpha = gamma / sdotp
        alphainv = 1.0 / alpha
        tridiag[i + 1] = alphainv
        tridiag[i] *= beta
        tridiag[i] += alphainv
        trioffd[i + 1] = alphainv
        trioffd[i] *= np.sqrt(beta)
        r -= alpha * s
        i += 1
    ev = tql1(tridiag[:i], trioffd[:i])
    return float(ev[i - 1]), float(ev[0])


class OHiptmair:
    """HiptmairSmoother::Mult / MultTranspose (ParELAG_HiptmairSmoother.cpp:48-109)."""

    def __init__(self, A: OParCSR, D: OParCSR, primary: OSmoother, aux: OSmoother):
        self.A, self.D, self.primary, self.aux = A, D, primary, aux
        self.iterative_mode = True
        self.primary.iterative_mode = True
        self.aux.iterative_mode = False

    def _residual(self, X, B):
        tmp = np.zeros(self.A.shape[0])
        self.A.matvec(1.0, X, 0.0, tmp)
        tmp -= B
        tmp *= -1
        return tmp

    def _aux_correct(self, R, X):
        auxb = np.zeros(self.D.shape[1]); auxx = np.zeros(self.D.shape[1])
        self.D.matvecT(1.0, R, 0.0, auxb)
        self.aux.Mult(auxb, auxx)
        pv = np.zeros(self.A.shape[0])
        self.D.matvec(1.0, auxx, 0.0, pv)
        X += pv

    def Mult(self, B, X):
        if not self.iterative_mode:
            X[:] = 0.0
        self.primary.Mult(B, X)
        self._aux_correct(self._residual(X, B), X)

    def MultTranspose(self, B, X):
        if not self.iterative_mode:
            X[:] = 0.0
            self._aux_correct(_f64(B), X)
        else:
            self._aux_correct(self._residual(X, B), X)
        self.primary.Mult(B, X)


class OHierarchy:
    """Hierarchy::Mult/Iterate (ParELAG_Hierarchy.cpp:109-253), composed from ParCSR pieces so it works
    with any exchange.  levels[l] = dict(A=OParCSR, P=OParCSR (l+1 -> l) or None, pre=..., post=...,
    coarse=callable(rhs, sol) or None)."""

    def __init__(self, levels, preconditioner=True):
        self.levels, self.preconditioner = levels, preconditioner

    def Mult(self, rhs, sol):
        A0 = self.levels[0]["A"]
        if self.preconditioner:
            r = _f64(rhs)
            s = sol
        else:
            r = np.zeros(len(rhs))
            A0.matvec(1.0, sol, 0.0, r)
            r *= -1.0
            r += rhs
            s = np.empty(len(sol))
        s[:] = 0.0
        self._iterate(r, s, 0)
        if not self.preconditioner:
            sol += s

    def _iterate(self, RHS, SOL, l):
        L = self.levels[l]
        if l == len(self.levels) - 1:
            if L.get("coarse") is not None:
                L["coarse"](RHS, SOL)
            else:
                if L.get("pre") is not None: L["pre"].Mult(RHS, SOL)
                if L.get("post") is not None: L["post"].Mult(RHS, SOL)
            return
        Cl = self.levels[l + 1]
        if L.get("pre") is not None:
            L["pre"].Mult(RHS, SOL)
        n = L["A"].shape[0]
        tmp_correct = np.zeros(n)
        L["A"].matvec(1.0, SOL, 0.0, tmp_correct)
        tmp_resid = RHS + (-1.0) * tmp_correct
        P = Cl["P"]
        cres = np.zeros(P.shape[1])
        P.matvecT(1.0, tmp_resid, 0.0, cres)
        csol = np.zeros(P.shape[1])
        self._iterate(cres, csol, l + 1)
        P.matvec(1.0, csol, 0.0, tmp_correct)
        SOL += tmp_correct
        if L.get("post") is not None:
            post = L["post"]
            if L.get("post_transpose"):
                post.MultTranspose(RHS, SOL)
            else:
                post.Mult(RHS, SOL)


# ---- blocked (saddle-point) layer -------------------------------------------------------------------

class OBlockOperator:
    """MfemBlockOperator::Mult / MultTranspose (ParELAG_MfemBlockOperator.cpp:93-103), i.e. mfem::BlockOperator:
    y_i = sum_j A_ij x_j accumulated block by block, j ascending.  ``blocks`` maps (i, j) -> OParCSR."""

    def __init__(self, row_off, col_off, blocks):
        self.row_off, self.col_off, self.blocks = list(row_off), list(col_off), dict(blocks)
        self.nbr, self.nbc = len(self.row_off) - 1, len(self.col_off) - 1

    @property
    def shape(self):
        return (self.row_off[-1], self.col_off[-1])

    def GetBlock(self, i, j):
        return self.blocks.get((i, j))

    def _apply(self, x, trans):
        x = _f64(x)
        ro, co = (self.col_off, self.row_off) if trans else (self.row_off, self.col_off)
        out = np.zeros(ro[-1])
        for i in range(len(ro) - 1):
            for j in range(len(co) - 1):
                blk = self.blocks.get((j, i) if trans else (i, j))
                if blk is None:
                    continue
                tmp = np.zeros(ro[i + 1] - ro[i])
                xs = np.ascontiguousarray(x[co[j]:co[j + 1]])
                (blk.matvecT if trans else blk.matvec)(1.0, xs, 0.0, tmp)
                out[ro[i]:ro[i + 1]] += tmp
        return out

    def matvec(self, alpha, x, beta, y):
        y *= beta
        y += alpha * self._apply(x, False)

    def matvecT(self, alpha, x, beta, y):
        y *= beta
        y += alpha * self._apply(x, True)


def schur_complement(A00, A01, A10, A11=None, s_type="DIAGONAL", alpha=1.0):
    """SchurComplementFactory::BuildOperator (ParELAG_SchurComplementFactory.cpp:52-170) on serial CSR blocks:
    A11 - alpha * A10 * D^-1 * A01 with D = diag(A00) or its absolute row sums."""
    A00, A01, A10 = csr(A00), csr(A01), csr(A10)
    if s_type.upper() == "DIAGONAL":
        d = A00.diagonal()
    elif s_type.upper() == "ABSROWSUM":
        d = np.asarray(abs(A00).sum(axis=1)).ravel()
    else:
        raise ValueError(f'Schur complement type = "{s_type}" is invalid.')
    tmp = csr(A01.copy())
    tmp.data = tmp.data / np.repeat(d, np.diff(tmp.indptr))        # InvScaleRows
    prod = matmul(A10, tmp)
    if A11 is not None:
        return add(1.0, csr(A11), -1.0 * alpha, prod)
    prod = csr(prod)
    prod.data = prod.data * (-1.0 * alpha)
    return prod


class _OBlockSolverBase:
    def __init__(self, A: OBlockOperator, inv_ops):
        self.A, self.inv_ops = A, list(inv_ops)
        self.iterative_mode = True
        for s in self.inv_ops:
            s.iterative_mode = False                               # Block2x2*SolverFactory: inv_ops[i]->iterative_mode = false

    def _residual(self, rhs, sol):
        r = np.zeros(len(rhs))
        self.A.matvec(1.0, sol, 0.0, r)
        r *= -1.0
        r += rhs
        return r

    def _run(self, rhs, sol, solve):
        rhs = _f64(rhs)
        if self.iterative_mode:                                    # not IsPreconditioner(): Resid = rhs - A sol
            r = self._residual(rhs, sol)
            corr = np.zeros(len(sol))
            solve(r, corr)
            sol += corr
        else:
            sol[:] = 0.0
            solve(rhs, sol)

    def _blk(self, v, i):
        o = self.A.row_off
        return v[o[i]:o[i + 1]]


class OBlockDiagonalSolver(_OBlockSolverBase):
    """BlockDiagonalSolver::Mult (ParELAG_BlockDiagonalSolver.cpp:67-112)."""

    def Mult(self, rhs, sol, transpose=False):
        def solve(r, x):
            for i, inv in enumerate(self.inv_ops):
                xi = np.zeros(len(self._blk(x, i)))
                (inv.MultTranspose if transpose and hasattr(inv, "MultTranspose") else inv.Mult)(np.ascontiguousarray(self._blk(r, i)), xi)
                self._blk(x, i)[:] = xi
        self._run(rhs, sol, solve)

    def MultTranspose(self, rhs, sol):
        self.Mult(rhs, sol, transpose=True)


class OBlockTriangularSolver(_OBlockSolverBase):
    """BlockTriangularSolver (ParELAG_BlockTriangularSolver.cpp:73-300): lower counts up, upper counts down; the
    transposes swap the direction and use A_ji^T."""

    def __init__(self, A, inv_ops, lower=True):
        super().__init__(A, inv_ops)
        self.lower = lower

    def _sweep(self, r, x, transpose):
        nb = len(self.inv_ops)
        ascending = (self.lower != transpose)
        order = range(nb) if ascending else range(nb - 1, -1, -1)
        for i in order:
            ri = np.array(self._blk(r, i), dtype=np.float64)
            others = range(0, i) if ascending else range(i + 1, nb)
            for j in others:
                blk = self.A.GetBlock(j, i) if transpose else self.A.GetBlock(i, j)
                if blk is None:
                    continue
                tmp = np.zeros(len(ri))
                (blk.matvecT if transpose else blk.matvec)(1.0, np.ascontiguousarray(self._blk(x, j)), 0.0, tmp)
                ri -= tmp
            xi = np.zeros(len(ri))
            inv = self.inv_ops[i]
            (inv.MultTranspose if transpose and hasattr(inv, "MultTranspose") else inv.Mult)(ri, xi)
            self._blk(x, i)[:] = xi

    def Mult(self, rhs, sol):
        self._run(rhs, sol, lambda r, x: self._sweep(r, x, False))

    def MultTranspose(self, rhs, sol):
        self._run(rhs, sol, lambda r, x: self._sweep(r, x, True))


class OBlock2x2LDU(_OBlockSolverBase):
    """Block2x2LDUInverseOperator::Mult (ParELAG_Block2x2LDUInverseOperator.cpp:73-137)."""

    def __init__(self, A, invA00_1, invA00_2, invA00_3, invS, damping=1.0):
        super().__init__(A, [invA00_1, invA00_2, invA00_3, invS])
        self.damping = damping
        self.iterative_mode = False                                # constructed as a preconditioner (:31)

    def Mult(self, rhs, sol):
        i1, i2, i3, iS = self.inv_ops

        def solve(r, c):
            r = np.array(r, dtype=np.float64)
            r0, r1 = self._blk(r, 0), self._blk(r, 1)
            n0, n1 = len(r0), len(r1)
            t0 = np.zeros(n0)
            i2.Mult(np.ascontiguousarray(r0), t0)                  # du~ = A_2^-1 r_f
            t1 = np.zeros(n1)
            self.A.GetBlock(1, 0).matvec(1.0, t0, 0.0, t1)
            r1 -= t1                                               # r_g - A10 du~
            c1 = np.zeros(n1)
            iS.Mult(np.ascontiguousarray(r1), c1)                  # dp
            c0 = np.zeros(n0)
            self.A.GetBlock(0, 1).matvec(1.0, c1, 0.0, c0)         # A01 dp
            t0 = np.zeros(n0)
            i3.Mult(c0, t0)                                        # A_3^-1 A01 dp
            c0 = np.zeros(n0)
            i1.Mult(np.ascontiguousarray(r0), c0)                  # du^
            c0 -= t0
            self._blk(c, 0)[:] = c0
            self._blk(c, 1)[:] = c1
            if self.damping != 1.0:
                c *= self.damping
        self._run(rhs, sol, solve)


# ---- Krylov shells (mfem::CGSolver / GMRESSolver / FGMRESSolver, parelag::StationarySolver) -------------------------

class KrylovResult:
    def __init__(self, converged, final_iter, final_norm):
        self.converged, self.final_iter, self.final_norm = int(converged), int(final_iter), float(final_norm)


def mfem_cg(A_mul, prec, b, x, iterative_mode=False, rel_tol=0.0, abs_tol=0.0, max_iter=10, dot=np.dot):
    """mfem::CGSolver::Mult (MFEM 4.x linalg/solvers.cpp [3P]; selected by KrylovSolver.cpp:44-45).  ``A_mul(v)`` returns
    A v, ``prec(r, z)`` writes z = B r (or None)."""
    b = _f64(b)
    if iterative_mode:
        r = b - A_mul(x)
    else:
        r = b.copy(); x[:] = 0.0
    z = np.zeros_like(r)
    if prec is not None:
        prec(r, z); d = z.copy()
    else:
        d = r.copy()
    nom = dot(d, r)
    if nom < 0.0:
        return KrylovResult(0, 0, nom)
    r0 = max(nom * rel_tol * rel_tol, abs_tol * abs_tol)
    if nom <= r0:
        return KrylovResult(1, 0, np.sqrt(nom))
    z = A_mul(d)
    den = dot(z, d)
    if den == 0.0:
        return KrylovResult(0, 0, np.sqrt(nom))
    converged, final_iter = 0, max_iter
    i = 1
    while True:
        alpha = nom / den
        x += alpha * d
        r += (-alpha) * z
        if prec is not None:
            z = np.zeros_like(r); prec(r, z)
            betanom = dot(r, z)
        else:
            betanom = dot(r, r)
        if betanom < 0.0:
            converged, final_iter = 0, i
            break
        if betanom <= r0:
            converged, final_iter = 1, i
            break
        i += 1
        if i > max_iter:
            break
        beta = betanom / nom
        d = (z if prec is not None else r) + beta * d
        z = A_mul(d)
        den = dot(d, z)
        if den == 0.0:
            final_iter = i
            break
        nom = betanom
    return KrylovResult(converged, final_iter, np.sqrt(betanom) if betanom >= 0 else float("nan"))


def _gen_rot(dx, dy):
    if dy == 0.0:
        return 1.0, 0.0
    if abs(dy) > abs(dx):
        t = dx / dy; sn = 1.0 / np.sqrt(1.0 + t * t); return t * sn, sn
    t = dy / dx; cs = 1.0 / np.sqrt(1.0 + t * t); return cs, t * cs


def _app_rot(dx, dy, cs, sn):
    return cs * dx + sn * dy, -sn * dx + cs * dy


def _gmres_update(x, k, H, s, V):
    y = s.copy()
    for i in range(k, -1, -1):
        y[i] /= H[i, i]
        for j in range(i - 1, -1, -1):
            y[j] -= H[j, i] * y[i]
    for j in range(k + 1):
        x += y[j] * V[j]


def mfem_gmres(A_mul, prec, b, x, iterative_mode=False, rel_tol=0.0, abs_tol=0.0, max_iter=10, m=50, flexible=False, dot=np.dot):
    """mfem::GMRESSolver::Mult (left-preconditioned) and, with ``flexible``, mfem::FGMRESSolver::Mult (right, flexible):
    modified Gram-Schmidt Arnoldi, Givens rotations, restart every ``m`` (MFEM 4.x linalg/solvers.cpp [3P];
    KrylovSolver.cpp:46-59)."""
    b = _f64(b)
    n = len(b)
    norm = lambda v: np.sqrt(dot(v, v))

    def pz(v):
        z = np.zeros(n); prec(v, z); return z
    if flexible or prec is None:
        if iterative_mode:
            r = b - A_mul(x)
        else:
            x[:] = 0.0; r = b.copy()
    else:
        if iterative_mode:
            r = pz(b - A_mul(x))
        else:
            x[:] = 0.0; r = pz(b)
    beta = norm(r)
    final_norm = max(rel_tol * beta, abs_tol)
    if beta <= final_norm:
        return KrylovResult(1, 0, beta)
    H = np.zeros((m + 1, m)); s = np.zeros(m + 1); cs = np.zeros(m + 1); sn = np.zeros(m + 1)
    V = [None] * (m + 1); Z = [None] * (m + 1)
    j = 1
    while j <= max_iter:
        V[0] = (1.0 / beta) * r
        s[:] = 0.0; s[0] = beta
        i = 0
        while i < m and j <= max_iter:
            if flexible:
                Z[i] = pz(V[i]) if prec is not None else V[i].copy()
                w = A_mul(Z[i])
            elif prec is not None:
                w = pz(A_mul(V[i]))
            else:
                w = A_mul(V[i])
            for k in range(i + 1):
                H[k, i] = dot(w, V[k])
                w = w + (-H[k, i]) * V[k]
            H[i + 1, i] = norm(w)
            V[i + 1] = (1.0 / H[i + 1, i]) * w
            for k in range(i):
                H[k, i], H[k + 1, i] = _app_rot(H[k, i], H[k + 1, i], cs[k], sn[k])
            cs[i], sn[i] = _gen_rot(H[i, i], H[i + 1, i])
            H[i, i], H[i + 1, i] = _app_rot(H[i, i], H[i + 1, i], cs[i], sn[i])
            s[i], s[i + 1] = _app_rot(s[i], s[i + 1], cs[i], sn[i])
            resid = abs(s[i + 1])
            if resid <= final_norm:
                _gmres_update(x, i, H, s, Z if flexible else V)
                return KrylovResult(1, j, resid)
            i += 1; j += 1
        _gmres_update(x, i - 1, H, s, Z if flexible else V)
        r = b - A_mul(x)
        if not flexible and prec is not None:
            r = pz(r)
        beta = norm(r)
        if beta <= final_norm:
            return KrylovResult(1, j, beta)
    return KrylovResult(0, max_iter, beta)


def mfem_bicgstab(A_mul, prec, b, x, iterative_mode=False, rel_tol=0.0, abs_tol=0.0, max_iter=10, dot=np.dot):
    """mfem::BiCGSTABSolver::Mult (MFEM 4.x linalg/solvers.cpp [3P], "p. 27 of the SIAM Templates book"; selected by
    ``KrylovSolver.cpp:39-40``), statement by statement."""
    b = _f64(b)
    if iterative_mode:
        r = b - A_mul(x)
    else:
        x[:] = 0.0; r = b.copy()
    rtilde = r.copy()
    resid = np.sqrt(dot(r, r))
    tol_goal = max(resid * rel_tol, abs_tol)
    if resid <= tol_goal:
        return KrylovResult(1, 0, resid)
    rho_2 = alpha = omega = 1.0
    p = v = None
    for i in range(1, max_iter + 1):
        rho_1 = dot(rtilde, r)
        if rho_1 == 0.0:
            return KrylovResult(0, i, resid)
        if i == 1:
            p = r.copy()
        else:
            beta = (rho_1 / rho_2) * (alpha / omega)
            p = p + (-omega) * v
            p = r + beta * p
        if prec is not None:
            phat = np.zeros_like(p); prec(p, phat)
        else:
            phat = p.copy()
        v = A_mul(phat)
        alpha = rho_1 / dot(rtilde, v)
        s = r + (-alpha) * v
        resid = np.sqrt(dot(s, s))
        if resid < tol_goal:
            x += alpha * phat
            return KrylovResult(1, i, resid)
        if prec is not None:
            shat = np.zeros_like(s); prec(s, shat)
        else:
            shat = s.copy()
        t = A_mul(shat)
        omega = dot(t, s) / dot(t, t)
        x += alpha * phat
        x += omega * shat
        r = s + (-omega) * t
        rho_2 = rho_1
        resid = np.sqrt(dot(r, r))
        if resid < tol_goal:
            return KrylovResult(1, i, resid)
        if omega == 0.0:
            return KrylovResult(0, i, resid)
    return KrylovResult(0, max_iter, resid)


def mfem_minres(A_mul, prec, b, x, iterative_mode=False, rel_tol=0.0, abs_tol=0.0, max_iter=10, dot=np.dot):
    """mfem::MINRESSolver::Mult (MFEM 4.x linalg/solvers.cpp [3P]: van der Vorst, "Iterative Krylov Methods for Large Linear
    Systems", Fig. 6.9, extended to an SPD preconditioner; selected by ``KrylovSolver.cpp:41-42``), statement by statement."""
    b = _f64(b)
    if not iterative_mode:
        v1 = b.copy(); x[:] = 0.0
    else:
        v1 = b - A_mul(x)
    u1 = None
    if prec is not None:
        u1 = np.zeros_like(v1); prec(v1, u1)
    z = lambda: u1 if prec is not None else v1
    eta = beta = np.sqrt(dot(z(), v1))
    gamma0 = gamma1 = 1.0
    sigma0 = sigma1 = 0.0
    norm_goal = max(rel_tol * eta, abs_tol)
    if eta <= norm_goal:
        return KrylovResult(1, 0, abs(eta))
    v0 = np.zeros_like(v1); w0 = np.zeros_like(v1); w1 = np.zeros_like(v1)
    for it in range(1, max_iter + 1):
        m = 1.0 / beta
        v1 = v1 * m
        if prec is not None:
            u1 = u1 * m
        q = A_mul(z())
        alpha = dot(z(), q)
        if it > 1:
            q = q + (-beta) * v0
        v0 = q + (-alpha) * v1
        delta = gamma1 * alpha - gamma0 * sigma1 * beta
        rho3 = sigma0 * beta
        rho2 = sigma1 * alpha + gamma0 * gamma1 * beta
        if prec is None:
            beta = np.sqrt(dot(v0, v0))
        else:
            q = np.zeros_like(v0); prec(v0, q)
            beta = np.sqrt(dot(v0, q))
        rho1 = np.hypot(delta, beta)
        if it == 1:
            w0 = (1.0 / rho1) * z()
        elif it == 2:
            w0 = (1.0 / rho1) * z() + (-rho2 / rho1) * w1
        else:
            w0 = (-rho3 / rho1) * w0 + (-rho2 / rho1) * w1
            w0 = w0 + (1.0 / rho1) * z()
        gamma0 = gamma1
        gamma1 = delta / rho1
        x += (gamma1 * eta) * w0
        sigma0 = sigma1
        sigma1 = beta / rho1
        eta = -sigma1 * eta
        if abs(eta) <= norm_goal:
            return KrylovResult(1, it, abs(eta))
        if prec is not None:
            u1, q = q, u1
        v0, v1 = v1, v0
        w0, w1 = w1, w0
    return KrylovResult(0, max_iter, abs(eta))


def stationary_solver(A_mul, solver, b, x, iterative_mode=False, rel_tol=0.0, abs_tol=0.0, max_iter=1, dot=np.dot):
    """StationarySolver::Mult (src/linalg/solver_ops/ParELAG_StationarySolver.cpp:41-130)."""
    b = _f64(b)
    if iterative_mode:
        r = b - A_mul(x)
    else:
        r = b.copy(); x[:] = 0.0
    norm_k_1 = np.sqrt(dot(r, r))
    its, ratio, converged, norm_k = 0, 1.0, False, norm_k_1
    while its < max_iter:
        its += 1
        c = np.zeros_like(r)
        solver(r, c)
        x += c
        r -= A_mul(c)
        norm_k = np.sqrt(dot(r, r))
        ratio *= norm_k / norm_k_1
        if norm_k < abs_tol or ratio < rel_tol:
            converged = True
            break
        if norm_k / norm_k_1 == 1.0:
            break
        norm_k_1 = norm_k
    return KrylovResult(converged, its, norm_k)


# ---- dense coarse solve and the all-C serial hierarchy ---------------------------------------------

def chol_factor(A_dense):
    Lf = _f64(np.array(A_dense, dtype=np.float64, copy=True))
    err = lib().orc_chol_factor(C.c_int(Lf.shape[0]), _dp(Lf))
    if err:
        raise RuntimeError(f"Cholesky failed at pivot {err - 1}")
    return Lf


def chol_solve(Lf, b, x):
    lib().orc_chol_solve(C.c_int(Lf.shape[0]), _dp(Lf), _dp(_f64(b)), _dp(x))


class CHierarchy:
    """All-C serial V-cycle (orc_hier_*), used for CPU-baseline timing and as a second statement of the
    cycle to cross-check OHierarchy."""

    def __init__(self, As, Ps, smoothers, coarse_chol=None, native=False, hipt=None, omp=False):
        """As: list of csr (diag-first for Chebyshev); Ps: list of csr; smoothers: list of dict(type,sweeps,
        weight,l1,order,max_eig,min_eig,fraction,ds) per level; hipt: optional list of dict(D, Aaux, sm)."""
        self.L = lib(native, omp)
        self.keep = [As, Ps, smoothers, coarse_chol, hipt]
        nl = len(As)
        self.n = As[0].shape[0]
        self.h = C.c_void_p(self.L.orc_hier_create(C.c_int(nl)))

        def sm_args(s):
            return (C.c_int(s["type"]), C.c_int(s.get("sweeps", 1)), C.c_double(s.get("weight", 1.0)),
                    _dp(s.get("l1")), C.c_int(s.get("order", 2)), C.c_double(s.get("max_eig", 0.0)),
                    C.c_double(s.get("min_eig", 0.0)), C.c_double(s.get("fraction", 0.3)), _dp(s.get("ds")))
        for l in range(nl):
            A = As[l]; P = Ps[l] if l < nl - 1 else None
            s = smoothers[l]
            self.L.orc_hier_set_level(self.h, C.c_int(l), C.c_int(A.shape[0]), _ip(A.indptr), _ip(A.indices),
                                      _dp(A.data), C.c_int(P.shape[1] if P is not None else 0),
                                      _ip(P.indptr) if P is not None else None,
                                      _ip(P.indices) if P is not None else None,
                                      _dp(P.data) if P is not None else None, *sm_args(s))
            if hipt is not None and hipt[l] is not None:
                D, X, sa = hipt[l]["D"], hipt[l]["Aaux"], hipt[l]["sm"]
                self.L.orc_hier_set_hiptmair(self.h, C.c_int(l), C.c_int(D.shape[1]), _ip(D.indptr), _ip(D.indices),
                                             _dp(D.data), _ip(X.indptr), _ip(X.indices), _dp(X.data), *sm_args(sa))
        if coarse_chol is not None:
            self.L.orc_hier_set_coarse_chol(self.h, C.c_int(coarse_chol.shape[0]), _dp(coarse_chol))

    def apply(self, rhs, sol, symmetric_post=False):
        self.L.orc_hier_apply(self.h, _dp(_f64(rhs)), _dp(sol), C.c_int(int(symmetric_post)))

    def spmd_bench(self, rhs, workers, seconds):
        """The reference's CPU layout: ``workers`` forked processes, one per core, each owning a row block of every level,
        vectors in shared memory (orc_hier_spmd_bench; l1-Jacobi or "L1 Gauss-Seidel" levels above a dense coarse solve).  Type-2 levels run
        hypre's HYBRID Gauss-Seidel on that many ranks -- each process sweeps its own rows, the others' values frozen, l1 norms of that
        partition (``l1_norms_blocks``) -- i.e. what ``elag_hip_set_gs_hybrid_blocks(workers)`` computes on the device.  Runs V-cycles for
        about ``seconds``; returns (cycles per second, last solution).  Forks: call it from a process that holds no GPU context."""
        As, _Ps, smoothers = self.keep[0], self.keep[1], self.keep[2]
        self._spmd_l1 = []
        for l, sm in enumerate(smoothers[:len(As) - 1]):
            if sm["type"] == 2:
                l1 = l1_norms_blocks(As[l], int(workers), option=2)
                self._spmd_l1.append(l1)
                self.L.orc_hier_set_spmd_l1(self.h, C.c_int(l), _dp(l1))
        sol = np.empty(self.n)
        cycles, elapsed = C.c_long(0), C.c_double(0.0)
        rc = self.L.orc_hier_spmd_bench(self.h, C.c_int(int(workers)), C.c_double(float(seconds)), _dp(_f64(rhs)), _dp(sol),
                                        C.byref(cycles), C.byref(elapsed))
        if rc:
            raise RuntimeError(f"orc_hier_spmd_bench failed with code {rc} (1: bad arguments, 2: not an l1-Jacobi hierarchy, 3: no shared memory)")
        return cycles.value / elapsed.value, sol

    def __del__(self):
        try:
            self.L.orc_hier_destroy(self.h)
        except Exception:
            pass


# ---- multi-rank exchanges (tests only) ---------------------------------------------------------------

class InProcessWorld:
    """All ranks of a partitioned operator inside one process: runs the per-rank oracle pieces in lock step and
    moves halo data by plain copies (checks the partition / comm-package logic without MPI)."""

    def __init__(self, parts):
        self.parts = parts
        self.size = len(parts)

    def matvec(self, x_global, trans=False):
        parts = self.parts
        if not trans:
            xs = [x_global[p["first_col"]:p["first_col"] + p["diag"].shape[1]] for p in parts]
            bufs = [np.asarray(x)[p["send_elmts"]] if len(p["send_elmts"]) else np.empty(0) for x, p in zip(xs, parts)]
            out = []
            for r, p in enumerate(parts):
                ghost = np.empty(p["offd"].shape[1])
                for i, q in enumerate(p["recv_procs"]):
                    pq = parts[q]
                    k = list(pq["send_procs"]).index(r)
                    ghost[p["recv_starts"][i]:p["recv_starts"][i + 1]] = bufs[q][pq["send_starts"][k]:pq["send_starts"][k + 1]]
                y = np.zeros(p["diag"].shape[0])
                csr_matvec(csr(p["diag"]), 1.0, _f64(xs[r]), 0.0, y)
                if p["offd"].shape[1]:
                    csr_matvec(csr(p["offd"]), 1.0, ghost, 1.0, y)
                out.append(y)
            return np.concatenate(out)
        xs = [x_global[p["first_row"]:p["first_row"] + p["diag"].shape[0]] for p in parts]
        ytmp = []
        for x, p in zip(xs, parts):
            t = np.zeros(p["offd"].shape[1])
            if p["offd"].shape[1]:
                csr_matvecT(csr(p["offd"]), 1.0, _f64(x), 0.0, t)
            ytmp.append(t)
        out = []
        for r, p in enumerate(parts):
            y = np.zeros(p["diag"].shape[1])
            csr_matvecT(csr(p["diag"]), 1.0, _f64(xs[r]), 0.0, y)
            for i, q in enumerate(p["send_procs"]):
                pq = parts[q]
                k = list(pq["recv_procs"]).index(r)
                seg = ytmp[q][pq["recv_starts"][k]:pq["recv_starts"][k + 1]]
                el = p["send_elmts"][p["send_starts"][i]:p["send_starts"][i + 1]]
                np.add.at(y, el, seg)
            out.append(y)
        return np.concatenate(out)


class TorchDistExchange:
    """Halo exchange over torch.distributed point-to-point (gloo on CPU): the pattern of
    hypre_ParCSRCommHandleCreate jobs 1 (forward) and 2 (reverse)."""

    def __init__(self, td):
        import torch
        self.td, self.torch = td, torch
        self.rank, self.size = td.get_rank(), td.get_world_size()

    def _xchg(self, send_buf, s_procs, s_starts, r_procs, r_starts):
        torch, td = self.torch, self.td
        out = np.empty(int(r_starts[-1]) if len(r_starts) else 0)
        recv_t = [torch.empty(int(r_starts[i + 1] - r_starts[i]), dtype=torch.float64) for i in range(len(r_procs))]
        reqs = [td.irecv(t, src=int(q)) for t, q in zip(recv_t, r_procs)]
        for i, q in enumerate(s_procs):
            seg = np.ascontiguousarray(send_buf[int(s_starts[i]):int(s_starts[i + 1])])
            reqs.append(td.isend(torch.from_numpy(seg.copy()), dst=int(q)))
        for r in reqs:
            r.wait()
        for i, t in enumerate(recv_t):
            out[int(r_starts[i]):int(r_starts[i + 1])] = t.numpy()
        return out

    def forward(self, pkg, send_buf):
        return self._xchg(send_buf, pkg["send_procs"], pkg["send_starts"], pkg["recv_procs"], pkg["recv_starts"])

    def reverse(self, pkg, ghost_buf):
        return self._xchg(ghost_buf, pkg["recv_procs"], pkg["recv_starts"], pkg["send_procs"], pkg["send_starts"])

    def _allreduce(self, v, op):
        t = self.torch.tensor([float(v)], dtype=self.torch.float64)
        self.td.all_reduce(t, op=op)
        return float(t[0])

    def allreduce_sum(self, v):
        return self._allreduce(v, self.td.ReduceOp.SUM)

    def allreduce_max(self, v):
        return self._allreduce(v, self.td.ReduceOp.MAX)

    def allgatherv(self, local, starts):
        torch, td = self.torch, self.td
        pieces = [torch.empty(int(starts[r + 1] - starts[r]), dtype=torch.float64) for r in range(self.size)]
        mx = max(len(p) for p in pieces)
        pad = [torch.zeros(mx, dtype=torch.float64) for _ in range(self.size)]
        mine = torch.zeros(mx, dtype=torch.float64); mine[:len(local)] = torch.from_numpy(np.asarray(local, dtype=np.float64))
        td.all_gather(pad, mine)
        return np.concatenate([pad[r][:len(pieces[r])].numpy() for r in range(self.size)])


def oparcsr_from_part(part, exch) -> OParCSR:
    pkg = {k: part[k] for k in ("send_procs", "send_starts", "send_elmts", "recv_procs", "recv_starts")}
    dg = part["diag"]
    if dg.shape[0] == dg.shape[1] and part["first_row"] == part["first_col"]:
        dg = diag_first(dg)      # hypre convention: a_ii is the first stored entry of a square diag block (A.0)
    return OParCSR(dg, part["offd"], part["col_map_offd"], pkg, exch, part["first_row"], part["first_col"],
                   part["global_shape"])
