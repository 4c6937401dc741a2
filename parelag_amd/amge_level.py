"""Level data of the (H(div), L2) part of a de Rham sequence, as ``DeRhamSequence::Coarsen`` consumes and produces it
when it is started at ``jformStart = nDim - 1`` (the unstructured-coarsening mode of ``examples/MultigridTestSPE10.cpp:441``).

One ``AmgeLevel`` holds what the reference keeps, for forms 2 and 3 and codimensions 0 (elements) and 1 (facets), in
``AgglomeratedTopology::B_[0]`` / ``facet_bdrAttribute`` (``src/topology/Topology.hpp``), ``DofHandler::entity_dof``
(``src/amge/DofHandler.hpp``), the element / facet mass matrices ``M_[idx]`` in repeated-dof ("rdof") space, ``D_[2]``,
``Targets_`` and ``L2_const_rep_`` (``src/amge/DeRhamSequence.hpp:655-727``).  Dense blocks are stored packed, one block
after the other, row-major, so the same arrays go to the device unchanged.

The two builders at the bottom manufacture finest levels (lowest-order spaces on the generator's tets or on a Cartesian
hex grid); MFEM is not available, so they replace ``DeRhamSequence3D_FE`` (``src/amge/DeRhamSequenceFE.cpp:633-722``).
Conventions of the builders (they differ from MFEM's by per-dof scalings only):
* a 2-form dof is the flux through its facet, a 3-form dof the integral over its element (Whitney forms), so ``D_2`` is
  the signed element-facet incidence, the facet mass is 1/area, the element L2 mass 1/volume;
* boundary facets are oriented outward (MFEM generates a boundary face from its only element, so it is positively
  oriented with respect to it); interior facets keep the generator's orientation;
* the "PV" trace vectors (``InterpolatePV_L2`` / ``InterpolatePV_HdivTraces``, ``DeRhamSequenceFE.cpp:799-860``) are the
  constant 1 and the unit normal flux: volume per element, area per facet.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp


def ptr_from_counts(cnt) -> np.ndarray:
    p = np.zeros(len(cnt) + 1, dtype=np.int64)
    np.cumsum(cnt, out=p[1:])
    return p


@dataclass
class AmgeLevel:
    nE: int
    nF: int
    # topology: element -> facets, signed (+1 = facet orientation points out of the element); row order is meaningful
    ef_ptr: np.ndarray
    ef_idx: np.ndarray
    ef_sgn: np.ndarray
    facet_attr: np.ndarray            # (nF,) 0 = interior facet or no attribute table, k >= 1 = boundary attribute
    # form 3 (all dofs interior to elements)
    ndof3: int
    e_dof3_ptr: np.ndarray
    # form 2: facet dofs are numbered first (facet by facet), element-interior dofs after them (element by element)
    ndof2: int
    f_dof2_ptr: np.ndarray            # (nF+1,) dofs of facet f are f_dof2_ptr[f] .. f_dof2_ptr[f+1]-1
    e_idof2_ptr: np.ndarray           # (nE+1,) interior dofs of element e are nfd + e_idof2_ptr[e] .. (nfd = f_dof2_ptr[nF])
    # packed dense blocks (row-major), block b occupies [ptr[b], ptr[b+1])
    M3e_ptr: np.ndarray
    M3e: np.ndarray                   # element mass of form 3 on the element's dofs
    M2e_ptr: np.ndarray
    M2e: np.ndarray                   # element mass of form 2 on elem_dof2(e) = [dofs of its facets in row order, interior]
    M2f_ptr: np.ndarray
    M2f: np.ndarray                   # facet mass of form 2 on the facet's dofs
    D2: sp.csr_matrix                 # ndof3 x ndof2
    T2: np.ndarray                    # (ndof2, nt2) targets
    T3: np.ndarray                    # (ndof3, nt3)
    const3: np.ndarray                # (ndof3,) representation of the constant 1
    pv2: np.ndarray                   # (ndof2,) PV trace per dof up to the facet's orientation in its agglomerate
    pv3: np.ndarray                   # (ndof3,)
    meta: dict = field(default_factory=dict)

    # ---- derived tables -------------------------------------------------------------------------------------------
    @property
    def nfd2(self) -> int:
        return int(self.f_dof2_ptr[-1])

    def elem_dof3(self, e: int) -> np.ndarray:
        return np.arange(self.e_dof3_ptr[e], self.e_dof3_ptr[e + 1])

    def facet_dof2(self, f: int) -> np.ndarray:
        return np.arange(self.f_dof2_ptr[f], self.f_dof2_ptr[f + 1])

    def elem_dof2(self, e: int) -> np.ndarray:
        """``entity_dof[ELEMENT]`` of form 2: dofs of the element's facets in ``B_[0]`` row order, then its interior
        dofs (``DofHandlerALG::buildElementDofTable``, ``src/amge/DofHandler.cpp:1400-1461``)."""
        fs = self.ef_idx[self.ef_ptr[e]:self.ef_ptr[e + 1]]
        parts = [np.arange(self.f_dof2_ptr[f], self.f_dof2_ptr[f + 1]) for f in fs]
        parts.append(self.nfd2 + np.arange(self.e_idof2_ptr[e], self.e_idof2_ptr[e + 1]))
        return np.concatenate(parts) if parts else np.zeros(0, np.int64)

    def elem_dof2_table(self):
        """(ptr, idx) of ``elem_dof2`` for all elements (threaded host code, ``csrc/host_amge_tables.hip``)."""
        from . import _lib
        L = _lib.load()
        c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        ef_ptr, ef_idx, fp, ip = c64(self.ef_ptr), c64(self.ef_idx), c64(self.f_dof2_ptr), c64(self.e_idof2_ptr)
        vp = lambda a: a.ctypes.data_as(_lib.C.c_void_p)
        ptr = np.empty(self.nE + 1, dtype=np.int64)
        _lib.check(L.elag_hip_host_amge_elem_dof_table(self.nE, vp(ef_ptr), vp(ef_idx), vp(fp), vp(ip), int(self.nfd2), vp(ptr), None))
        idx = np.empty(int(ptr[-1]), dtype=np.int64)
        _lib.check(L.elag_hip_host_amge_elem_dof_table(self.nE, vp(ef_ptr), vp(ef_idx), vp(fp), vp(ip), int(self.nfd2), vp(ptr), vp(idx)))
        return ptr, idx

    def elem_dof2_table_numpy(self):
        """Whole-array statement of :meth:`elem_dof2_table` (the tests compare the two)."""
        nfac = np.diff(self.f_dof2_ptr)
        cnt_f = np.add.reduceat(nfac[self.ef_idx], self.ef_ptr[:-1]) if self.ef_idx.size else np.zeros(self.nE, np.int64)
        cnt_f = np.where(np.diff(self.ef_ptr) > 0, cnt_f, 0)
        cnt_i = np.diff(self.e_idof2_ptr)
        ptr = ptr_from_counts(cnt_f + cnt_i)
        idx = np.empty(ptr[-1], dtype=np.int64)
        # facet part
        rep = nfac[self.ef_idx]
        start = np.repeat(self.f_dof2_ptr[self.ef_idx], rep)
        within = np.arange(rep.sum()) - np.repeat(ptr_from_counts(rep)[:-1], rep)
        fpart = start + within
        e_of_slot = np.repeat(np.arange(self.nE), np.diff(self.ef_ptr))
        e_of_fd = np.repeat(e_of_slot, rep)
        pos_f = ptr[e_of_fd] + (np.arange(fpart.size) - np.repeat(ptr_from_counts(cnt_f)[:-1], cnt_f))
        idx[pos_f] = fpart
        e_of_i = np.repeat(np.arange(self.nE), cnt_i)
        pos_i = ptr[e_of_i] + cnt_f[e_of_i] + (np.arange(cnt_i.sum()) - np.repeat(self.e_idof2_ptr[:-1], cnt_i))
        idx[pos_i] = self.nfd2 + np.arange(cnt_i.sum())
        return ptr, idx

    def block(self, which: str, b: int) -> np.ndarray:
        ptr, val = {"M3e": (self.M3e_ptr, self.M3e), "M2e": (self.M2e_ptr, self.M2e), "M2f": (self.M2f_ptr, self.M2f)}[which]
        v = val[ptr[b]:ptr[b + 1]]
        n = int(round(np.sqrt(v.size)))
        return v.reshape(n, n)

    # ---- assembled operators (``ComputeMassOperator`` = rdof -> dof assembly, ``DeRhamSequence.cpp:1082-1116``) --------
    def mass3(self) -> sp.csr_matrix:
        n3 = np.diff(self.e_dof3_ptr)
        rows, cols = _block_rc(self.e_dof3_ptr[:-1], n3)
        M = sp.csr_matrix((self.M3e, (rows, cols)), shape=(self.ndof3, self.ndof3))
        M.sum_duplicates(); M.sort_indices()
        return M

    def mass2(self) -> sp.csr_matrix:
        ptr, idx = self.elem_dof2_table()
        n2 = np.diff(ptr)
        lr, lc = _block_rc(ptr[:-1], n2)
        M = sp.csr_matrix((self.M2e, (idx[lr], idx[lc])), shape=(self.ndof2, self.ndof2))
        M.sum_duplicates(); M.sort_indices()
        return M

    def darcy_blocks(self):
        """``M``, ``B = W D``, ``Bt`` of the mixed system (``examples/MultigridTestDarcy.cpp:417-426``)."""
        M = self.mass2()
        B = (self.mass3() @ self.D2).tocsr()
        B.sort_indices()
        Bt = B.T.tocsr()
        Bt.sort_indices()
        return M, B, Bt


def _block_rc(starts, sizes):
    """Row / column index of every entry of consecutive dense row-major blocks of order ``sizes[b]`` whose index ranges
    start at ``starts[b]``."""
    sizes = np.asarray(sizes, dtype=np.int64)
    sq = sizes * sizes
    b_of = np.repeat(np.arange(sizes.size), sq)
    k = np.arange(sq.sum()) - np.repeat(ptr_from_counts(sq)[:-1], sq)
    n = sizes[b_of]
    return starts[b_of] + k // np.maximum(n, 1), starts[b_of] + k % np.maximum(n, 1)


# ---- finest levels ------------------------------------------------------------------------------------------------------

def _unit_cube_attr(centroid: np.ndarray, is_bdr: np.ndarray) -> np.ndarray:
    """Boundary attribute 1..6 of a facet of the unit cube from its centroid (x=0,x=1,y=0,y=1,z=0,z=1); the six
    attributes of ``meshes/cube456.mesh`` are the six sides of the cube as well."""
    attr = np.zeros(centroid.shape[0], dtype=np.int32)
    lo, hi = centroid.min(axis=0), centroid.max(axis=0)
    for d in range(3):
        attr[is_bdr & (np.abs(centroid[:, d] - lo[d]) < 1e-12)] = 2 * d + 1
        attr[is_bdr & (np.abs(centroid[:, d] - hi[d]) < 1e-12)] = 2 * d + 2
    assert np.all(attr[is_bdr] > 0)
    return attr


def level_from_tets(mesh, kinv: np.ndarray | None = None, bdr_attr: bool = True) -> AmgeLevel:
    """Lowest-order (RT0, P0) finest level on a ``meshgen.TetMesh``.  ``meta`` carries ``S2`` / ``S3`` (+-1 per dof) with
    ``M2' = S2 M2 S2``, ``D2' = S3 D2 S2`` relating this level's dofs to the generator's operators."""
    mesh.build_faces()
    T, F = mesh.nt, mesh.nf
    x = mesh.vertices[mesh.stets]                                   # (T,4,3)
    _g, vol, _sg = mesh._grads()
    sig = mesh._face_sign()                                         # +1: sorted-orientation normal points out of the tet
    cnt = np.bincount(mesh.tet_faces.reshape(-1), minlength=F)
    is_bdr = cnt == 1
    o = np.ones(F)
    tf = mesh.tet_faces
    bslot = is_bdr[tf]
    o[tf[bslot]] = sig[bslot]                                       # boundary facets: outward
    s = sig * o[tf]                                                 # element-facet sign
    sgn_t = sig * np.array([1.0, -1.0, 1.0, -1.0])[None, :]
    assert np.all(sgn_t == sgn_t[:, :1])
    S3 = sgn_t[:, 0].copy()
    I = (np.ones((4, 4)) + np.eye(4)) / 20.0
    d = x[:, None, :, :] - x[:, :, None, :]
    loc = np.einsum("ij,tkid,tljd->tkl", I, d, d) / (9.0 * vol)[:, None, None]
    loc *= s[:, :, None] * s[:, None, :]
    if kinv is not None:
        loc *= np.asarray(kinv, dtype=np.float64)[:, None, None]
    fv = mesh.vertices[mesh.faces]                                  # (F,3,3)
    nrm = 0.5 * np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])  # area-weighted normal, sorted orientation
    area = np.linalg.norm(nrm, axis=1)
    ef_ptr = 4 * np.arange(T + 1, dtype=np.int64)
    D2 = sp.csr_matrix((s.reshape(-1), tf.reshape(-1).astype(np.int64), ef_ptr), shape=(T, F))
    attr = _unit_cube_attr(fv.mean(axis=1), is_bdr) if bdr_attr else np.zeros(F, np.int32)
    return AmgeLevel(
        nE=T, nF=F, ef_ptr=ef_ptr, ef_idx=tf.reshape(-1).astype(np.int64), ef_sgn=s.reshape(-1).copy(), facet_attr=attr,
        ndof3=T, e_dof3_ptr=np.arange(T + 1, dtype=np.int64),
        ndof2=F, f_dof2_ptr=np.arange(F + 1, dtype=np.int64), e_idof2_ptr=np.zeros(T + 1, dtype=np.int64),
        M3e_ptr=np.arange(T + 1, dtype=np.int64), M3e=1.0 / vol,
        M2e_ptr=16 * np.arange(T + 1, dtype=np.int64), M2e=np.ascontiguousarray(loc).reshape(-1),
        M2f_ptr=np.arange(F + 1, dtype=np.int64), M2f=1.0 / area,
        D2=D2, T2=nrm * o[:, None], T3=vol[:, None].copy(), const3=vol.copy(), pv2=area.copy(), pv3=vol.copy(),
        meta=dict(S2=o, S3=S3, kind="tet"))


def level_from_hex(n, h=(1.0, 1.0, 1.0), kinv: np.ndarray | None = None, bdr_attr: bool = True) -> AmgeLevel:
    """Lowest-order (RT0, Q0) finest level on an ``n[0] x n[1] x n[2]`` Cartesian grid of boxes of size ``h`` -- the
    element type of ``examples/MultigridTestSPE10.cpp:194-208`` (60 x 220 x 85 cells of 20 x 10 x 2 ft)."""
    nx, ny, nz = (int(v) for v in n)
    hx, hy, hz = (float(v) for v in h)
    nE = nx * ny * nz
    I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    I, J, K = I.ravel(), J.ravel(), K.ravel()
    nFx, nFy, nFz = (nx + 1) * ny * nz, nx * (ny + 1) * nz, nx * ny * (nz + 1)
    fx = lambda i, j, k: (i * ny + j) * nz + k
    fy = lambda i, j, k: nFx + (i * (ny + 1) + j) * nz + k
    fz = lambda i, j, k: nFx + nFy + (i * ny + j) * (nz + 1) + k
    F = nFx + nFy + nFz
    ef = np.stack([fx(I, J, K), fx(I + 1, J, K), fy(I, J, K), fy(I, J + 1, K), fz(I, J, K), fz(I, J, K + 1)], axis=1)
    lowb = np.stack([I == 0, np.zeros(nE, bool), J == 0, np.zeros(nE, bool), K == 0, np.zeros(nE, bool)], axis=1)
    s = np.tile(np.array([-1.0, 1.0, -1.0, 1.0, -1.0, 1.0]), (nE, 1))
    s[lowb] = 1.0                                                   # low-side boundary facets are oriented outward
    o = np.ones(F)
    o[ef[lowb]] = -1.0                                              # ... i.e. against the axis
    area = np.concatenate([np.full(nFx, hy * hz), np.full(nFy, hx * hz), np.full(nFz, hx * hy)])
    vol = np.full(nE, hx * hy * hz)
    pair = np.array([[1.0 / 3, 1.0 / 6], [1.0 / 6, 1.0 / 3]])
    loc = np.zeros((nE, 6, 6))
    for dax, (hl, a) in enumerate(((hx, hy * hz), (hy, hx * hz), (hz, hx * hy))):
        loc[:, 2 * dax:2 * dax + 2, 2 * dax:2 * dax + 2] = pair * hl / a
    oo = o[ef]
    loc *= oo[:, :, None] * oo[:, None, :]
    if kinv is not None:
        loc *= np.asarray(kinv, dtype=np.float64)[:, None, None]
    T2 = np.zeros((F, 3))
    T2[:nFx, 0] = hy * hz
    T2[nFx:nFx + nFy, 1] = hx * hz
    T2[nFx + nFy:, 2] = hx * hy
    T2 *= o[:, None]
    ef_ptr = 6 * np.arange(nE + 1, dtype=np.int64)
    D2 = sp.csr_matrix((s.reshape(-1), ef.reshape(-1).astype(np.int64), ef_ptr), shape=(nE, F))
    attr = np.zeros(F, np.int32)
    if bdr_attr:
        attr[fx(0 * J, J, K)] = 1; attr[fx(0 * J + nx, J, K)] = 2
        attr[fy(I, 0 * I, K)] = 3; attr[fy(I, 0 * I + ny, K)] = 4
        attr[fz(I, J, 0 * I)] = 5; attr[fz(I, J, 0 * I + nz)] = 6
    return AmgeLevel(
        nE=nE, nF=F, ef_ptr=ef_ptr, ef_idx=ef.reshape(-1).astype(np.int64), ef_sgn=s.reshape(-1).copy(), facet_attr=attr,
        ndof3=nE, e_dof3_ptr=np.arange(nE + 1, dtype=np.int64),
        ndof2=F, f_dof2_ptr=np.arange(F + 1, dtype=np.int64), e_idof2_ptr=np.zeros(nE + 1, dtype=np.int64),
        M3e_ptr=np.arange(nE + 1, dtype=np.int64), M3e=1.0 / vol,
        M2e_ptr=36 * np.arange(nE + 1, dtype=np.int64), M2e=np.ascontiguousarray(loc).reshape(-1),
        M2f_ptr=np.arange(F + 1, dtype=np.int64), M2f=1.0 / area,
        D2=D2, T2=T2, T3=vol[:, None].copy(), const3=vol.copy(), pv2=area.copy(), pv3=vol.copy(),
        meta=dict(S2=o, kind="hex", n=(nx, ny, nz), h=(hx, hy, hz)))


def box_partition(n, b) -> np.ndarray:
    """Element -> agglomerate map of a Cartesian grid cut into boxes of ``b[0] x b[1] x b[2]`` cells (the last box of a
    direction is smaller when the size does not divide)."""
    nx, ny, nz = n
    bi, bj, bk = (np.arange(m) // bb for m, bb in zip(n, b))
    nbj, nbk = bj.max() + 1, bk.max() + 1
    P = (bi[:, None, None] * nbj + bj[None, :, None]) * nbk + bk[None, None, :]
    return P.reshape(-1).astype(np.int32)
