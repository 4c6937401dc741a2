"""Level data of a whole 3-D de Rham sequence (forms ``jstart .. 3``) as ``DeRhamSequence::Coarsen`` consumes and produces
it (``src/amge/DeRhamSequence.hpp:655-727``): the agglomerated topology's boundary operators ``B_[0..2]``
(``src/topology/Topology.hpp``), one dof handler per form (interior dofs per entity, numbered peak dofs first, then ridge,
facet and element dofs: ``DofHandlerALG::computeOffset``, ``src/amge/DofHandler.cpp:1058-1186``), the mass matrices
``M_(form, codim)`` entity by entity in repeated-dof space, the derivatives ``D_[j]``, targets and PV base vectors.

``AmgeLevel`` (``amge_level.py``) is the two-form special case the device path consumes today; ``SeqLevel`` is the general
container used by the oracle of the full ``Coarsen`` (``oracle/derham.py``).  The builder at the bottom manufactures the
lowest-order finest level on the generator's tets (MFEM is not available: it replaces ``DeRhamSequence3D_FE``,
``src/amge/DeRhamSequenceFE.cpp:633-722``) with the conventions of ``amge_level.level_from_tets`` (Whitney forms: a dof is the
integral of the form over its entity; boundary facets oriented outward).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .amge_level import _block_rc, _unit_cube_attr, ptr_from_counts

NFORMS = 4


def bool_mult(a_ptr, a_idx, b_ptr, b_idx):
    """``BoolMult`` / ``mfem::Mult`` pattern product with the columns of a row in first-touch order."""
    out_ptr, out_idx = [0], []
    for r in range(len(a_ptr) - 1):
        seen = {}
        for k in a_idx[a_ptr[r]:a_ptr[r + 1]]:
            for c in b_idx[b_ptr[k]:b_ptr[k + 1]]:
                seen.setdefault(int(c), None)
        out_idx.extend(seen.keys())
        out_ptr.append(len(out_idx))
    return np.array(out_ptr, dtype=np.int64), np.array(out_idx, dtype=np.int64)


def _expand_ranges(starts, counts):
    from . import amge
    return amge.expand_ranges(starts, counts)


def _product(a_ptr, a_idx, b_ptr, b_idx, a_sgn=None, b_sgn=None):
    """Row-by-row product of two ragged tables, columns in first-touch order (``mfem::Mult``).  With signs: the signed sum
    per column, cancelled entries dropped, reduced to the sign (``MultOrientation``, ``TopologyTable.cpp:130-139``).  Native
    host code (``csrc/host_tables.hip``)."""
    from . import _lib
    from .amge import _ptr
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    a_ptr, a_idx, b_ptr, b_idx = c64(a_ptr), c64(a_idx), c64(b_ptr), c64(b_idx)
    nrow, nmid = len(a_ptr) - 1, len(b_ptr) - 1
    if a_idx.size and (a_idx.min() < 0 or a_idx.max() >= nmid):
        raise ValueError("table product: an entry of the left table is outside the right table")
    cap = int(np.diff(b_ptr)[a_idx].sum()) if a_idx.size else 0
    ncol = int(b_idx.max()) + 1 if b_idx.size else 1
    out_ptr, out_idx = np.empty(nrow + 1, dtype=np.int64), np.empty(cap, dtype=np.int64)
    signed = a_sgn is not None
    if signed:
        a_sgn, b_sgn = np.ascontiguousarray(a_sgn, dtype=np.float64), np.ascontiguousarray(b_sgn, dtype=np.float64)
        out_sgn = np.empty(cap)
    nnz = np.zeros(1, dtype=np.int64)
    _lib.check(_lib.load().elag_hip_host_table_product(nrow, _ptr(a_ptr), _ptr(a_idx), _ptr(a_sgn) if signed else None, nmid, _ptr(b_ptr), _ptr(b_idx),
                                                   _ptr(b_sgn) if signed else None, ncol, _ptr(out_ptr), _ptr(out_idx),
                                                   _ptr(out_sgn) if signed else None, _ptr(nnz)))
    n = int(nnz[0])
    return out_ptr, out_idx[:n].copy(), (out_sgn[:n].copy() if signed else np.ones(n))


def entity_dof_table(nrow, layers, own_off, own_base):
    """``entity_dof`` rows: for every layer ((ptr, idx) of the rows' sub-entities, the sub-entities' interior dof offsets, their base) the dofs in
    connectivity order, then the rows' own interior dofs.  Native host code (``csrc/host_tables.hip``, ``elag_hip_host_entity_dof_table``); round 5 -- the
    whole-array numpy form of it (ranges expanded, counted with bincount over repeat, rows concatenated) was the largest single piece of a coarsening."""
    import ctypes as C
    from . import _lib
    from .amge import _ptr
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
    keep = [(c64(p), c64(i), c64(off)) for (p, i), off, _ in layers]
    nl = len(keep)
    arr = lambda k: (C.c_void_p * max(nl, 1))(*[t[k].ctypes.data for t in keep]) if nl else None
    cp, ci, do = arr(0), arr(1), arr(2)
    base = c64([b for _, _, b in layers]) if nl else np.zeros(1, dtype=np.int64)
    own = c64(own_off)
    if len(own) != nrow + 1 or any(len(t[0]) != nrow + 1 for t in keep):
        raise ValueError("entity_dof_table: a table does not have one row per entity")
    for p_, i_, off_ in keep:
        if i_.size and (i_.min() < 0 or i_.max() + 1 >= len(off_)):
            raise ValueError("entity_dof_table: a connectivity entry is outside its layer's dof offsets")
    ptr = np.empty(nrow + 1, dtype=np.int64)
    lib = _lib.load()
    _lib.check(lib.elag_hip_host_entity_dof_table(nrow, nl, cp, ci, do, _ptr(base), _ptr(own), int(own_base), _ptr(ptr), None))
    idx = np.empty(int(ptr[-1]), dtype=np.int64)
    _lib.check(lib.elag_hip_host_entity_dof_table(nrow, nl, cp, ci, do, _ptr(base), _ptr(own), int(own_base), _ptr(ptr), _ptr(idx)))
    return ptr, idx


def _concat_rows(tables, nrow):
    """Row-wise concatenation of ragged tables [(ptr, idx), ...] in the given order."""
    expand_ranges = _expand_ranges
    cnts = [np.diff(p) for p, _ in tables]
    ptr = ptr_from_counts(sum(cnts))
    out = np.empty(int(ptr[-1]), dtype=np.int64)
    off = ptr[:-1].copy()
    for (p, i), cn in zip(tables, cnts):
        pos, own = expand_ranges(off, cn)
        out[pos] = i
        off = off + cn
    return ptr, out


class _Tables:
    """Vectorised ``entity_dof`` tables of a level and the agglomerated rows built from them."""

    def __init__(self, lv: SeqLevel):
        self.lv = lv
        self._conn, self._edof = {}, {}

    def conn(self, c, c2):
        if (c, c2) not in self._conn:
            B = self.lv.B
            if c2 == c + 1:
                self._conn[(c, c2)] = (B[c][0], B[c][1])
            elif (c, c2) == (0, 3):
                p, i = self.conn(0, 2)
                self._conn[(c, c2)] = _product(p, i, B[2][0], B[2][1])[:2]
            else:
                self._conn[(c, c2)] = _product(B[c][0], B[c][1], B[c + 1][0], B[c + 1][1])[:2]
        return self._conn[(c, c2)]

    def edof(self, j, c):
        """(ptr, idx) of ``entity_dof[c]`` of form j: peaks', ridges', facets' dofs in connectivity order, then interior."""
        if (j, c) not in self._edof:
            lv = self.lv
            cb = lv.cb(j)
            layers = [(self.conn(c, c2), lv.idof[j][c2], lv.base(j, c2)) for c2 in range(3, c, -1) if c2 <= cb]
            self._edof[(j, c)] = entity_dof_table(lv.n[c], layers, lv.idof[j][c], lv.base(j, c))
        return self._edof[(j, c)]


@dataclass
class SeqLevel:
    n: list                      # entity counts by codimension: elements, facets, ridges, peaks
    B: list                      # B[c] = (ptr, idx, sgn): codim c -> c+1, c = 0, 1, 2 (row order is meaningful)
    facet_attr: np.ndarray
    jstart: int
    idof: list                   # idof[j][c] = (n[c]+1,) offsets of the interior dofs of form j on codim-c entities (c <= 3-j)
    M: list                      # M[j][c] = (ptr, vals): packed row-major blocks over entity_dof(j, c, e)
    D: list                      # D[j]: csr ndof[j+1] x ndof[j], j = jstart .. 2
    T: list                      # T[j]: (ndof[j], nT_j)
    pv: list                     # pv[j]: (ndof[j],) PV base values (times the entity's orientation in its agglomerate)
    const3: np.ndarray
    meta: dict = field(default_factory=dict)
    _conn: dict = field(default_factory=dict)
    _edof: dict = field(default_factory=dict)

    def cb(self, j):
        return 3 - j

    def base(self, j, c):
        """First dof of form j living on codim-c entities (higher codimensions are numbered first)."""
        return int(sum(self.idof[j][c2][-1] for c2 in range(self.cb(j), c, -1)))

    def ndof(self, j):
        return int(sum(self.idof[j][c][-1] for c in range(self.cb(j) + 1)))

    def conn(self, c, c2):
        """``GetConnectivity(c, c2)`` (``src/topology/Topology.cpp:240-268``)."""
        key = (c, c2)
        if key not in self._conn:
            if c2 == c + 1:
                self._conn[key] = (self.B[c][0], self.B[c][1])
            elif (c, c2) == (0, 3):
                p, i = self.conn(0, 2)
                self._conn[key] = bool_mult(p, i, self.B[2][0], self.B[2][1])
            else:
                self._conn[key] = bool_mult(self.B[c][0], self.B[c][1], self.B[c + 1][0], self.B[c + 1][1])
        return self._conn[key]

    def interior_dofs(self, j, c, e):
        return self.base(j, c) + np.arange(self.idof[j][c][e], self.idof[j][c][e + 1])

    def entity_dof(self, j, c):
        """``entity_dof[c]`` of form j as a list of arrays: dofs of the entity's peaks, ridges, facets (connectivity order),
        then its interior dofs (``DofHandlerALG::build*DofTable``, ``src/amge/DofHandler.cpp:1188-1461``)."""
        key = (j, c)
        if key not in self._edof:
            rows = []
            cb = self.cb(j)
            subs = [(c2, self.conn(c, c2)) for c2 in range(3, c, -1) if c2 <= cb]
            bases = {c2: self.base(j, c2) for c2 in range(cb + 1)}
            for e in range(self.n[c]):
                parts = []
                for c2, (p, i) in subs:
                    ip = self.idof[j][c2]
                    for s in i[p[e]:p[e + 1]]:
                        parts.append(bases[c2] + np.arange(ip[s], ip[s + 1]))
                parts.append(bases[c] + np.arange(self.idof[j][c][e], self.idof[j][c][e + 1]))
                rows.append(np.concatenate(parts).astype(np.int64))
            self._edof[key] = rows
        return self._edof[key]

    def block(self, j, c, e):
        ptr, val = self.M[j][c]
        v = val[ptr[e]:ptr[e + 1]]
        k = int(round(np.sqrt(v.size)))
        return v.reshape(k, k)

    def mass(self, j, c=0, scaling=None):
        """Assembled mass of form j from the codim-c blocks (``ComputeMassOperator`` for c = 0); ``scaling``: one factor per
        entity applied to its block first (``ComputeMassOperator(jform, elemMatrixScaling)``, ``DeRhamSequence.cpp:1325-1370``)."""
        ep, ei = _Tables(self).edof(j, c)
        lr, lc = _block_rc(ep[:-1], np.diff(ep))
        vals = self.M[j][c][1]
        if scaling is not None:
            scaling = np.asarray(scaling, dtype=np.float64)
            if scaling.shape != (self.n[c],):
                raise ValueError(f"elemMatrixScaling has the wrong size. Size is {scaling.size}. Should be {self.n[c]}.")
            vals = vals * np.repeat(scaling, np.diff(self.M[j][c][0]))
        M = sp.csr_matrix((vals, (ei[lr], ei[lc])), shape=(self.ndof(j), self.ndof(j)))
        M.sum_duplicates(); M.sort_indices()
        return M


# ---- finest level on tets --------------------------------------------------------------------------------------------------

def _reorder_blocks(loc, have, want):
    """``loc`` (E,k,k) is indexed like the id lists ``have`` (E,k); return it indexed like ``want`` (E,k) (same ids per row)."""
    oh = np.argsort(have, axis=1); ow = np.argsort(want, axis=1)
    perm = np.empty_like(oh)
    np.put_along_axis(perm, ow, oh, axis=1)          # want[:, q] == have[:, perm[:, q]]
    assert np.array_equal(np.take_along_axis(have, perm, axis=1), want)
    return loc[np.arange(loc.shape[0])[:, None, None], perm[:, :, None], perm[:, None, :]]


def seq_level_from_tets(mesh, jstart: int = 0, kinv: np.ndarray | None = None, bdr_attr: bool = True) -> SeqLevel:
    from .meshgen import _LE
    mesh.build_faces()
    T, F, E, V = mesh.nt, mesh.nf, mesh.ne, mesh.nv
    x = mesh.vertices[mesh.stets]
    g, vol, _sg = mesh._grads()
    sig = mesh._face_sign()
    tf = mesh.tet_faces
    is_bdr = np.bincount(tf.reshape(-1), minlength=F) == 1
    o = np.ones(F)
    bslot = is_bdr[tf]
    o[tf[bslot]] = sig[bslot]
    s = sig * o[tf]
    S3 = (sig * np.array([1.0, -1.0, 1.0, -1.0])[None, :])[:, 0].copy()
    # topology
    B0 = (4 * np.arange(T + 1, dtype=np.int64), tf.reshape(-1).astype(np.int64), s.reshape(-1).copy())
    f = mesh.faces
    e_bc, e_ac, e_ab = mesh._edge_ids(f[:, [1, 2]]), mesh._edge_ids(f[:, [0, 2]]), mesh._edge_ids(f[:, [0, 1]])
    fe = np.stack([e_bc, e_ac, e_ab], 1).astype(np.int64)
    B1 = (3 * np.arange(F + 1, dtype=np.int64), fe.reshape(-1), (np.array([1.0, -1.0, 1.0])[None, :] * o[:, None]).reshape(-1))
    B2 = (2 * np.arange(E + 1, dtype=np.int64), mesh.edges.reshape(-1).astype(np.int64), np.tile(np.array([-1.0, 1.0]), E))
    fv = mesh.vertices[f]
    nrm = 0.5 * np.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0])
    area = np.linalg.norm(nrm, axis=1)
    ev = mesh.vertices[mesh.edges[:, 1]] - mesh.vertices[mesh.edges[:, 0]]
    elen = np.linalg.norm(ev, axis=1)
    attr = _unit_cube_attr(fv.mean(axis=1), is_bdr) if bdr_attr else np.zeros(F, np.int32)
    ones = lambda k: np.arange(k + 1, dtype=np.int64)
    zeros = lambda k: np.zeros(k + 1, dtype=np.int64)
    n = [T, F, E, V]
    idof = [[zeros(T), zeros(F), zeros(E), ones(V)], [zeros(T), zeros(F), ones(E)], [zeros(T), ones(F)], [ones(T)]]
    lv = SeqLevel(n=n, B=[B0, B1, B2], facet_attr=attr, jstart=jstart, idof=idof, M=[None] * 4, D=[None] * 3, T=[None] * 4, pv=[None] * 4,
                  const3=vol.copy(), meta=dict(S2=o, S3=S3, kind="tet"))
    I4 = (np.ones((4, 4)) + np.eye(4)) / 20.0
    # ---- form 3 / 2 (as amge_level.level_from_tets)
    d = x[:, None, :, :] - x[:, :, None, :]
    loc2 = np.einsum("ij,tkid,tljd->tkl", I4, d, d) / (9.0 * vol)[:, None, None] * (s[:, :, None] * s[:, None, :])
    if kinv is not None:
        loc2 = loc2 * np.asarray(kinv, dtype=np.float64)[:, None, None]
    lv.M[3] = [(ones(T), 1.0 / vol)]
    lv.M[2] = [(16 * ones(T), np.ascontiguousarray(loc2).reshape(-1)), (ones(F), 1.0 / area)]
    lv.D[2] = sp.csr_matrix((s.reshape(-1), tf.reshape(-1).astype(np.int64), B0[0]), shape=(T, F))
    lv.T[3] = vol[:, None].copy(); lv.pv[3] = vol.copy()
    lv.T[2] = nrm * o[:, None]; lv.pv[2] = area.copy()
    if jstart <= 1:
        # ---- form 1: Whitney 1-forms, dof = tangential integral along the edge (low -> high vertex)
        G = np.einsum("tid,tjd->tij", g, g)
        i_, j_ = _LE[:, 0], _LE[:, 1]
        ik = I4[np.ix_(i_, i_)]; il = I4[np.ix_(i_, j_)]; jk = I4[np.ix_(j_, i_)]; jl = I4[np.ix_(j_, j_)]
        loc1 = (ik[None] * G[:, j_][:, :, j_] - il[None] * G[:, j_][:, :, i_] - jk[None] * G[:, i_][:, :, j_] + jl[None] * G[:, i_][:, :, i_]) * vol[:, None, None]
        p02, i02 = _Tables(lv).conn(0, 2)
        assert np.all(np.diff(p02) == 6)
        loc1 = _reorder_blocks(loc1, mesh.tet_edges, i02.reshape(T, 6))
        # facet mass of the tangential trace: 2-D Whitney 1-forms on the triangle (ND_3D_FacetMassIntegrator)
        a, b, c = fv[:, 0], fv[:, 1], fv[:, 2]
        nn = np.cross(b - a, c - a); n2 = np.einsum("fd,fd->f", nn, nn)
        gl = np.stack([np.cross(nn, c - b), np.cross(nn, a - c), np.cross(nn, b - a)], axis=1) / n2[:, None, None]   # surface grad lambda_i
        Gf = np.einsum("fid,fjd->fij", gl, gl)
        I3 = (np.ones((3, 3)) + np.eye(3)) / 12.0
        le = np.array([[1, 2], [0, 2], [0, 1]])          # local edges in B1 row order: bc, ac, ab
        fi, fj = le[:, 0], le[:, 1]
        ik = I3[np.ix_(fi, fi)]; il = I3[np.ix_(fi, fj)]; jk = I3[np.ix_(fj, fi)]; jl = I3[np.ix_(fj, fj)]
        loc1f = (ik[None] * Gf[:, fj][:, :, fj] - il[None] * Gf[:, fj][:, :, fi] - jk[None] * Gf[:, fi][:, :, fj] + jl[None] * Gf[:, fi][:, :, fi]) * area[:, None, None]
        lv.M[1] = [(36 * ones(T), np.ascontiguousarray(loc1).reshape(-1)), (9 * ones(F), np.ascontiguousarray(loc1f).reshape(-1)), (ones(E), 1.0 / elen)]
        lv.D[1] = sp.csr_matrix((B1[2], B1[1], B1[0]), shape=(F, E))
        lv.T[1] = ev.copy(); lv.pv[1] = elen.copy()
    if jstart <= 0:
        p03, i03 = _Tables(lv).conn(0, 3)
        assert np.all(np.diff(p03) == 4)
        loc0 = np.broadcast_to(I4[None], (T, 4, 4)) * vol[:, None, None]
        loc0 = _reorder_blocks(loc0, mesh.stets, i03.reshape(T, 4))      # (symmetric under permutation, kept for clarity)
        I3 = (np.ones((3, 3)) + np.eye(3)) / 12.0
        loc0f = np.broadcast_to(I3[None], (F, 3, 3)) * area[:, None, None]
        loc0r = np.broadcast_to(np.array([[2.0, 1.0], [1.0, 2.0]])[None] / 6.0, (E, 2, 2)) * elen[:, None, None]
        lv.M[0] = [(16 * ones(T), np.ascontiguousarray(loc0).reshape(-1)), (9 * ones(F), np.ascontiguousarray(loc0f).reshape(-1)),
                   (4 * ones(E), np.ascontiguousarray(loc0r).reshape(-1)), (ones(V), np.ones(V))]
        lv.D[0] = sp.csr_matrix((B2[2], B2[1], B2[0]), shape=(E, V))
        lv.T[0] = np.concatenate([np.ones((V, 1)), mesh.vertices], axis=1); lv.pv[0] = np.ones(V)
    return lv


# ---- finest level on a Cartesian hex grid -----------------------------------------------------------------------------------

def seq_level_from_hex(n, h=(1.0, 1.0, 1.0), jstart: int = 0, kinv: np.ndarray | None = None, bdr_attr: bool = True) -> SeqLevel:
    """Lowest-order sequence (Q1, Nedelec, Raviart-Thomas, Q0 on boxes; dofs = integrals of the forms over vertices / edges /
    faces / cells) on an ``n[0] x n[1] x n[2]`` Cartesian grid: the element type of ``examples/MultigridTestSPE10.cpp:194-208``
    for every form (``amge_level.level_from_hex`` holds forms 2 and 3).  Edges point along their axis, faces along theirs
    except the low-side boundary faces (outward), the face -> edge table carries the right-hand circulation."""
    from .amge_level import level_from_hex
    nx, ny, nz = (int(v) for v in n)
    hx, hy, hz = (float(v) for v in h)
    l2 = level_from_hex(n, h, kinv=kinv, bdr_attr=bdr_attr)
    o = l2.meta["S2"]
    nE, nF = l2.nE, l2.nF
    I, J, K = (g.ravel() for g in np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"))
    vid = lambda i, j, k: (i * (ny + 1) + j) * (nz + 1) + k
    nV = (nx + 1) * (ny + 1) * (nz + 1)
    nEx, nEy, nEz = nx * (ny + 1) * (nz + 1), (nx + 1) * ny * (nz + 1), (nx + 1) * (ny + 1) * nz
    ex = lambda i, j, k: (i * (ny + 1) + j) * (nz + 1) + k
    ey = lambda i, j, k: nEx + (i * ny + j) * (nz + 1) + k
    ez = lambda i, j, k: nEx + nEy + (i * (ny + 1) + j) * nz + k
    nR = nEx + nEy + nEz
    # edges -> vertices (tail, head)
    def grid(a, b, c):
        g = np.meshgrid(np.arange(a), np.arange(b), np.arange(c), indexing="ij")
        return tuple(q.ravel() for q in g)
    xi, xj, xk = grid(nx, ny + 1, nz + 1); yi, yj, yk = grid(nx + 1, ny, nz + 1); zi, zj, zk = grid(nx + 1, ny + 1, nz)
    ev = np.concatenate([np.stack([vid(xi, xj, xk), vid(xi + 1, xj, xk)], 1), np.stack([vid(yi, yj, yk), vid(yi, yj + 1, yk)], 1),
                         np.stack([vid(zi, zj, zk), vid(zi, zj, zk + 1)], 1)]).astype(np.int64)
    elen = np.concatenate([np.full(nEx, hx), np.full(nEy, hy), np.full(nEz, hz)])
    B2 = (2 * np.arange(nR + 1, dtype=np.int64), ev.reshape(-1), np.tile(np.array([-1.0, 1.0]), nR))
    # faces -> edges with the right-hand circulation (numbering of the faces as in level_from_hex: x-, y-, z-faces)
    fi, fj, fk = grid(nx + 1, ny, nz)
    fe_x = np.stack([ey(fi, fj, fk), ez(fi, fj + 1, fk), ey(fi, fj, fk + 1), ez(fi, fj, fk)], 1)
    gi, gj, gk = grid(nx, ny + 1, nz)
    fe_y = np.stack([ez(gi, gj, gk), ex(gi, gj, gk + 1), ez(gi + 1, gj, gk), ex(gi, gj, gk)], 1)
    hi, hj, hk = grid(nx, ny, nz + 1)
    fe_z = np.stack([ex(hi, hj, hk), ey(hi + 1, hj, hk), ex(hi, hj + 1, hk), ey(hi, hj, hk)], 1)
    fe = np.concatenate([fe_x, fe_y, fe_z]).astype(np.int64)
    assert fe.shape[0] == nF
    B1 = (4 * np.arange(nF + 1, dtype=np.int64), fe.reshape(-1), (np.array([1.0, 1.0, -1.0, -1.0])[None, :] * o[:, None]).reshape(-1))
    B0 = (l2.ef_ptr, l2.ef_idx, l2.ef_sgn)
    ones = lambda k: np.arange(k + 1, dtype=np.int64)
    zeros = lambda k: np.zeros(k + 1, dtype=np.int64)
    nn = [nE, nF, nR, nV]
    idof = [[zeros(nE), zeros(nF), zeros(nR), ones(nV)], [zeros(nE), zeros(nF), ones(nR)], [zeros(nE), ones(nF)], [ones(nE)]]
    lv = SeqLevel(n=nn, B=[B0, B1, B2], facet_attr=l2.facet_attr, jstart=jstart, idof=idof, M=[None] * 4, D=[None] * 3, T=[None] * 4,
                  pv=[None] * 4, const3=l2.const3.copy(), meta=dict(S2=o, kind="hex", n=(nx, ny, nz), h=(hx, hy, hz)))
    lv.M[3] = [(l2.M3e_ptr, l2.M3e)]
    lv.M[2] = [(l2.M2e_ptr, l2.M2e), (l2.M2f_ptr, l2.M2f)]
    lv.D[2] = l2.D2
    lv.T[3], lv.pv[3], lv.T[2], lv.pv[2] = l2.T3, l2.pv3, l2.T2, l2.pv2
    m2 = np.array([[1.0 / 3, 1.0 / 6], [1.0 / 6, 1.0 / 3]])
    mm = np.kron(m2, m2)                                            # index (a, b) -> 2 a + b
    tb = _Tables(lv)
    if jstart <= 1:
        # cell mass of the edge elements: per direction the four parallel edges, (h_b h_c / h_a) m (x) m
        have = np.stack([ex(I, J, K), ex(I, J, K + 1), ex(I, J + 1, K), ex(I, J + 1, K + 1),          # (j', k') -> 2 j' + k'
                         ey(I, J, K), ey(I, J, K + 1), ey(I + 1, J, K), ey(I + 1, J, K + 1),          # (i', k')
                         ez(I, J, K), ez(I, J + 1, K), ez(I + 1, J, K), ez(I + 1, J + 1, K)], 1)      # (i', j')
        loc = np.zeros((12, 12))
        loc[0:4, 0:4] = mm * (hy * hz / hx); loc[4:8, 4:8] = mm * (hx * hz / hy); loc[8:12, 8:12] = mm * (hx * hy / hz)
        p02, i02 = tb.conn(0, 2)
        assert np.all(np.diff(p02) == 12)
        loc1 = _reorder_blocks(np.broadcast_to(loc[None], (nE, 12, 12)).copy(), have, i02.reshape(nE, 12))
        # face mass of the tangential trace, rows in the B1 row order (e0, e1, e2, e3): pairs (e0, e2) and (e1, e3) are parallel
        def face_block(ha, hb):                                     # e0/e2 along a (spaced by hb), e1/e3 along b (spaced by ha)
            blk = np.zeros((4, 4))
            blk[np.ix_([0, 2], [0, 2])] = m2 * (hb / ha)
            blk[np.ix_([3, 1], [3, 1])] = m2 * (ha / hb)
            return blk
        loc1f = np.concatenate([np.broadcast_to(face_block(hy, hz)[None], (fe_x.shape[0], 4, 4)),     # x-face: e0 = y-edge, e1 = z-edge
                                np.broadcast_to(face_block(hz, hx)[None], (fe_y.shape[0], 4, 4)),     # y-face: e0 = z-edge, e1 = x-edge
                                np.broadcast_to(face_block(hx, hy)[None], (fe_z.shape[0], 4, 4))])    # z-face: e0 = x-edge, e1 = y-edge
        lv.M[1] = [(144 * ones(nE), np.ascontiguousarray(loc1).reshape(-1)), (16 * ones(nF), np.ascontiguousarray(loc1f).reshape(-1)),
                   (ones(nR), 1.0 / elen)]
        lv.D[1] = sp.csr_matrix((B1[2], B1[1], B1[0]), shape=(nF, nR))
        T1 = np.zeros((nR, 3))
        T1[:nEx, 0] = hx; T1[nEx:nEx + nEy, 1] = hy; T1[nEx + nEy:, 2] = hz
        lv.T[1] = T1; lv.pv[1] = elen.copy()
    if jstart <= 0:
        have = np.stack([vid(I + a, J + b, K + c) for a in (0, 1) for b in (0, 1) for c in (0, 1)], 1)
        loc = np.kron(m2, np.kron(m2, m2)) * (hx * hy * hz)
        p03, i03 = tb.conn(0, 3)
        assert np.all(np.diff(p03) == 8)
        loc0 = _reorder_blocks(np.broadcast_to(loc[None], (nE, 8, 8)).copy(), have, i03.reshape(nE, 8))
        # face mass: vertices of the face in its two in-plane directions
        hv_x = np.stack([vid(fi, fj + b, fk + c) for b in (0, 1) for c in (0, 1)], 1)
        hv_y = np.stack([vid(gi + a, gj, gk + c) for a in (0, 1) for c in (0, 1)], 1)
        hv_z = np.stack([vid(hi + a, hj + b, hk) for a in (0, 1) for b in (0, 1)], 1)
        have_f = np.concatenate([hv_x, hv_y, hv_z])
        area = l2.pv2
        p13, i13 = tb.conn(1, 3)
        assert np.all(np.diff(p13) == 4)
        loc0f = _reorder_blocks(mm[None] * area[:, None, None], have_f, i13.reshape(nF, 4))
        loc0r = m2[None] * elen[:, None, None]
        lv.M[0] = [(64 * ones(nE), np.ascontiguousarray(loc0).reshape(-1)), (16 * ones(nF), np.ascontiguousarray(loc0f).reshape(-1)),
                   (4 * ones(nR), np.ascontiguousarray(loc0r).reshape(-1)), (ones(nV), np.ones(nV))]
        lv.D[0] = sp.csr_matrix((B2[2], B2[1], B2[0]), shape=(nR, nV))
        Vi, Vj, Vk = grid(nx + 1, ny + 1, nz + 1)
        lv.T[0] = np.stack([np.ones(nV), Vi * hx, Vj * hy, Vk * hz], 1); lv.pv[0] = np.ones(nV)
    return lv
