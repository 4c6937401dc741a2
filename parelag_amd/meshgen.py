"""Synthetic refined-cube456 de Rham operators (benchmark / test INPUT generator).

The reference obtains its operators from MFEM (``examples/MultigridTest0Form.cpp:124-212``
reads ``meshes/cube456.mesh`` and refines it uniformly; ``src/amge/DeRhamSequenceFE.cpp:633-684``
builds the lowest-order H1 / H(curl) / H(div) / L2 spaces, their mass matrices and the exterior
derivatives D0, D1, D2).  MFEM is not available, so this module restates the *lowest-order*
(Whitney) sequence on uniformly red-refined tetrahedra directly with numpy:

* entities are oriented by ascending global vertex id, so the derivative matrices are the
  simplicial boundary operators and ``D_{j+1} D_j = 0`` holds identically
  (invariant checked by ``DeRhamSequence::CheckInvariants``, ``src/amge/DeRhamSequence.cpp:707-751``);
* children of tet ``i`` are numbered ``8 i .. 8 i + 7`` so that the derefinement partitioner
  (``src/partitioning/MFEMRefinedMeshPartitioner.cpp:64-70``, ``p[j] = j / 8``) recovers the
  parent -- every AMGe level therefore *is* the mesh with one refinement less;
* the interpolators ``P_j`` are the canonical nested Whitney interpolations, which satisfy
  ``D_f P_j = P_{j+1} D_c`` (``DeRhamSequence.cpp:753-800``).  ParElag's own AMGe ``P`` comes from
  per-agglomerate saddle-point solves (``DeRhamSequence::Coarsen``) and is out of scope.

Nothing in here is on the product path: it only manufactures inputs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_FIXTURE = os.path.join(_HERE, "data", "cube456.npz")     # the reference's benchmark mesh as arrays (an input of the workload generator, not a test fixture)

# local edges / faces of a tet in terms of local vertex numbers
_LE = np.array([[0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3]])
_LF = np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]])  # face k is opposite vertex k


def load_cube456(path: str = _FIXTURE):
    z = np.load(path)
    return z["vertices"].copy(), z["tets"].astype(np.int64)


def _unique_rows(a: np.ndarray, nv: int):
    """Unique rows of a small-int matrix via a scalar key; returns (unique_rows, inverse)."""
    key = a[:, 0].astype(np.int64)
    for c in range(1, a.shape[1]):
        key = key * nv + a[:, c]
    uk, inv = np.unique(key, return_inverse=True)
    cols = []
    k = uk.copy()
    for c in range(a.shape[1] - 1, -1, -1):
        cols.append(k % nv)
        k //= nv
    return np.stack(cols[::-1], axis=1), inv


@dataclass
class TetMesh:
    """A conforming tet mesh with globally numbered, vertex-sorted edges and faces."""
    vertices: np.ndarray          # (V,3) float64
    tets: np.ndarray              # (T,4) int64, any vertex order
    edges: np.ndarray = field(default=None)      # (E,2) sorted
    tet_edges: np.ndarray = field(default=None)  # (T,6) edge ids of _LE on *sorted* tet vertices
    faces: np.ndarray = field(default=None)      # (F,3) sorted
    tet_faces: np.ndarray = field(default=None)  # (T,4) face ids (face k opposite sorted vertex k)
    stets: np.ndarray = field(default=None)      # (T,4) tets with vertices sorted ascending

    @property
    def nv(self): return self.vertices.shape[0]
    @property
    def nt(self): return self.tets.shape[0]
    @property
    def ne(self): return self.edges.shape[0]
    @property
    def nf(self): return self.faces.shape[0]

    def build_edges(self):
        if self.edges is not None:
            return
        self.stets = np.sort(self.tets, axis=1)
        e = self.stets[:, _LE].reshape(-1, 2)
        self.edges, inv = _unique_rows(e, self.nv)
        self.tet_edges = inv.reshape(-1, 6)

    def build_faces(self):
        if self.faces is not None:
            return
        self.build_edges()
        f = self.stets[:, _LF].reshape(-1, 3)
        self.faces, inv = _unique_rows(f, self.nv)
        self.tet_faces = inv.reshape(-1, 4)

    # ---- uniform (red) refinement; children of tet i are 8i..8i+7 -----------------------
    def refine(self) -> "TetMesh":
        self.build_edges()
        V = self.nv
        mid = 0.5 * (self.vertices[self.edges[:, 0]] + self.vertices[self.edges[:, 1]])
        verts = np.vstack([self.vertices, mid])
        t = self.stets
        m = V + self.tet_edges  # midpoints m01 m02 m03 m12 m13 m23
        v0, v1, v2, v3 = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        m01, m02, m03, m12, m13, m23 = (m[:, i] for i in range(6))
        ch = np.stack([
            np.stack([v0, m01, m02, m03], 1),
            np.stack([m01, v1, m12, m13], 1),
            np.stack([m02, m12, v2, m23], 1),
            np.stack([m03, m13, m23, v3], 1),
            # octahedron split along the m02-m13 diagonal (Bey)
            np.stack([m01, m02, m03, m13], 1),
            np.stack([m01, m02, m12, m13], 1),
            np.stack([m02, m03, m13, m23], 1),
            np.stack([m02, m12, m13, m23], 1),
        ], axis=1)  # (T,8,4)
        return TetMesh(verts, ch.reshape(-1, 4))

    # ---- geometry ----------------------------------------------------------------------
    def _sorted_tets(self):
        if self.stets is None:
            self.stets = np.sort(self.tets, axis=1)
        return self.stets

    def _grads(self, sl=slice(None)):
        """Gradients of the 4 barycentric functions on sorted tets, |T| and orientation sign (chunk ``sl``)."""
        x = self.vertices[self._sorted_tets()[sl]]              # (n,4,3)
        e1, e2, e3 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0], x[:, 3] - x[:, 0]
        c23, c31, c12 = np.cross(e2, e3), np.cross(e3, e1), np.cross(e1, e2)
        det = np.einsum("td,td->t", e1, c23)
        g = np.empty((x.shape[0], 4, 3))
        g[:, 1, :] = c23 / det[:, None]
        g[:, 2, :] = c31 / det[:, None]
        g[:, 3, :] = c12 / det[:, None]
        g[:, 0, :] = -(g[:, 1, :] + g[:, 2, :] + g[:, 3, :])
        return g, np.abs(det) / 6.0, np.sign(det)

    # ---- exterior derivatives (simplicial coboundary, entity x sub-entity) ----------------
    def D0(self) -> sp.csr_matrix:
        self.build_edges()
        E = self.ne
        rows = np.repeat(np.arange(E), 2)
        cols = self.edges.reshape(-1)
        vals = np.tile(np.array([-1.0, 1.0]), E)
        return sp.csr_matrix((vals, (rows, cols)), shape=(E, self.nv))

    def _edge_ids(self, pairs):
        key = pairs[:, 0] * self.nv + pairs[:, 1]
        ekey = self.edges[:, 0] * self.nv + self.edges[:, 1]
        idx = np.searchsorted(ekey, key)
        assert np.array_equal(ekey[idx], key)
        return idx

    def D1(self) -> sp.csr_matrix:
        self.build_faces()
        F = self.nf
        f = self.faces
        e_bc = self._edge_ids(f[:, [1, 2]])
        e_ac = self._edge_ids(f[:, [0, 2]])
        e_ab = self._edge_ids(f[:, [0, 1]])
        rows = np.repeat(np.arange(F), 3)
        cols = np.stack([e_bc, e_ac, e_ab], 1).reshape(-1)
        vals = np.tile(np.array([1.0, -1.0, 1.0]), F)
        return sp.csr_matrix((vals, (rows, cols)), shape=(F, self.ne))

    def D2(self) -> sp.csr_matrix:
        self.build_faces()
        T = self.nt
        rows = np.repeat(np.arange(T), 4)
        cols = self.tet_faces.reshape(-1)
        vals = np.tile(np.array([1.0, -1.0, 1.0, -1.0]), T)
        return sp.csr_matrix((vals, (rows, cols)), shape=(T, self.nf))

    # ---- Whitney mass matrices -----------------------------------------------------------
    def _assemble(self, dofs, loc, n, chunk):
        """Sum local matrices ``loc`` (T,k,k) into an (n,n) CSR on dof table ``dofs`` (T,k)."""
        k = dofs.shape[1]
        r = np.repeat(dofs, k, axis=1).reshape(-1)
        c = np.tile(dofs, (1, k)).reshape(-1)
        return sp.csr_matrix((loc.reshape(-1), (r, c)), shape=(n, n))

    def M0_P1_stiffness(self, chunk=1 << 20) -> sp.csr_matrix:
        """P1 stiffness (grad u, grad v); equals D0^T M1 D0 (checked in tests).  Assembled from the six
        off-diagonal entries per tet through a COO -> CSR conversion (linear time, no edge table needed), so the
        x5 / x6 operators of the roofline measurement are generated in seconds."""
        V = self.nv
        st = self._sorted_tets()
        nt = self.nt
        rows = np.empty(6 * nt, np.int32); cols = np.empty(6 * nt, np.int32); vals = np.empty(6 * nt)
        vx, vy, vz = (np.ascontiguousarray(self.vertices[:, d]) for d in range(3))
        for s in range(0, nt, chunk):
            e = min(nt, s + chunk)
            t = st[s:e]
            # edge vectors from vertex 0, component-wise (1-D arrays keep numpy in its fast path)
            X = [[c[t[:, q]] for c in (vx, vy, vz)] for q in range(4)]
            E = [[X[q][d] - X[0][d] for d in range(3)] for q in (1, 2, 3)]

            def cross(a, b):
                return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
            c23, c31, c12 = cross(E[1], E[2]), cross(E[2], E[0]), cross(E[0], E[1])
            det = E[0][0] * c23[0] + E[0][1] * c23[1] + E[0][2] * c23[2]
            g = [None, c23, c31, c12]                       # det * grad(lambda_q), q = 1..3
            g[0] = [-(c23[d] + c31[d] + c12[d]) for d in range(3)]
            scale = 1.0 / (6.0 * np.abs(det))               # |T| / det^2
            o = slice(6 * s, 6 * e)
            r_, c_, v_ = rows[o].reshape(-1, 6), cols[o].reshape(-1, 6), vals[o].reshape(-1, 6)
            for k, (i, j) in enumerate(_LE):
                v_[:, k] = (g[i][0] * g[j][0] + g[i][1] * g[j][1] + g[i][2] * g[j][2]) * scale
                r_[:, k] = t[:, i]
                c_[:, k] = t[:, j]
        U = sp.coo_matrix((vals, (rows, cols)), shape=(V, V)).tocsr()
        U.sum_duplicates()
        del rows, cols, vals
        diag = -(np.asarray(U.sum(axis=1)).ravel() + np.asarray(U.sum(axis=0)).ravel())
        # assemble by index lists (scipy's sparse "+" would drop the structurally present exact zeros)
        ur = np.repeat(np.arange(V, dtype=np.int32), np.diff(U.indptr))
        uc, uv = U.indices, U.data
        dv = np.arange(V, dtype=np.int32)
        A = sp.coo_matrix((np.concatenate([diag, uv, uv]), (np.concatenate([dv, ur, uc]), np.concatenate([dv, uc, ur]))),
                          shape=(V, V)).tocsr()
        A.sort_indices()
        return A

    def M1(self, chunk=1 << 18) -> sp.csr_matrix:
        """Lowest-order Nedelec (Whitney 1-form) mass matrix, unit coefficient."""
        self.build_edges()
        acc = None
        I = (np.ones((4, 4)) + np.eye(4)) / 20.0
        i, j = _LE[:, 0], _LE[:, 1]
        for s in range(0, self.nt, chunk):
            sl = slice(s, min(self.nt, s + chunk))
            g, vol, _ = self._grads(sl)
            G = np.einsum("tid,tjd->tij", g, g)
            # M[e,f] = I_ik G_jl - I_il G_jk - I_jk G_il + I_jl G_ik   (e=(i,j), f=(k,l))
            ik = I[np.ix_(i, i)]; il = I[np.ix_(i, j)]; jk = I[np.ix_(j, i)]; jl = I[np.ix_(j, j)]
            loc = (ik[None] * G[:, j][:, :, j] - il[None] * G[:, j][:, :, i]
                   - jk[None] * G[:, i][:, :, j] + jl[None] * G[:, i][:, :, i]) * vol[:, None, None]
            m = self._assemble(self.tet_edges[sl], loc, self.ne, chunk)
            acc = m if acc is None else acc + m
        acc.sum_duplicates(); acc.sort_indices()
        return acc

    def _face_sign(self, sl=slice(None)):
        """sigma[t,k] = +1 if the sorted orientation of face k (opposite sorted vertex k) has its
        normal pointing out of tet t."""
        x = self.vertices[self.stets[sl]]
        sig = np.empty((x.shape[0], 4))
        for k in range(4):
            a, b, c = (x[:, _LF[k, q]] for q in range(3))
            nrm = np.cross(b - a, c - a)
            sig[:, k] = np.sign(np.einsum("td,td->t", nrm, a - x[:, k]))
        return sig

    def M2(self, chunk=1 << 18, weight: np.ndarray | None = None) -> sp.csr_matrix:
        """Lowest-order Raviart-Thomas (Whitney 2-form) mass matrix; ``weight`` is an optional per-element scalar
        coefficient (e.g. the inverse permeability of ``VectorFEMassIntegrator(kinv)``, MultigridTestSPE10.cpp:377-395)."""
        self.build_faces()
        acc = None
        I = (np.ones((4, 4)) + np.eye(4)) / 20.0
        for s in range(0, self.nt, chunk):
            sl = slice(s, min(self.nt, s + chunk))
            x = self.vertices[self.stets[sl]]
            _, vol, _ = self._grads(sl)
            sig = self._face_sign(sl)
            # d[t,k,i,:] = x_i - x_k
            d = x[:, None, :, :] - x[:, :, None, :]
            # loc[k,l] = sum_ij I_ij (x_i-x_k).(x_j-x_l) * vol / (9 vol^2)
            loc = np.einsum("ij,tkid,tljd->tkl", I, d, d) / (9.0 * vol)[:, None, None]
            loc *= sig[:, :, None] * sig[:, None, :]
            if weight is not None:
                loc *= np.asarray(weight, dtype=np.float64)[sl][:, None, None]
            m = self._assemble(self.tet_faces[sl], loc, self.nf, chunk)
            acc = m if acc is None else acc + m
        acc.sum_duplicates(); acc.sort_indices()
        return acc

    def M3(self) -> sp.csr_matrix:
        """L2 (piecewise-constant density 1/|T|) mass matrix."""
        self.build_edges()
        vol = np.concatenate([self._grads(slice(s, s + (1 << 20)))[1] for s in range(0, self.nt, 1 << 20)])
        return sp.diags(1.0 / vol).tocsr()

    # ---- boundary markers (all 6 boundary attributes essential: MultigridTest0Form.cpp:194-196)
    def boundary_faces(self):
        self.build_faces()
        cnt = np.bincount(self.tet_faces.reshape(-1), minlength=self.nf)
        return np.nonzero(cnt == 1)[0]

    def boundary_marker_unit_cube(self) -> np.ndarray:
        """Vertex marker from coordinates: cube456 meshes [0,1]^3 and every boundary vertex has a coordinate that is
        exactly 0.0 or 1.0 (midpoints of such points on one face stay exact).  Equal to boundary_marker(0) (tested);
        avoids building the face table of the x5 mesh."""
        v = self.vertices
        return np.any((v == 0.0) | (v == 1.0), axis=1)

    def boundary_marker(self, form: int) -> np.ndarray:
        self.build_faces()
        bf = self.faces[self.boundary_faces()]
        if form == 0:
            mk = np.zeros(self.nv, dtype=bool); mk[bf.reshape(-1)] = True
        elif form == 1:
            mk = np.zeros(self.ne, dtype=bool)
            for p in ([1, 2], [0, 2], [0, 1]):
                mk[self._edge_ids(bf[:, p])] = True
        elif form == 2:
            mk = np.zeros(self.nf, dtype=bool); mk[self.boundary_faces()] = True
        else:
            mk = np.zeros(self.nt, dtype=bool)
        return mk


# ---- nested Whitney interpolators  coarse mesh -> its refinement ---------------------------

def interp_P0(coarse: TetMesh, fine: TetMesh) -> sp.csr_matrix:
    coarse.build_edges()
    Vc, Ec = coarse.nv, coarse.ne
    assert fine.nv == Vc + Ec
    rows = np.concatenate([np.arange(Vc), np.repeat(Vc + np.arange(Ec), 2)])
    cols = np.concatenate([np.arange(Vc), coarse.edges.reshape(-1)])
    vals = np.concatenate([np.ones(Vc), np.full(2 * Ec, 0.5)])
    P = sp.csr_matrix((vals, (rows, cols)), shape=(fine.nv, Vc))
    P.sort_indices()
    return P


def _bary_of_fine_vertices(coarse: TetMesh, fine: TetMesh):
    """lam[t, c, q, i] = barycentric coordinate (w.r.t. sorted coarse tet t, vertex i) of vertex q of
    child c's *sorted* vertex list."""
    ct = coarse.stets                                        # (T,4)
    fs = fine.stets.reshape(coarse.nt, 8, 4)                 # children 8t..8t+7
    Vc = coarse.nv
    lam = np.zeros((coarse.nt, 8, 4, 4))
    is_old = fs < Vc
    for i in range(4):
        lam[..., i] += (is_old & (fs == ct[:, None, None, i]))
    eid = np.where(is_old, 0, fs - Vc)
    ends = coarse.edges[eid]                                 # (T,8,4,2)
    for i in range(4):
        hit = (~is_old)[..., None] & (ends == ct[:, None, None, None, i])
        lam[..., i] += 0.5 * hit.sum(axis=-1)
    return lam


def _interp_generic(rows_f, cols_c, vals, shape):
    nz = vals != 0.0
    rows_f, cols_c, vals = rows_f[nz], cols_c[nz], vals[nz]
    key = rows_f.astype(np.int64) * shape[1] + cols_c
    uk, first = np.unique(key, return_index=True)
    P = sp.csr_matrix((vals[first], (uk // shape[1], uk % shape[1])), shape=shape)
    P.sort_indices()
    return P


def interp_P1(coarse: TetMesh, fine: TetMesh) -> sp.csr_matrix:
    """Whitney 1-form interpolation: P[e_f, E] = int_{e_f} w_E = det [lam_i(a) lam_j(a); lam_i(b) lam_j(b)]."""
    coarse.build_edges(); fine.build_edges()
    lam = _bary_of_fine_vertices(coarse, fine)               # (T,8,4,4)
    fe = fine.tet_edges.reshape(coarse.nt, 8, 6)
    ce = coarse.tet_edges                                    # (T,6)
    a, b = _LE[:, 0], _LE[:, 1]                              # fine local edge endpoints
    i, j = _LE[:, 0], _LE[:, 1]                              # coarse local edge endpoints
    la = lam[:, :, a, :]; lb = lam[:, :, b, :]               # (T,8,6,4)
    # val[t,c,p,q] = lam_i(a_p) lam_j(b_p) - lam_j(a_p) lam_i(b_p)   (i,j)=coarse edge q
    val = la[..., i] * lb[..., j] - la[..., j] * lb[..., i]  # (T,8,6,6)
    rows = np.broadcast_to(fe[..., None], val.shape).reshape(-1)
    cols = np.broadcast_to(ce[:, None, None, :], val.shape).reshape(-1)
    return _interp_generic(rows, cols, val.reshape(-1), (fine.ne, coarse.ne))


def interp_P2(coarse: TetMesh, fine: TetMesh) -> sp.csr_matrix:
    """Whitney 2-form interpolation: P[f, F] = int_f w_F = det[lam_{ijk}(a,b,c)]."""
    coarse.build_faces(); fine.build_faces()
    lam = _bary_of_fine_vertices(coarse, fine)
    ff = fine.tet_faces.reshape(coarse.nt, 8, 4)
    cf = coarse.tet_faces
    val = np.empty((coarse.nt, 8, 4, 4))
    for p in range(4):                                       # fine local face
        L = lam[:, :, _LF[p], :]                             # (T,8,3,4) rows = a,b,c
        for q in range(4):                                   # coarse local face
            val[:, :, p, q] = np.linalg.det(L[..., _LF[q]])
    val[np.abs(val) < 1e-14] = 0.0
    rows = np.broadcast_to(ff[..., None], val.shape).reshape(-1)
    cols = np.broadcast_to(cf[:, None, None, :], val.shape).reshape(-1)
    return _interp_generic(rows, cols, val.reshape(-1), (fine.nf, coarse.nf))


def interp_P3(coarse: TetMesh, fine: TetMesh) -> sp.csr_matrix:
    """Whitney 3-form (densities): signed volume ratio of child to parent."""
    _, volc, sc = coarse._grads()
    _, volf, sf = fine._grads()
    val = (volf * sf).reshape(coarse.nt, 8) / (volc * sc)[:, None]
    rows = np.arange(fine.nt)
    cols = np.repeat(np.arange(coarse.nt), 8)
    return sp.csr_matrix((val.reshape(-1), (rows, cols)), shape=(fine.nt, coarse.nt))


# ---- boundary-condition helpers mirroring MFEM semantics -----------------------------------

def eliminate_rowcol(A: sp.csr_matrix, marker: np.ndarray) -> sp.csr_matrix:
    """``SparseMatrix::EliminateRowCol(rc, 0.0, rhs)`` (DIAG_ONE) for every marked dof: stored entries
    in marked rows/columns become 0 (pattern kept), diagonal becomes 1
    (call site ``examples/MultigridTest0Form.cpp:462-464``)."""
    A = A.tocsr(copy=True)
    A.sort_indices()
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    kill = marker[rows] | marker[A.indices]
    A.data[kill] = 0.0
    A.data[kill & (rows == A.indices)] = 1.0
    return A


def eliminate_cols(P: sp.csr_matrix, marker: np.ndarray) -> sp.csr_matrix:
    """``SparseMatrix::EliminateCols(marker)``: zero stored values in marked columns, keep the pattern
    (``src/amge/DeRhamSequence.cpp:1230-1253``)."""
    P = P.tocsr(copy=True)
    P.data[marker[P.indices]] = 0.0
    return P


@dataclass
class Sequence:
    """Meshes of a ``nref``-times refined cube456, finest first (level 0 = finest, as in ParElag)."""
    meshes: list

    @classmethod
    def cube456(cls, nref: int, nlevels: int | None = None) -> "Sequence":
        v, t = load_cube456()
        ms = [TetMesh(v, t)]
        for _ in range(nref):
            ms.append(ms[-1].refine())
        ms = ms[::-1]
        if nlevels is not None:
            ms = ms[:nlevels]
        return cls(ms)

    @classmethod
    def kuhn_cube(cls, n: int, nref: int, nlevels: int | None = None) -> "Sequence":
        """A small synthetic base mesh for unit tests: the unit cube cut into n^3 cells of 6 Kuhn tets each (conforming),
        refined ``nref`` times like cube456."""
        g = np.arange(n + 1)
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float64) / n
        vid = lambda i, j, k: (i * (n + 1) + j) * (n + 1) + k
        perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
        tets = []
        for i in range(n):
            for j in range(n):
                for k in range(n):
                    for pm in perms:
                        c = [i, j, k]
                        path = [vid(*c)]
                        for ax in pm:
                            c[ax] += 1
                            path.append(vid(*c))
                        tets.append(path)
        ms = [TetMesh(verts, np.asarray(tets, dtype=np.int64))]
        for _ in range(nref):
            ms.append(ms[-1].refine())
        ms = ms[::-1]
        if nlevels is not None:
            ms = ms[:nlevels]
        return cls(ms)

    def partition(self, level: int) -> np.ndarray:
        """Element -> agglomerate map of level ``level`` (``MFEMRefinedMeshPartitioner::Partition``)."""
        return np.arange(self.meshes[level].nt, dtype=np.int32) // 8


def load_vector_P1(mesh: TetMesh, marker: np.ndarray | None = None) -> np.ndarray:
    """(f, phi_i) for f = 1 and P1 basis functions: |T|/4 per vertex of every tet (the DomainLFIntegrator of
    ``examples/MultigridTest0Form.cpp:455-466``), essential entries zeroed (homogeneous Dirichlet data)."""
    _g, vol, _s = mesh._grads()
    b = np.zeros(mesh.nv)
    np.add.at(b, mesh._sorted_tets().reshape(-1), np.repeat(vol / 4.0, 4))
    if marker is not None:
        b[marker] = 0.0
    return b


def h1_problem(nref: int, nlevels: int | None = None):
    """0-form problem of MultigridTest0Form: A = D0^T M1 D0 with all boundary dofs essential, and the
    true interpolators with essential columns zeroed (ComputeTrueP(jform, ess)).

    Returns (A_fine, [P_0, P_1, ...]) as scipy CSR; P_l maps level l+1 -> level l."""
    seq = Sequence.cube456(nref, nlevels)
    fine = seq.meshes[0]
    A = eliminate_rowcol(fine.M0_P1_stiffness(), fine.boundary_marker_unit_cube())
    Ps = []
    for l in range(len(seq.meshes) - 1):
        f, c = seq.meshes[l], seq.meshes[l + 1]
        Ps.append(eliminate_cols(interp_P0(c, f), c.boundary_marker_unit_cube()))
    return A, Ps, seq


def hdiv_problem(nref: int, nlevels: int | None = None, alpha: float = 1.0, beta: float = 1.0):
    """2-form problem of MultigridTest2Form: A = alpha M2 + beta D2^T M3 D2 (``examples/MultigridTest2Form.cpp:541-559``)
    with every boundary face essential, the true interpolators P2 with essential columns zeroed and, for the Hiptmair
    smoother, the true curl D1 of every level with essential (boundary-edge) columns zeroed
    (``HiptmairSmootherFactory.cpp:84-103`` -> ``DeRhamSequence::ComputeTrueD(form-1, ess)``).

    Returns (A_fine, [P_l], [D_l], seq); P_l maps level l+1 -> l, D_l is faces x edges on level l."""
    seq = Sequence.cube456(nref, nlevels)
    fine = seq.meshes[0]
    D2 = fine.D2()
    A = (alpha * fine.M2() + beta * (D2.T @ fine.M3() @ D2)).tocsr()
    A.sort_indices()
    A = eliminate_rowcol(A, fine.boundary_marker(2))
    Ps, Ds = [], []
    for l, m in enumerate(seq.meshes):
        Ds.append(eliminate_cols(m.D1(), m.boundary_marker(1)))
        if l < len(seq.meshes) - 1:
            c = seq.meshes[l + 1]
            Ps.append(eliminate_cols(interp_P2(c, m), c.boundary_marker(2)))
    return A, Ps, Ds, seq


def hcurl_problem(nref: int, nlevels: int | None = None, alpha: float = 1.0, beta: float = 1.0):
    """1-form problem of MultigridTest1Form: A = beta M1 + alpha D1^T M2 D1 (``examples/MultigridTest1Form.cpp:262-263,487-492``)
    with every boundary edge essential, the true interpolators P1 with essential columns zeroed and, for the Hiptmair
    smoother, the true gradient D0 of every level with essential (boundary-vertex) columns zeroed
    (``HiptmairSmootherFactory.cpp:84-103``).  Returns (A_fine, [P_l], [D_l], seq)."""
    seq = Sequence.cube456(nref, nlevels)
    fine = seq.meshes[0]
    D1 = fine.D1()
    A = (beta * fine.M1() + alpha * (D1.T @ fine.M2() @ D1)).tocsr()
    A.sort_indices()
    A = eliminate_rowcol(A, fine.boundary_marker(1))
    Ps, Ds = [], []
    for l, m in enumerate(seq.meshes):
        Ds.append(eliminate_cols(m.D0(), m.boundary_marker(0)))
        if l < len(seq.meshes) - 1:
            c = seq.meshes[l + 1]
            Ps.append(eliminate_cols(interp_P1(c, m), c.boundary_marker(1)))
    return A, Ps, Ds, seq


def lognormal_inverse_permeability(seq: "Sequence", sigma: float = 2.0, seed: int = 2025) -> np.ndarray:
    """Synthetic stand-in for the SPE10 data set (``data/spe_perm.dat`` is not shipped with the reference,
    ``examples/MultigridTestSPE10.cpp:85``): one log-normal value ``exp(sigma * N(0,1))`` per element of the COARSEST mesh
    (piecewise constant on the agglomerates, so contrast survives every level), inherited by the children.
    Returns the per-element inverse permeability on the finest mesh."""
    rng = np.random.default_rng(seed)
    k = np.exp(sigma * rng.standard_normal(seq.meshes[-1].nt))
    for _ in range(len(seq.meshes) - 1):
        k = np.repeat(k, 8)
    return 1.0 / k


def darcy_problem(nref: int, nlevels: int | None = None, seq: "Sequence | None" = None, kinv: np.ndarray | None = None):
    """Mixed Darcy system of MultigridTestDarcy (``examples/MultigridTestDarcy.cpp:417-441``): blocks ``M`` (2-form mass),
    ``B = W D`` (``W`` the 3-form mass, ``D`` the divergence) and ``Bt = B^T`` of ``[[M, Bt], [B, 0]]``; no essential
    boundary labels on either form (``:479-480`` passes two empty label lists), so the true interpolators are the plain
    ``P2`` / ``P3`` of every level.

    ``kinv``: optional per-element inverse permeability (heterogeneous-coefficient Darcy, BASELINE configs[4]).

    Returns dict(M, B, Bt, Ps=[(P2_l, P3_l)], seq); P*_l maps level l+1 -> l."""
    if seq is None:
        seq = Sequence.cube456(nref, nlevels)
    fine = seq.meshes[0]
    M = fine.M2(weight=kinv).tocsr()
    B = (fine.M3() @ fine.D2()).tocsr()
    for X in (M, B):
        X.sort_indices()
    Bt = B.T.tocsr()
    Bt.sort_indices()
    Ps = []
    for l in range(len(seq.meshes) - 1):
        f, c = seq.meshes[l], seq.meshes[l + 1]
        Ps.append((interp_P2(c, f), interp_P3(c, f)))
    return dict(M=M, B=B, Bt=Bt, Ps=Ps, seq=seq)
