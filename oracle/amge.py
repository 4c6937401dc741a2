"""CPU restatement of ``DeRhamSequence::Coarsen`` for the (H(div), L2) pair -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``parelag_amd/amge.py`` + ``csrc/amge.hip``) never does.  The reference cannot be built here (MFEM, hypre, METIS,
SuiteSparse, LAPACK wrappers absent).  Pinned indirectly: the general oracle ``oracle/derham.py`` reproduces the reference's
printed upscaling errors (``tests/test_upscaling_golden.py``) and equals this module bit for bit on forms 2-3
(``test_general_oracle_equals_the_two_form_oracle``); directly by the reference's own invariants (``DeRhamSequence::CheckInvariants``, ``src/amge/DeRhamSequence.cpp:694-970``):
``D_f P_2 = P_3 D_c``, ``M_c = P^T M_f P``, ``Pi P = I``, targets reproduced -- see ``tests/test_amge_oracle.py``.

What is restated (reference file:line in every function): the path ``Coarsen()`` takes with ``jformStart_ = nDim-1 = 2``
(``examples/MultigridTestSPE10.cpp:441``, unstructured coarsening): ``CoarsenLocalPartitioning`` for codimensions 0/1,
``DofAgglomeration``, ``ComputeCoarseTracesWithTargets`` for forms 3 and 2, ``hFacetExtension(2)``,
``CochainProjector`` and the projection of targets.  Single rank (every agglomerated entity is owned; the shared-entity
branches ``AE_TrueAE.IsShared`` of ``DeRhamSequence.cpp:1820,2024`` do not occur).

Everything is written entity by entity with small dense numpy blocks, like the reference's loops over ``iAE``.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

from parelag_amd.amge_level import AmgeLevel, ptr_from_counts

EPS = np.finfo(np.float64).eps


# ---- topology ------------------------------------------------------------------------------------------------------------

def coarsen_topology(lv: AmgeLevel, part: np.ndarray) -> dict:
    """``AgglomeratedTopology::CoarsenLocalPartitioning`` (``src/topology/Topology.cpp:686-828``) for elements and facets with
    ``coarsefaces_algo = 0``; ``part`` must already have connected, non-empty parts (``connectedComponents``, ``:713-721``).

    Returns AE_elem (ptr, idx), AF_facet (ptr, idx, sgn), AE_AF (ptr, idx, sgn) and the coarse ``facet_attr``."""
    part = np.asarray(part)
    nAE = int(part.max()) + 1
    # AEntity_entity[0] = TransposeOrientation(partitioning): elements of an agglomerate in ascending order (:725)
    order = np.argsort(part, kind="stable")
    ae_ptr = ptr_from_counts(np.bincount(part, minlength=nAE))
    # AE_fc = MultOrientation(AEntity_entity[0], B_[0]) (:743): Gustavson product -> columns in first-touch order, entries
    # that cancel (facets interior to the agglomerate) dropped (TopologyTable.cpp:130-139)
    ae_fc = []
    for a in range(nAE):
        acc = {}
        for e in order[ae_ptr[a]:ae_ptr[a + 1]]:
            for q in range(lv.ef_ptr[e], lv.ef_ptr[e + 1]):
                f = int(lv.ef_idx[q])
                acc[f] = acc.get(f, 0.0) + lv.ef_sgn[q]
        ae_fc.append([(f, np.sign(v)) for f, v in acc.items() if abs(v) > 1e-10])
    # Z = AE_fc^T AE_fc (+ bdrAttr bdrAttr^T when the attribute table exists) (:757-766)
    rows = np.concatenate([np.full(len(r), a) for a, r in enumerate(ae_fc)])
    cols = np.array([f for r in ae_fc for f, _ in r], dtype=np.int64)
    vals = np.array([s for r in ae_fc for _, s in r])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(nAE, lv.nF))
    Z = (A.T @ A).tocsr()
    if np.any(lv.facet_attr > 0):
        bf = np.nonzero(lv.facet_attr > 0)[0]
        Battr = sp.csr_matrix((np.ones(bf.size), (bf, lv.facet_attr[bf] - 1)), shape=(lv.nF, int(lv.facet_attr.max())))
        Z = (Z + Battr @ Battr.T).tocsr()
    # findMinimalIntersectionSets(Z, .5) (src/structures/minimalIntersectionSet.cpp:43-132)
    tol = 1e-10
    dZ = Z.diagonal()
    mis = np.full(lv.nF, -1, dtype=np.int64)
    ori = np.zeros(lv.nF)
    in_mis = dZ - 0.5 > -tol
    cur = 0
    Zp, Zj, Zd = Z.indptr, Z.indices, Z.data
    for i in range(lv.nF):
        if in_mis[i] and mis[i] == -1:
            for q in range(Zp[i], Zp[i + 1]):
                j = Zj[q]
                if abs(dZ[j] - dZ[i]) < tol and (abs(Zd[q] - dZ[i]) < tol or abs(Zd[q] + dZ[i]) < tol):
                    mis[j] = cur
                    ori[j] = Zd[q] / dZ[i]
            cur += 1
    nAF = cur
    # AEntity_entity[1] = fc_AF^T (:775): facets of an AF ascending
    fidx = np.nonzero(mis >= 0)[0]
    o2 = np.argsort(mis[fidx], kind="stable")
    af_ptr = ptr_from_counts(np.bincount(mis[fidx], minlength=nAF))
    af_idx = fidx[o2]
    af_sgn = ori[af_idx]
    # CoarseTopology->B_[0] = MultOrientation(AE_fc, fc_AF) (:781)
    b_ptr, b_idx, b_sgn = [0], [], []
    for a in range(nAE):
        acc = {}
        for f, s in ae_fc[a]:
            acc[mis[f]] = acc.get(mis[f], 0.0) + s * ori[f]
        for F, v in acc.items():
            if abs(v) > 1e-10:
                b_idx.append(int(F)); b_sgn.append(np.sign(v))
        b_ptr.append(len(b_idx))
    # coarse facet_bdrAttribute = MultOrientation(AEntity_entity[1], facet_bdrAttribute) (:811-813)
    cattr = np.zeros(nAF, dtype=np.int32)
    for F in range(nAF):
        cattr[F] = lv.facet_attr[af_idx[af_ptr[F]]]
    return dict(nAE=nAE, nAF=nAF, ae_ptr=ae_ptr, ae_idx=order.astype(np.int64),
                af_ptr=af_ptr, af_idx=af_idx.astype(np.int64), af_sgn=af_sgn,
                B_ptr=np.array(b_ptr, dtype=np.int64), B_idx=np.array(b_idx, dtype=np.int64), B_sgn=np.array(b_sgn),
                facet_attr=cattr)


# ---- dof agglomeration --------------------------------------------------------------------------------------------------

def agglomerate_dofs(lv: AmgeLevel, topo: dict) -> dict:
    """``DofAgglomeration`` (``src/amge/DOFAgglomeration.cpp:33-314``) for forms 2 and 3.

    * form 3 (``maxCodimensionBaseForDof = 0``): row of AE = dofs of its elements, no reordering (``:122-127``);
    * form 2, facets (``= maxCodimensionBaseForDof``): row of AF = dofs of its facets (ascending facets), no reordering;
    * form 2, elements: ``AE_elem x elem_dof`` in first-touch order, then ``reorderRow`` (``:503-530``) sorts by
      (separator type, dof id): dofs interior to the agglomerate first, dofs on agglomerated facets after."""
    nAE, nAF = topo["nAE"], topo["nAF"]
    sep = np.zeros(lv.ndof2, dtype=np.int8)
    af_dofs = []
    for F in range(nAF):
        d = np.concatenate([lv.facet_dof2(f) for f in topo["af_idx"][topo["af_ptr"][F]:topo["af_ptr"][F + 1]]])
        af_dofs.append(d)
        sep[d] = 1
    ae_dofs3, ae_dofs2, ae_nint2 = [], [], []
    for a in range(nAE):
        els = topo["ae_idx"][topo["ae_ptr"][a]:topo["ae_ptr"][a + 1]]
        ae_dofs3.append(np.concatenate([lv.elem_dof3(e) for e in els]))
        seen = {}
        for e in els:
            for d in lv.elem_dof2(e):
                seen.setdefault(int(d), None)
        d2 = np.fromiter(seen.keys(), dtype=np.int64, count=len(seen))
        key = np.lexsort((d2, sep[d2]))
        d2 = d2[key]
        ae_dofs2.append(d2)
        ae_nint2.append(int(np.sum(sep[d2] == 0)))
    return dict(af_dofs=af_dofs, ae_dofs3=ae_dofs3, ae_dofs2=ae_dofs2, ae_nint2=ae_nint2)


def _assemble_ae_block(lv: AmgeLevel, els, dofs, which: str) -> np.ndarray:
    """Block of ``AssembleAgglomerateMatrix`` (``DOFAgglomeration.cpp:533-546``: ``ADof_rDof M_e ADof_rDof^T``) belonging to
    one agglomerate: element matrices summed in the agglomerate's own dof numbering (contributions of the agglomerate's
    elements only)."""
    loc = {int(d): i for i, d in enumerate(dofs)}
    M = np.zeros((len(dofs), len(dofs)))
    for e in els:
        ed = lv.elem_dof3(e) if which == "M3e" else lv.elem_dof2(e)
        li = np.array([loc[int(d)] for d in ed], dtype=np.int64)
        M[np.ix_(li, li)] += lv.block(which, e)
    return M


def _svd_on(W, A):
    """``SVD_Calculator::ComputeON`` (``src/linalg/dense/ParELAG_SVDCalculator.cpp:192-284``): left singular vectors of ``A``
    in the inner product ``W`` (``None`` = Euclidean, 1-D = diagonal weights, 2-D = dense SPD), singular values descending.
    Returns (U, s) with ``U`` overwriting the role of ``A``."""
    if A.shape[1] == 0:
        return A, np.zeros(0)
    if W is None:
        U, s, _ = np.linalg.svd(A, full_matrices=False)
        return U, s
    if W.ndim == 1:
        sw = np.sqrt(W)
        U, s, _ = np.linalg.svd(A * sw[:, None], full_matrices=False)
        return U / sw[:, None], s
    ev, V = np.linalg.eigh(W)
    X = (V * np.sqrt(ev)) @ V.T
    U, s, _ = np.linalg.svd(X @ A, full_matrices=False)
    return ((V / np.sqrt(ev)) @ V.T) @ U, s


def _dof_functional(P_loc, M_ii):
    """``CochainProjector::CreateDofFunctional`` (``src/amge/CochainProjector.cpp:53-137``): ``(P^T M P)^{-1} P^T M``."""
    nc = P_loc.shape[1]
    if nc == 0:
        return np.zeros((0, P_loc.shape[0]))
    MP = M_ii @ P_loc
    return np.linalg.solve(P_loc.T @ MP, MP.T)


# ---- coarse traces -------------------------------------------------------------------------------------------------------

def _coarse_traces(loc_pv, loc_T, Mloc, svd_tol):
    """Body of the first loop of ``ComputeCoarseTracesWithTargets`` (``DeRhamSequence.cpp:1818-1926``) for one agglomerated
    entity: deflate the PV vector out of the targets, orthonormalise them in the entity's mass inner product, keep the
    directions above the tolerance.  Returns (p_loc, cElemMass, dof functional)."""
    pv_dot_pv = float(loc_pv @ (Mloc @ loc_pv))
    T = loc_T - np.outer(loc_pv, (loc_pv @ (Mloc @ loc_T)) / pv_dot_pv)          # Deflate (InnerProduct.cpp:157-169)
    if T.shape[1] > 0:
        offdiag = Mloc - np.diag(np.diag(Mloc))
        U, s = _svd_on(np.diag(Mloc).copy() if not np.any(offdiag) else Mloc, T)
    else:
        U, s = T, np.zeros(0)
    s_max_tol = pv_dot_pv * svd_tol
    k = 0
    while k < s.size and not s[k] < s_max_tol:
        k += 1
    p_loc = np.concatenate([loc_pv[:, None], U[:, :k] * np.sqrt(pv_dot_pv)], axis=1)
    cmass = p_loc.T @ (Mloc @ p_loc)
    cmass = 0.5 * (cmass + cmass.T)
    return p_loc, cmass, _dof_functional(p_loc, Mloc)


# ---- Coarsen --------------------------------------------------------------------------------------------------------------

def coarsen(lv: AmgeLevel, part: np.ndarray, svd_tol: float = 1e-9, smallest_entry: float = EPS) -> dict:
    """``DeRhamSequence::Coarsen`` (``src/amge/DeRhamSequence.cpp:572-692``) with ``nForms_ = 4``, ``jformStart_ = 2``.

    Returns dict(P2, P3, Pi2, Pi3 (scipy CSR), coarse (AmgeLevel), topo)."""
    topo = coarsen_topology(lv, part)
    agg = agglomerate_dofs(lv, topo)
    nAE, nAF = topo["nAE"], topo["nAF"]

    # ===== codim 0, jform 3: ComputeCoarseTracesWithTargets(3) (:1723-2085); PV vector = constant 1 (computePVTraces) =====
    p3_blocks, m3c_blocks, fun3 = [], [], []
    for a in range(nAE):
        els = topo["ae_idx"][topo["ae_ptr"][a]:topo["ae_ptr"][a + 1]]
        dofs = agg["ae_dofs3"][a]
        Mloc = _assemble_ae_block(lv, els, dofs, "M3e")
        p_loc, cm, fn = _coarse_traces(lv.pv3[dofs].copy(), lv.T3[dofs].copy(), Mloc, svd_tol)
        p3_blocks.append(p_loc); m3c_blocks.append(cm); fun3.append(fn)
    n3c = np.array([b.shape[1] for b in p3_blocks], dtype=np.int64)
    c3_ptr = ptr_from_counts(n3c)                       # coarse form-3 dofs: [PV, null...] agglomerate by agglomerate (:1943-1951)
    ndof3c = int(c3_ptr[-1])
    P3 = _blocks_to_csr([(agg["ae_dofs3"][a], np.arange(c3_ptr[a], c3_ptr[a + 1]), p3_blocks[a]) for a in range(nAE)],
                        (lv.ndof3, ndof3c))
    Pi3 = _blocks_to_csr([(np.arange(c3_ptr[a], c3_ptr[a + 1]), agg["ae_dofs3"][a], fun3[a]) for a in range(nAE)],
                         (ndof3c, lv.ndof3))

    # ===== codim 1, jform 2: ComputeCoarseTracesWithTargets(2) on the agglomerated facets =====
    p2f_blocks, m2fc_blocks, fun2f = [], [], []
    for F in range(nAF):
        fs = topo["af_idx"][topo["af_ptr"][F]:topo["af_ptr"][F + 1]]
        sg = topo["af_sgn"][topo["af_ptr"][F]:topo["af_ptr"][F + 1]]
        dofs = agg["af_dofs"][F]
        nfd = np.array([lv.f_dof2_ptr[f + 1] - lv.f_dof2_ptr[f] for f in fs])
        Mloc = sla.block_diag(*[lv.block("M2f", f) for f in fs])
        # computePVTraces: orientation of the facet in its AF times the facet's PV dof values (:3235-3276, FE: DeRhamSequenceFE.cpp:811-860)
        loc_pv = lv.pv2[dofs] * np.repeat(sg, nfd)
        p_loc, cm, fn = _coarse_traces(loc_pv, lv.T2[dofs].copy(), Mloc, svd_tol)
        p2f_blocks.append(p_loc); m2fc_blocks.append(cm); fun2f.append(fn)
    n2f = np.array([b.shape[1] for b in p2f_blocks], dtype=np.int64)
    cf_ptr = ptr_from_counts(n2f)                       # coarse facet dofs [PV, null...] per AF, numbered first
    ntrace = int(cf_ptr[-1])

    # ===== hFacetExtension(2) (:2214-2581) =====
    D2csr = lv.D2.tocsr()
    T_p3 = P3.tocsr()
    ext_blocks, rt_blocks, null_blocks, dc_rows, m2ec_blocks, fun2e = [], [], [], [], [], []
    n_rt, n_null = np.zeros(nAE, np.int64), np.zeros(nAE, np.int64)
    # trace values R_d: fine dof -> (coarse trace dofs, values)
    Rd = _blocks_to_csr([(agg["af_dofs"][F], np.arange(cf_ptr[F], cf_ptr[F + 1]), p2f_blocks[F]) for F in range(nAF)],
                        (lv.ndof2, ntrace), keep_zeros=True).tocsr()
    for a in range(nAE):
        els = topo["ae_idx"][topo["ae_ptr"][a]:topo["ae_ptr"][a + 1]]
        ud = agg["ae_dofs2"][a]; ni = agg["ae_nint2"][a]
        pd = agg["ae_dofs3"][a]
        M_aa = _assemble_ae_block(lv, els, ud, "M2e")                 # M_d block
        W = _assemble_ae_block(lv, els, pd, "M3e")                    # W_d block
        D = D2csr[pd][:, ud].toarray()                                # D_d block (DistributeAgglomerateMatrix)
        B = W @ D                                                     # B_d = W_d D_d (:2258)
        npd = len(pd)
        # coarse dofs on the boundary of the agglomerate: hAE_AFCD = B_c x entity_dof(facets) (:2285-2290)
        AFs = topo["B_idx"][topo["B_ptr"][a]:topo["B_ptr"][a + 1]]
        cbd = np.concatenate([np.arange(cf_ptr[F], cf_ptr[F + 1]) for F in AFs]) if len(AFs) else np.zeros(0, np.int64)
        R_b = Rd[ud[ni:]][:, cbd].toarray()                            # (nb x nrhs_ext)
        # FacetSaddlePoint (src/linalg/solver_core/ParELAG_SaddlePointSolver.cpp:70-130)
        pv_p = T_p3[pd][:, [c3_ptr[a]]].toarray()[:, 0]                # P_[3](finePDof, PV dof of the agglomerate)
        tloc = W @ pv_p
        n = ni + npd + 1
        K = np.zeros((n, n))
        K[:ni, :ni] = M_aa[:ni, :ni]
        K[ni:ni + npd, :ni] = B[:, :ni]; K[:ni, ni:ni + npd] = B[:, :ni].T
        K[n - 1, ni:ni + npd] = tloc; K[ni:ni + npd, n - 1] = tloc
        lu = sla.lu_factor(K)
        # (3) extension of the traces (:2380-2409)
        rhs = np.zeros((n, len(cbd)))
        rhs[:ni] = -M_aa[:ni, ni:] @ R_b
        rhs[ni:ni + npd] = -B[:, ni:] @ R_b
        sol = sla.lu_solve(lu, rhs) if rhs.shape[1] else rhs
        ext = sol[:ni]
        l = sol[n - 1]
        dc_pv = np.where(np.abs(l) > smallest_entry, -l, 0.0)
        # (4) RangeT dofs: one bubble per non-PV coarse form-3 dof of the agglomerate (:2411-2446)
        nrt = int(c3_ptr[a + 1] - c3_ptr[a] - 1)
        rt = np.zeros((ni, nrt))
        if nrt:
            cP = T_p3[pd][:, np.arange(c3_ptr[a] + 1, c3_ptr[a + 1])].toarray()
            rhs = np.zeros((n, nrt))
            rhs[ni:ni + npd] = W @ cP
            rt = sla.lu_solve(lu, rhs)[:ni]
        # (5) null-space dofs from the targets (:2448-2528)
        nT = lv.T2.shape[1]
        nul = np.zeros((ni, 0))
        if ni > nrt and nT:
            t_i = lv.T2[ud[:ni]].copy(); t_b = lv.T2[ud[ni:]]
            rhs = np.zeros((n, nT))
            rhs[:ni] = -M_aa[:ni, ni:] @ t_b
            rhs[ni:ni + npd] = B[:, :ni] @ t_i
            t_i = t_i - sla.lu_solve(lu, rhs)[:ni]
            U, s = _svd_on(None, t_i)
            k = 0
            while k < s.size and not s[k] < svd_tol:
                k += 1
            nul = U[:, :k]
        n_rt[a], n_null[a] = nrt, nul.shape[1]
        ext_blocks.append((cbd, ext)); rt_blocks.append(rt); null_blocks.append(nul); dc_rows.append(dc_pv)
        # cochain projector functional (:2530-2538) and coarse element mass (:2540-2555)
        fun2e.append(_dof_functional(np.concatenate([rt, nul], axis=1), M_aa[:ni, :ni]))
        basis = np.zeros((len(ud), len(cbd) + nrt + nul.shape[1]))
        basis[:ni, :len(cbd)] = ext; basis[ni:, :len(cbd)] = R_b
        basis[:ni, len(cbd):len(cbd) + nrt] = rt
        basis[:ni, len(cbd) + nrt:] = nul
        cm = basis.T @ (M_aa @ basis)
        m2ec_blocks.append(0.5 * (cm + cm.T))
    nint_c = n_rt + n_null
    ci_ptr = ptr_from_counts(nint_c)                    # coarse element-interior dofs (RangeT then null), after the traces
    ndof2c = ntrace + int(ci_ptr[-1])

    # ---- assemble P_2, coarse D_2 ----
    blocks = [(agg["af_dofs"][F], np.arange(cf_ptr[F], cf_ptr[F + 1]), p2f_blocks[F]) for F in range(nAF)]
    drow, dcol, dval = [], [], []
    for a in range(nAE):
        ud = agg["ae_dofs2"][a]; ni = agg["ae_nint2"][a]
        cbd, ext = ext_blocks[a]
        ci = ntrace + np.arange(ci_ptr[a], ci_ptr[a + 1])
        blocks.append((ud[:ni], cbd, ext))
        blocks.append((ud[:ni], ci, np.concatenate([rt_blocks[a], null_blocks[a]], axis=1)))
        nz = dc_rows[a] != 0.0
        drow += [c3_ptr[a]] * int(nz.sum()); dcol += list(cbd[nz]); dval += list(dc_rows[a][nz])
        for i in range(n_rt[a]):
            drow.append(c3_ptr[a] + 1 + i); dcol.append(ci[i]); dval.append(1.0)
    P2 = _blocks_to_csr(blocks, (lv.ndof2, ndof2c))
    D2c = sp.csr_matrix((np.array(dval), (np.array(drow, dtype=np.int64), np.array(dcol, dtype=np.int64))), shape=(ndof3c, ndof2c))
    D2c.sort_indices()

    # ---- CochainProjector::ComputeProjector (CochainProjector.cpp:218-261): Pi_0 = Pi_1 + hatPi_0 (I - P Pi_1) ----
    hat1 = _blocks_to_csr([(np.arange(cf_ptr[F], cf_ptr[F + 1]), agg["af_dofs"][F], fun2f[F]) for F in range(nAF)],
                          (ndof2c, lv.ndof2))
    hat0 = _blocks_to_csr([(ntrace + np.arange(ci_ptr[a], ci_ptr[a + 1]), agg["ae_dofs2"][a][:agg["ae_nint2"][a]], fun2e[a])
                           for a in range(nAE)], (ndof2c, lv.ndof2))
    Pi2 = (hat1 + hat0 - hat0 @ (P2 @ hat1)).tocsr()
    Pi2.sort_indices()

    # ---- the coarser sequence ----
    n2e_c = np.array([m.shape[0] for m in m2ec_blocks], dtype=np.int64)
    coarse = AmgeLevel(
        nE=nAE, nF=nAF, ef_ptr=topo["B_ptr"], ef_idx=topo["B_idx"], ef_sgn=topo["B_sgn"], facet_attr=topo["facet_attr"],
        ndof3=ndof3c, e_dof3_ptr=c3_ptr, ndof2=ndof2c, f_dof2_ptr=cf_ptr, e_idof2_ptr=ci_ptr,
        M3e_ptr=ptr_from_counts(n3c * n3c), M3e=_pack(m3c_blocks),
        M2e_ptr=ptr_from_counts(n2e_c * n2e_c), M2e=_pack(m2ec_blocks),
        M2f_ptr=ptr_from_counts(n2f * n2f), M2f=_pack(m2fc_blocks),
        D2=D2c, T2=Pi2 @ lv.T2, T3=Pi3 @ lv.T3, const3=Pi3 @ lv.const3,      # :673-689
        pv2=_first_dof_indicator(cf_ptr, ndof2c), pv3=_first_dof_indicator(c3_ptr, ndof3c),   # DeRhamSequenceAlg::computePVTraces
        meta=dict(kind="alg"))
    return dict(P2=P2, P3=P3, Pi2=Pi2, Pi3=Pi3, coarse=coarse, topo=topo, agg=agg, n_rt=n_rt, n_null=n_null)


def _first_dof_indicator(ptr, n):
    v = np.zeros(n)
    v[ptr[:-1][np.diff(ptr) > 0]] = 1.0
    return v


def _pack(blocks):
    return np.concatenate([np.ascontiguousarray(b).reshape(-1) for b in blocks]) if blocks else np.zeros(0)


def _blocks_to_csr(blocks, shape, keep_zeros=False):
    """``SparseMatrix::SetSubMatrix`` of dense blocks; exact zeros are skipped like mfem's default ``skip_zeros``."""
    r, c, v = [], [], []
    for rows, cols, B in blocks:
        if B.size == 0:
            continue
        rr = np.repeat(np.asarray(rows, dtype=np.int64), len(cols))
        cc = np.tile(np.asarray(cols, dtype=np.int64), len(rows))
        vv = np.asarray(B).reshape(-1)
        if not keep_zeros:
            m = vv != 0.0
            rr, cc, vv = rr[m], cc[m], vv[m]
        r.append(rr); c.append(cc); v.append(vv)
    if not r:
        return sp.csr_matrix(shape)
    M = sp.csr_matrix((np.concatenate(v), (np.concatenate(r), np.concatenate(c))), shape=shape)
    M.sort_indices()
    return M


def coarsen_sequence(lv: AmgeLevel, parts: list, **kw) -> list:
    """Coarsen level after level (``examples/MultigridTestSPE10.cpp:460-466``); returns the list of ``coarsen`` results."""
    out = []
    for part in parts:
        res = coarsen(lv, part, **kw)
        out.append(res)
        lv = res["coarse"]
    return out
