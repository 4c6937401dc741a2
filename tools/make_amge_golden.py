#!/usr/bin/env python3
"""Golden vectors of the AMGe coarsening generated with the CPU oracle (nothing can be captured from the
reference: MFEM/LAPACK wrappers are absent).  Case: 5 x 4 x 3 hex grid, cells 20 x 10 x 2, log-normal coefficient (seed
below), boxes of 3 x 2 x 2 cells, then one more coarsening to 1 x 2 x 2 boxes.  Columns that come out of an SVD are stored
with the sign that makes their first significant entry positive.

    python tools/make_amge_golden.py            # writes tests/golden/oracle_amge_hex.npz
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from oracle import amge as oa
from parelag_amd.amge_level import box_partition, level_from_hex

SEED = 20240611


def canonical_signs(P):
    """+-1 per column: sign of the first entry (in row order) whose magnitude is at least half the column's largest -- robust
    against rounding when two entries have (nearly) equal magnitude."""
    P = sp.csc_matrix(P)
    s = np.ones(P.shape[1])
    for j in range(P.shape[1]):
        col = P.data[P.indptr[j]:P.indptr[j + 1]]
        if col.size:
            big = np.nonzero(np.abs(col) >= 0.5 * np.max(np.abs(col)))[0][0]
            s[j] = 1.0 if col[big] >= 0 else -1.0
    return s


def case():
    n = (5, 4, 3)
    kinv = np.exp(1.5 * np.random.default_rng(SEED).standard_normal(n[0] * n[1] * n[2]))
    lv = level_from_hex(n, h=(20.0, 10.0, 2.0), kinv=kinv)
    parts = [box_partition(n, (3, 2, 2)), box_partition((2, 2, 2), (1, 2, 2))]
    return lv, parts, kinv


def flatten(res, tag, out):
    c = res["coarse"]
    s2, s3 = canonical_signs(res["P2"]), canonical_signs(res["P3"])
    S2, S3 = sp.diags(s2), sp.diags(s3)
    for name, M in (("P2", res["P2"] @ S2), ("P3", res["P3"] @ S3), ("D2c", S3 @ c.D2 @ S2), ("M2c", S2 @ c.mass2() @ S2),
                    ("M3c", S3 @ c.mass3() @ S3), ("Pi2", S2 @ res["Pi2"])):
        M = sp.csr_matrix(M); M.sort_indices()
        out[f"{tag}_{name}_indptr"], out[f"{tag}_{name}_indices"], out[f"{tag}_{name}_data"] = M.indptr, M.indices, M.data
    out[f"{tag}_T2c"] = S2 @ c.T2
    out[f"{tag}_n_null"] = res["n_null"]
    for k in ("af_ptr", "af_idx", "af_sgn", "B_ptr", "B_idx", "B_sgn"):
        out[f"{tag}_{k}"] = res["topo"][k]


if __name__ == "__main__":
    lv, parts, kinv = case()
    out = dict(seed=SEED, kinv=kinv)
    cur = lv
    for l, part in enumerate(parts):
        res = oa.coarsen(cur, part)
        flatten(res, f"L{l}", out)
        cur = res["coarse"]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_amge_hex.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "bytes;", {k: np.asarray(v).shape for k, v in out.items() if k.endswith("n_null")})
