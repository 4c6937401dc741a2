"""Golden vectors of the AMGe coarsening (tests/golden/oracle_amge_hex.npz, written by tools/make_amge_golden.py from the
CPU oracle; nothing can be captured from the reference itself).  The CPU half pins the oracle against change, the GPU half
checks the device against the same stored numbers."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tools.make_amge_golden import canonical_signs, case, flatten

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_amge_hex.npz"))


def check_against_golden(coarsen, tol):
    """Every level is coarsened from the ORACLE's previous level, so that the inputs of level 1 (whose null-space dofs carry
    an arbitrary sign) are the ones the golden file was made from."""
    from oracle import amge as oa
    lv, parts, kinv = case()
    assert np.array_equal(kinv, G["kinv"])                                  # inputs are reproducible from the seed
    cur = lv
    for l, part in enumerate(parts):
        res = coarsen(cur, part)
        out = {}
        flatten(res, f"L{l}", out)
        for k, v in out.items():
            ref = G[k]
            if k.endswith(("_indptr", "_indices", "_n_null", "_ptr", "_idx", "_sgn")):
                assert np.array_equal(v, ref), k
            else:
                assert np.max(np.abs(v - ref)) <= tol * max(np.max(np.abs(ref)), 1e-300), k
        cur = oa.coarsen(cur, part)["coarse"]


def test_oracle_reproduces_its_golden_vectors():
    from oracle import amge as oa
    check_against_golden(oa.coarsen, 1e-12)


@pytest.mark.gpu
def test_device_reproduces_the_golden_vectors():
    from parelag_amd import amge

    def dev(lv, part):
        res = amge.coarsen(lv, part)
        res["Pi2"] = (res["hat1"] + res["hat0"] - res["hat0"] @ (res["P2"] @ res["hat1"])).tocsr()
        return res
    check_against_golden(dev, 1e-10)
