#!/usr/bin/env python3
"""Convert the reference's benchmark mesh (meshes/cube456.mesh, NETGEN neutral
format, read by examples/MultigridTest0Form.cpp:147) into a compact numpy
fixture so the GPU box (which has no /root/reference) can build the synthetic
refined-cube456 operators.  Data only: 141 vertices, 456 tets, 206 boundary
triangles with attributes 1..6.

    python tools/make_cube456_fixture.py [/root/reference/meshes/cube456.mesh]
"""
import sys
import numpy as np

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/meshes/cube456.mesh"
tok = open(src).read().split()
assert tok[0] == "NETGEN_Neutral_Format"
p = 1
nv = int(tok[p]); p += 1
verts = np.array(tok[p:p + 3 * nv], dtype=np.float64).reshape(nv, 3); p += 3 * nv
nt = int(tok[p]); p += 1
t = np.array(tok[p:p + 5 * nt], dtype=np.int64).reshape(nt, 5); p += 5 * nt
nb = int(tok[p]); p += 1
b = np.array(tok[p:p + 4 * nb], dtype=np.int64).reshape(nb, 4); p += 4 * nb
assert p == len(tok)
np.savez_compressed("parelag_amd/data/cube456.npz",
                    vertices=verts,
                    tets=(t[:, 1:] - 1).astype(np.int32), tet_attr=t[:, 0].astype(np.int32),
                    bdr_tris=(b[:, 1:] - 1).astype(np.int32), bdr_attr=b[:, 0].astype(np.int32))
print(nv, nt, nb)
