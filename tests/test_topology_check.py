"""The reference's own known-answer tests for the topology checks on the AMGe path: ``testsuite/twentyseven.cpp`` coarsens the
27-cell hex mesh ``mfem::Mesh(3, 3, 3, HEXAHEDRON)`` with hand-made partitionings and ctest greps its diagnostics
(``testsuite/CMakeLists.txt:35-92``):

    disconnected  ->  "N_elements =          3"          donut     ->  "Element 0 has 1 tunnels."
    void          ->  "Element 1 has 1 holes."            discface  ->  "Facet 3 is disconnected."
    facehole      ->  "Facet 3 has 1 holes."              sharededge->  "codim 0 iAE 1 has bad connectivity"
    discedge      ->  "Ridge 16 is disconnected."   (on ``Mesh(3, 3, 4, HEXAHEDRON)``)

These strings are golden outputs of the reference itself.  The facet numbers depend on MFEM's numbering of the
mesh, which ``mfem_hex_topology`` below restates: elements x-fastest, faces in the order ``GetElementToFaceTable`` first meets
them walking the elements and their local faces (bottom, front, right, back, left, top: ``hex_t::FaceVert``), boundary
attributes 1..6 in that order (``Mesh::Make3D``).  The CPU half runs the oracle's entity-by-entity checker (numpy SVD ranks), the GPU half the
product's (vectorised tables, ranks from the device kernel) and compares the two."""
import numpy as np
import pytest

from oracle import derham as od, oracle as orc
from parelag_amd import derham
from parelag_amd.derham_level import SeqLevel, seq_level_from_hex


def mfem_hex_topology(nx, ny, nz) -> SeqLevel:
    """The topology tables of ``seq_level_from_hex`` renumbered like MFEM numbers the same Cartesian mesh (elements, facets,
    ridges, boundary attributes); peaks keep the generator's numbers.  Ridges: ``Mesh::GetVertexToVertexTable`` pushes the
    edges of every element in ``hex_t::Edges`` order into a table that numbers a pair the first time it sees it."""
    lv = seq_level_from_hex((nx, ny, nz), jstart=2)
    # generator: element (i, j, k) -> (i * ny + j) * nz + k with local facets [x-, x+, y-, y+, z-, z+]
    my_elem = lambda i, j, k: (i * ny + j) * nz + k
    order_e = np.array([my_elem(i, j, k) for k in range(nz) for j in range(ny) for i in range(nx)], dtype=np.int64)   # mfem id -> mine
    loc = [4, 2, 1, 3, 0, 5]                              # mfem local face (z-, y-, x+, y+, x-, z+) -> generator slot
    attr_of_slot = {4: 1, 2: 2, 1: 3, 3: 4, 0: 5, 5: 6}
    fmap = -np.ones(lv.n[1], dtype=np.int64)              # my facet -> mfem facet
    nf = 0
    rows_idx, rows_sgn = [], []
    for e in order_e:
        q0 = lv.B[0][0][e]
        for s_ in loc:
            f = lv.B[0][1][q0 + s_]
            if fmap[f] < 0:
                fmap[f] = nf
                nf += 1
            rows_idx.append(fmap[f]); rows_sgn.append(lv.B[0][2][q0 + s_])
    assert nf == lv.n[1]
    inv_f = np.empty(nf, dtype=np.int64); inv_f[fmap] = np.arange(nf)          # mfem facet -> mine
    B0 = (6 * np.arange(lv.n[0] + 1, dtype=np.int64), np.array(rows_idx, dtype=np.int64), np.array(rows_sgn))
    p1, i1, s1 = lv.B[1]
    cnt = np.diff(p1)[inv_f]
    src = np.concatenate([np.arange(p1[f], p1[f + 1]) for f in inv_f])
    B1 = (np.r_[0, np.cumsum(cnt)].astype(np.int64), i1[src], s1[src])
    # boundary attributes by side
    attr = np.zeros(nf, dtype=np.int32)
    for e in range(lv.n[0]):
        q0 = lv.B[0][0][e]
        for s_ in range(6):
            f = lv.B[0][1][q0 + s_]
            if lv.facet_attr[f] > 0:
                attr[fmap[f]] = attr_of_slot[s_]
    # ridges: first touch over mfem elements x hex_t::Edges, local vertices 0..7 = (0,0,0) (1,0,0) (1,1,0) (0,1,0) + the same at z+1
    vid = lambda i, j, k: (i * (ny + 1) + j) * (nz + 1) + k
    lvert = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
    ledge = [(0, 1), (1, 2), (3, 2), (0, 3), (4, 5), (5, 6), (7, 6), (4, 7), (0, 4), (1, 5), (2, 6), (3, 7)]
    p2, i2, _s2 = lv.B[2]
    pair_to_ridge = {tuple(sorted(i2[p2[r]:p2[r + 1]])): r for r in range(lv.n[2])}
    rmap = -np.ones(lv.n[2], dtype=np.int64)              # my ridge -> mfem ridge
    nr = 0
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for a, b in ledge:
                    va, vb = (vid(i + o[0], j + o[1], k + o[2]) for o in (lvert[a], lvert[b]))
                    r = pair_to_ridge[tuple(sorted((va, vb)))]
                    if rmap[r] < 0:
                        rmap[r] = nr
                        nr += 1
    assert nr == lv.n[2]
    inv_r = np.empty(nr, dtype=np.int64); inv_r[rmap] = np.arange(nr)
    B1 = (B1[0], rmap[B1[1]], B1[2])
    src2 = np.concatenate([np.arange(p2[r], p2[r + 1]) for r in inv_r])
    B2 = (2 * np.arange(nr + 1, dtype=np.int64), i2[src2], lv.B[2][2][src2])
    return SeqLevel(n=lv.n, B=[B0, B1, B2], facet_attr=attr, jstart=2, idof=lv.idof, M=[None] * 4, D=[None] * 3, T=[None] * 4,
                    pv=[None] * 4, const3=lv.const3)


def partitioning(name):
    """``Make*Partitioning`` of ``testsuite/twentyseven.cpp:36-141`` (the element index is MFEM's)."""
    if name == "discedge":                                # on the 3 x 3 x 4 mesh
        p = np.full(36, 4, dtype=np.int64)
        p[9:13] = 0; p[18:22] = 1
        p[14:18] = 2; p[23:27] = 3
        p[:9] = 0; p[27:36] = 3
        return p
    p = np.ones(27, dtype=np.int64)
    if name == "disconnected":
        p[0] = 0; p[26] = 0
    elif name == "donut":
        for i in range(3):
            for j in range(3):
                p[9 * i + 3 * j + 1] = 0
        p[13] = 1
    elif name == "void":
        p[13] = 0
    elif name == "discface":
        p[:] = 0
        p[:9] = 1
        p[12:15] = 2
    elif name == "facehole":
        p[:] = 2
        p[:9] = 0
        p[13] = 1
    elif name == "sharededge":
        p[:] = 0
        p[[0, 4, 5, 9, 14, 18, 21, 22, 23]] = 1
    elif name == "simple":
        p[:9] = 0
    return p


def run(name, product_ranks):
    lv = mfem_hex_topology(3, 3, 4 if name == "discedge" else 3)
    part = partitioning(name)
    # CoarsenLocalPartitioning starts with connectedComponents on the element-element table (Topology.cpp:713-721)
    import scipy.sparse as sp
    B0 = sp.csr_matrix((np.abs(lv.B[0][2]), lv.B[0][1], lv.B[0][0]), shape=(lv.n[0], lv.n[1]))
    ncomp, part = orc.connected_components(part.astype(np.int32), (B0 @ B0.T).tocsr())
    topo = derham.coarsen_topology(lv, np.asarray(part, dtype=np.int64))
    topo_o = od.coarsen_topology(lv, np.asarray(part, dtype=np.int64))
    same_topology(topo, topo_o)
    res = derham.check_topology(lv, topo) if product_ranks else od.check_topology(lv, topo_o)
    return ncomp, topo, res


def same_topology(a, b):
    assert a["n"] == b["n"]
    for c in range(4):
        for k in range(3):
            assert np.array_equal(a["A"][c][k], b["A"][c][k])
    for c in range(3):
        for k in range(3):
            assert np.array_equal(a["B"][c][k], b["B"][c][k])
    assert np.array_equal(a["facet_attr"], b["facet_attr"])


def checked(name, product):
    """``CoarsenLocalPartitioning(partitioning, check_topology = true, ...)`` as ``twentyseven.cpp:222, 312`` calls it."""
    import scipy.sparse as sp
    lv = mfem_hex_topology(3, 3, 4 if name == "discedge" else 3)
    B0 = sp.csr_matrix((np.abs(lv.B[0][2]), lv.B[0][1], lv.B[0][0]), shape=(lv.n[0], lv.n[1]))
    _n, part = orc.connected_components(partitioning(name).astype(np.int32), (B0 @ B0.T).tocsr())
    mod = derham if product else od
    return lv, mod.coarsen_topology(lv, np.asarray(part, dtype=np.int64), check=True)


def boundary_of_boundary_is_zero(topo):
    import scipy.sparse as sp
    B = [sp.csr_matrix((topo["B"][c][2], topo["B"][c][1], topo["B"][c][0]), shape=(topo["n"][c], topo["n"][c + 1])) for c in range(3)]
    for c in range(2):
        BB = B[c] @ B[c + 1]
        assert BB.nnz == 0 or np.abs(BB.data).max() < 1e-12                        # twentyseven.cpp:331-337
        assert np.abs(B[c].data).max() == 1.0


EXPECT = {"donut": "Element 0 has 1 tunnels.", "void": "Element 1 has 1 holes.", "discface": "Facet 3 is disconnected.",
          "facehole": "Facet 3 has 1 holes.", "discedge": "Ridge 16 is disconnected.", "sharededge": "codim 0 iAE 1 has bad connectivity"}


@pytest.mark.parametrize("name", sorted(EXPECT))
def test_reference_diagnostics_cpu(name):
    _n, _topo, res = run(name, product_ranks=False)
    assert any(EXPECT[name] in m for m in res["messages"]), res["messages"]


def test_disconnected_partition_is_split_into_three():
    ncomp, topo, res = run("disconnected", product_ranks=False)
    assert ncomp == 3 and topo["n"][0] == 3
    lv = mfem_hex_topology(3, 3, 3)
    text = derham.show_me(topo, weights=[np.ones(lv.n[c]) for c in range(3)])
    assert "N_elements =          3" in text                      # the line ctest greps (testsuite/CMakeLists.txt:35-40)
    assert "Euler Characteristic =          1         1" in text and "Element 1 25 9" in text
    assert not res["isbad"][0].any()


def test_simple_partition_is_clean():
    _n, _topo, res = run("simple", product_ranks=False)
    assert res["messages"] == [] and not any(v.any() for v in res["isbad"].values())
    assert np.array_equal(res["betti"][0], [[1, 0, 0], [1, 0, 0]])


@pytest.mark.parametrize("name", sorted(EXPECT) + ["simple"])
def test_checked_coarsening_repairs_the_topology_cpu(name):
    """With the check on (the reference's default) the same lines are printed while the topology is built, the bad entities
    are split into their members, and what comes out passes the checks and has B_c B_{c+1} = 0."""
    lv, topo = checked(name, product=False)
    if name in EXPECT:
        assert any(EXPECT[name] in m for m in topo["messages"]), topo["messages"]
    else:
        assert topo["messages"] == []
    boundary_of_boundary_is_zero(topo)
    again = od.check_topology(lv, topo)
    assert not any(v.any() for v in again["isbad"].values()), again["messages"]
    assert topo["n"][0] - topo["n"][1] + topo["n"][2] - topo["n"][3] == -1          # Euler characteristic of the ball (ShowMe)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(EXPECT) + ["simple"])
def test_checked_coarsening_on_device_matches_oracle(name):
    lv, topo = checked(name, product=True)
    _lv, topo_o = checked(name, product=False)
    same_topology(topo, topo_o)
    assert topo["messages"] == topo_o["messages"]
    assert not any(v.any() for v in derham.check_topology(lv, topo)["isbad"].values())


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(EXPECT) + ["simple", "disconnected"])
def test_reference_diagnostics_device(name):
    _n, _topo, dev = run(name, product_ranks=True)
    _n, _topo, ref = run(name, product_ranks=False)
    assert dev["messages"] == ref["messages"]
    for c in range(3):
        assert np.array_equal(dev["betti"][c], ref["betti"][c]) and np.array_equal(dev["isbad"][c], ref["isbad"][c])
    if name in EXPECT:
        assert any(EXPECT[name] in m for m in dev["messages"])


@pytest.mark.gpu
def test_betti_numbers_of_bad_unstructured_agglomerates_on_device():
    """Agglomerates cut by coordinate bisection through a tet mesh are not all balls / discs: the device's Betti numbers equal
    numpy's on every agglomerated element, facet and ridge, and flag the entities that make the lower forms fail."""
    from parelag_amd import meshgen
    from parelag_amd.derham_level import seq_level_from_tets
    m = meshgen.Sequence.kuhn_cube(2, 1).meshes[0]
    lv = seq_level_from_tets(m, jstart=2)
    cent = m.vertices[m.tets].mean(1)
    part = (np.floor(cent[:, 0] * 3).clip(0, 2) * 4 + np.floor(cent[:, 1] * 2).clip(0, 1) * 2 + np.floor(cent[:, 2] * 2).clip(0, 1)).astype(np.int64)
    _u, part = np.unique(part, return_inverse=True)
    import scipy.sparse as sp
    B0 = sp.csr_matrix((np.abs(lv.B[0][2]), lv.B[0][1], lv.B[0][0]), shape=(lv.n[0], lv.n[1]))
    _n, part = orc.connected_components(part.astype(np.int32), (B0 @ B0.T).tocsr())
    topo = derham.coarsen_topology(lv, np.asarray(part, dtype=np.int64))
    dev, ref = derham.check_topology(lv, topo), od.check_topology(lv, topo)
    for c in range(3):
        assert np.array_equal(dev["betti"][c], ref["betti"][c]) and np.array_equal(dev["isbad"][c], ref["isbad"][c])
    assert dev["messages"] == ref["messages"] and len(dev["messages"]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["donut", "void", "facehole", "sharededge", "simple"])
def test_sequence_coarsens_on_the_repaired_topology(name):
    """The tail of ``twentyseven.cpp:353-383``: all four forms (jFormStart = 0, SVD tolerance 1e-9) are coarsened on the
    agglomerated topology the checked coarsening produced, and the result satisfies the sequence's invariants."""
    import scipy.sparse as sp
    lv = seq_level_from_hex((3, 3, 3), jstart=0)
    # the partitionings are written for MFEM's element order (x fastest); the generator's is z fastest
    mine = np.array([(i * 3 + j) * 3 + k for k in range(3) for j in range(3) for i in range(3)])
    part = np.empty(27, dtype=np.int64)
    part[mine] = partitioning(name)
    B0 = sp.csr_matrix((np.abs(lv.B[0][2]), lv.B[0][1], lv.B[0][0]), shape=(lv.n[0], lv.n[1]))
    _n, part = orc.connected_components(part.astype(np.int32), (B0 @ B0.T).tocsr())
    fine = derham.DeRhamSequence(lv)
    fine.SetjformStart(0)
    fine.SetTargets(lv.T)
    fine.SetSVDTol(1e-9)
    coarse = fine.Coarsen(np.asarray(part, dtype=np.int64), check_topology=True)
    if name in EXPECT:
        key = EXPECT[name].split(" ")[0]                       # entity numbers depend on the mesh numbering; the kind of defect does not
        assert any(key in m for m in fine.TopologyMessages), fine.TopologyMessages
    else:
        assert fine.TopologyMessages == []
    defects = fine.CheckInvariants()
    assert max(defects.values()) < 1e-9, defects
    assert all(coarse.GetNumberOfDofs(j) > 0 for j in range(4))
