"""Element partitioners that feed ``DeRhamSequence.Coarsen`` (host integer work, like the reference's ``src/partitioning``).

``GeometricBoxPartitioner`` follows ``src/partitioning/GeometricBoxPartitioner.cpp:20-80``; ``LogicalPartitioner`` with the Cartesian /
material-id coarsening operators ``src/partitioning/LogicalPartitioner.hpp:41-135`` and ``CartesianPartitioner.hpp:25-152`` (any other
"coarse logical value" works the same way: see the derefinement + material operator in ``tests/test_logical_partitioner_golden.py``);
the plain Cartesian box partition of a structured grid is ``amge_level.box_partition``; the derefinement partition of a refined
tet mesh is ``meshgen.Sequence.partition`` / ``linalg.partition_refined``.  The graph partitioners (METIS / ParMETIS) are
third-party and out of scope.  Known answers of the reference reproduced with these: ``test_GeometricBoxPartitioner``,
``UpscalingGeneralForm --geometric``, ``LogicalPartitionerDemo``, ``EmbeddedMeshPartitionerDemo``."""
from __future__ import annotations

import numpy as np


class GeometricBoxPartitioner:
    """Cuts the bounding box of the mesh into ``round(extent_d / r)`` slabs per direction, ``r`` the edge of a cube of the
    target volume, and bins the elements by the mean of their vertices.  Empty or disconnected boxes are left to
    ``connected_components`` / the topology checks downstream, as in the reference."""

    def __init__(self):
        self.partition_sizes_ = np.zeros(0, dtype=np.int64)

    def doPartition(self, vertices: np.ndarray, elem_vertices: np.ndarray, num_partitions: int) -> np.ndarray:
        vertices = np.asarray(vertices, dtype=np.float64)
        elem_vertices = np.asarray(elem_vertices)
        if vertices.ndim != 2 or elem_vertices.ndim != 2 or num_partitions < 1:
            raise ValueError("GeometricBoxPartitioner: vertices (nv, dim), elem_vertices (ne, k), num_partitions >= 1")
        dim = vertices.shape[1]
        bmin, bmax = vertices.min(axis=0), vertices.max(axis=0)
        target_radius = (np.prod(bmax - bmin) / float(num_partitions)) ** (1.0 / dim)
        ndir = ((bmax - bmin) / target_radius + 0.5).astype(np.int64)            # the implicit floor of :38-39
        if np.any(ndir < 1):
            raise ValueError("GeometricBoxPartitioner: a direction of the bounding box gets no partition")
        radius = (bmax - bmin) / ndir
        centre = vertices[elem_vertices].mean(axis=1)
        which = ((centre - bmin) / radius).astype(np.int64)
        stride = np.concatenate([[1], np.cumprod(ndir[:-1])])
        part = which @ stride
        self.partition_sizes_ = np.bincount(part, minlength=int(np.prod(ndir)))
        return part.astype(np.int64)

    def GetPartitionSizes(self) -> np.ndarray:
        return self.partition_sizes_


# ---- logical (Cartesian / material) partitioning -------------------------------------------------------------------------------

class CartesianIJK:
    """``CartesianIJK`` (``src/partitioning/CartesianPartitioner.hpp:41-103``): logical (i, j, k) of the elements of a structured
    grid; only the XYZ ordering exists there (element ``k N0 N1 + j N0 + i``).  ``order="generator"`` gives the same table for
    the numbering of ``seq_level_from_hex`` (element ``(i N1 + j) N2 + k``)."""

    @staticmethod
    def SetupCartesianIJK(N, order: str = "XYZ") -> np.ndarray:
        N = [int(v) for v in N]
        e = np.arange(int(np.prod(N)), dtype=np.int64)
        if order == "XYZ":
            cols = [e % N[0]] + ([(e // N[0]) % N[1]] if len(N) > 1 else []) + ([e // (N[0] * N[1])] if len(N) > 2 else [])
        elif order == "generator":
            cols = [e // int(np.prod(N[1:]))] + ([(e // N[2]) % N[1], e % N[2]] if len(N) == 3 else [e % N[1]] if len(N) == 2 else [])
        else:
            raise ValueError("CartesianIJK only works for XYZ ordering")
        return np.stack(cols, 1)

    @staticmethod
    def SetupCartesianIJKMaterialId(N, material_id, order: str = "XYZ") -> np.ndarray:
        ijk = CartesianIJK.SetupCartesianIJK(N, order)
        return np.concatenate([ijk, np.asarray(material_id, dtype=np.int64).reshape(-1, 1)], 1)


class CoarsenLogicalCartesianOperator:
    """``operator()`` of ``CoarsenLogicalCartesianOperator[MaterialId]`` (``CartesianPartitioner.hpp:117-152``): the logical
    indices divided by the coarsening ratio, further columns (the material id) carried over."""

    def __init__(self, coarseningratio):
        self.ratio = np.asarray(coarseningratio, dtype=np.int64)

    def __call__(self, fine: np.ndarray) -> np.ndarray:
        fine = np.asarray(fine, dtype=np.int64)
        out = fine.copy()
        out[..., :self.ratio.size] //= self.ratio
        return out


CoarsenLogicalCartesianOperatorMaterialId = CoarsenLogicalCartesianOperator


class LogicalPartitioner:
    """``LogicalPartitioner`` (``src/partitioning/LogicalPartitioner.hpp:41-135``): a partition is a connected set of elements with
    the same coarse logical value, partitions numbered in the order of their first element."""

    def Partition(self, elem_elem, fine_logical: np.ndarray, coarsening_logical) -> np.ndarray:
        from . import linalg
        coarse = np.asarray(coarsening_logical(fine_logical))
        _u, label = np.unique(coarse.reshape(coarse.shape[0], -1), axis=0, return_inverse=True)
        _n, comp = linalg.connected_components(label.reshape(-1), elem_elem)            # connected pieces of equal labels
        _u, first, inv = np.unique(comp, return_index=True, return_inverse=True)
        rank = np.empty(first.size, dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(first.size)                 # the stack search meets them in element order
        return rank[inv.reshape(-1)]

    def ComputeCoarseLogical(self, coarsen_map, AE_element, fine_logical: np.ndarray) -> np.ndarray:
        """The coarse logical value of an agglomerate = that of its first element (``:108-129``)."""
        ptr, idx = AE_element[0], AE_element[1]
        if np.any(np.diff(ptr) == 0):
            raise ValueError("ComputeCoarseLogical: empty agglomerate")
        return np.asarray(coarsen_map(np.asarray(fine_logical)[idx[ptr[:-1]]]))
