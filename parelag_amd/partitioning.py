"""Element partitioners that feed ``DeRhamSequence.Coarsen`` (host integer work, like the reference's ``src/partitioning``).

``GeometricBoxPartitioner`` follows ``src/partitioning/GeometricBoxPartitioner.cpp:20-80``; the Cartesian box partition of a
structured grid is ``amge_level.box_partition``; the derefinement partition of a refined tet mesh is
``meshgen.Sequence.partition``.  The graph partitioners (METIS / ParMETIS) are third-party and out of scope."""
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
