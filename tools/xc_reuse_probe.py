#!/usr/bin/env python3
"""Column reuse per window and the measured SpMV forms of the cube456 x<k> operators, renumbered (hierarchy) and in the caller's
numbering (ELAG_HIP_XC_VERBOSE=1 prints both): the data behind the rule that replaces the setup-time measurement."""
import os, sys
os.environ["ELAG_HIP_XC_VERBOSE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from parelag_amd import linalg as la, meshgen, _lib
_lib.init()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
A, Ps, _ = meshgen.h1_problem(k)
dA = la.ParallelCSRMatrix.from_scipy(A)
dPs = [la.ParallelCSRMatrix.from_scipy(P) for P in Ps]
print("== hierarchy (renumbered operators)", flush=True)
H = la.buildHierarchyFromDeRhamSequence(dA, dPs, "L1 Jacobi")
print("== caller's numbering", flush=True)
x = la.DeviceVector(A.shape[0], 1.0); y = la.DeviceVector(A.shape[0])
dA.Mult(x, y)
la.synchronize()
