#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic of the dominant
kernel).  Launches, on the level-6 cavity matrices (analytic vortex wind, no
nonlinear solve needed for a byte count):

  * 20 x k_scale_dinv on n_u rows  - calibration: exactly 24*n_u bytes with the
    same 8-byte-per-lane access width as the stream kernels;
  * 20 x 8 fused Chebyshev steps on A00 (k_cheb_step_s) - the kernel priced.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- \
        python3 tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT2 -- \
        python3 tools/pmc_workload.py
then  python3 tools/pmc_summarise.py OUT OUT2
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _cabi as c                       # noqa: E402
from fenapack_amd.fem import Cavity, Cavity3D              # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cube = len(sys.argv) > 2 and sys.argv[2] == "cube"     # level = refinements
pb = Cavity3D(level, nu=0.01, n0=4) if cube else Cavity(level, nu=0.01)
V = pb.space
xy = V.node_coords
x, y = xy[:, 0], xy[:, 1]
U = np.stack([np.sin(np.pi * x) ** 2 * np.sin(np.pi * y) * np.cos(np.pi * y),
              -np.sin(np.pi * x) * np.cos(np.pi * x) * np.sin(np.pi * y) ** 2]
             + ([0.1 * np.sin(np.pi * xy[:, 2])] if cube else []), axis=1)
L = pb.linearise(U.ravel(), np.zeros(V.n_p))
e = c.Engine(c.hip_library(), "BRM1", 0)
e.set_velocity_block(V.dim)
e.set_csr(c.MAT_A00, L["A00"])
b = np.random.default_rng(0).standard_normal(V.n_u)
out = np.empty_like(b)
e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 0, 0.0, 0.2, 2.2)
for _ in range(20):
    e.inner_solve(c.KSP_A00, b, out)
e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 8, 0.0, 0.2, 2.2)
for _ in range(20):
    e.inner_solve(c.KSP_A00, b, out)
print("n_u", V.n_u, "nnz_A00", L["A00"].nnz)
