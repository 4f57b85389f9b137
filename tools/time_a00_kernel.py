#!/usr/bin/env python3
"""Time the fused Chebyshev step on the A00 of a cavity level (A/B switches:
PCD_NO_XCD_REMAP, PCD_NO_KRON2, PCD_FORCE_CSR_VECTOR, PCD_MAX_RB)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _cabi as c                       # noqa: E402
from fenapack_amd.fem import Cavity, Cavity3D              # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cube = len(sys.argv) > 2 and sys.argv[2] == "cube"     # level = refinements of n0 (default 4)
n0 = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cache = "/tmp/a00_%s_%d_%d.npz" % ("cube" if cube else "cavity", level, n0)
import scipy.sparse as sp                                  # noqa: E402
if os.path.exists(cache):
    # (the operator alone: no mesh, no spaces - a 10 M-DOF problem takes a
    # minute to build, a sweep over kernel variants asks for it many times)
    A00 = sp.load_npz(cache)
    dim = 3 if cube else 2
else:
    pb = Cavity3D(level, nu=0.01, n0=n0) if cube else Cavity(level, nu=0.01)
    V = pb.space
    x, y = V.node_coords[:, 0], V.node_coords[:, 1]
    U = np.stack([np.sin(np.pi * x) ** 2 * np.sin(2 * np.pi * y),
                  -np.sin(2 * np.pi * x) * np.sin(np.pi * y) ** 2]
                 + ([0.1 * np.sin(np.pi * V.node_coords[:, 2])] if cube else []),
                 axis=1)
    A00 = sp.csr_matrix(pb.linearise(U.ravel(), np.zeros(V.n_p))["A00"])
    sp.save_npz(cache, A00, compressed=False)
    dim = V.dim
    del pb, V, U
n_u = A00.shape[0]
import torch                                               # noqa: E402
e = c.Engine(c.hip_library(), "BRM1", 0)
e.set_velocity_block(dim)
e.set_csr(c.MAT_A00, A00)
b = torch.randn(n_u, dtype=torch.float64, device="cuda")
out = torch.empty_like(b)


def t(m, reps=10):
    e.set_inner(c.KSP_A00, "chebyshev", "jacobi", m, 0.0, 0.2, 2.2)
    e.inner_solve(c.KSP_A00, b, out, c.MEM_DEVICE)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            e.inner_solve(c.KSP_A00, b, out, c.MEM_DEVICE)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


us = (t(65) - t(1)) / 64 * 1e6
nbytes = 12 * A00.nnz + 92 * n_u + 4
print("level %d n_u %d nnz %d: %.2f us per launch, %.0f GB/s algorithmic (%s)"
      % (level, n_u, A00.nnz, us, nbytes / us / 1e3,
         " ".join("%s=%s" % (k, os.environ[k]) for k in
                  ("PCD_NO_XCD_REMAP", "PCD_NO_KRON2", "PCD_FORCE_CSR_VECTOR",
                   "PCD_MAX_RB", "PCD_MAX_CHUNKS", "PCD_MIN_WGS", "PCD_NO_COL16",
                   "PCD_XCD_REMAP_MAX_ROWS", "PCD_NT_BYTES", "PCD_XCD_REMAP_NT",
                   "PCD_VEC_TILE", "PCD_VT_NT", "PCD_NO_ROWKRON", "PCD_OVERLAP",
                   "FENAPACK_AMD_HIP_LIB") if k in os.environ) or "default"))
