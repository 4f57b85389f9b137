#!/usr/bin/env python3
"""Per-launch cost of a small, latency-bound kernel chain: Chebyshev(its) +
Jacobi on the cavity's pressure mass matrix (level 6: 103 041 rows, 7 entries
per row) - the five 4.8-us launches of the benchmark's M_p solve.  Used to
price storage-format experiments (FENAPACK_AMD_HIP_LIB=<variant build>)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
from fenapack_amd import _cabi as c                             # noqa: E402
from fenapack_amd.fem import Cavity                              # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pb = Cavity(level, nu=0.01)
Mp = pb.Mp
e = c.Engine(c.hip_library(), "BRM1", 0)
e.set_csr(c.MAT_MP, Mp)
b = torch.randn(Mp.shape[0], dtype=torch.float64, device="cuda")
x = torch.empty_like(b)


def t(its, reps=200):
    e.set_inner(c.KSP_MP, "chebyshev", "jacobi", its, 0.0, 0.5, 2.0)
    for _ in range(5):
        e.inner_solve(c.KSP_MP, b, x, c.MEM_DEVICE)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            e.inner_solve(c.KSP_MP, b, x, c.MEM_DEVICE)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


t25, t5 = t(25), t(5)
print("level %d n %d nnz %d (%.2f per row): %.2f us per Chebyshev step (%s)"
      % (level, Mp.shape[0], Mp.nnz, Mp.nnz / Mp.shape[0],
         (t25 - t5) / 20 * 1e6, os.environ.get("FENAPACK_AMD_HIP_LIB", "default")))
