#!/usr/bin/env python3
"""Cost of one outer GMRES iteration (PCApply + system SpMV + classical
Gram-Schmidt + Hessenberg column to the host), eager vs hipGraph PCApply."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                            # noqa
from fenapack_amd import PETScOptions, _cabi as c                       # noqa
from fenapack_amd.driver import make_solver, multigrid_inner_options    # noqa
from fenapack_amd.fem import Cavity                                     # noqa

level = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pb = Cavity(level, nu=0.01)
PETScOptions.clear()
multigrid_inner_options(dim=2)
w, nls, nlp = make_solver(pb, max_newton=2)
nls.parameters["error_on_nonconvergence"] = False
nls.solve(nlp, w.vector(), on_update=w.touch)
eng = nls.linear_solver().ksp().engine
n = pb.space.ndof
b = torch.randn(n, dtype=torch.float64, device="cuda")
x = torch.zeros_like(b)
for graph in (False, True):
    eng.graph_enable(graph)
    for its in (10, 40, 100):
        eng.gmres_solve(b, x, c.MEM_DEVICE, 1e-30, 0.0, 150, its)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, _ = eng.gmres_solve(b, x, c.MEM_DEVICE, 1e-30, 0.0, 150, its)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("graph %-5s %3d iterations: %.3f ms per iteration"
              % (graph, got, 1e3 * dt / got))
x.zero_()
y = torch.empty_like(b)
for graph in (False, True):
    eng.graph_enable(graph)
    eng.fieldsplit_apply(b, y, c.MEM_DEVICE)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.fieldsplit_apply(b, y, c.MEM_DEVICE)
    torch.cuda.synchronize()
    print("graph %-5s PCApply %.3f ms" % (graph, 10 * (time.perf_counter() - t0)))
