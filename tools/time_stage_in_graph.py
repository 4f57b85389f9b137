#!/usr/bin/env python3
"""What one small dependent stage costs inside the replayed graph: the
fieldsplit PCApply of a cavity level with Chebyshev(its) on M_p, its = 5 and
45, eager and replayed; (t45 - t5) / 40 = one k_cheb_step_s launch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                            # noqa
from fenapack_amd import PETScOptions, _cabi as c                       # noqa
from fenapack_amd.driver import make_solver, multigrid_inner_options    # noqa
from fenapack_amd.fem import Cavity                                     # noqa

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pb = Cavity(level, nu=0.01)
PETScOptions.clear()
multigrid_inner_options(dim=2)
w, nls, nlp = make_solver(pb, max_newton=1)
nls.parameters["error_on_nonconvergence"] = False
nls.solve(nlp, w.vector(), on_update=w.touch)
eng = nls.linear_solver().ksp().engine
n = pb.space.ndof
x = torch.randn(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)


def t(its, graph, reps=300):
    eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", its, 0.0, 0.5, 2.0)
    eng.graph_enable(graph)
    for _ in range(10):
        eng.fieldsplit_apply(x, y, c.MEM_DEVICE)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.fieldsplit_apply(x, y, c.MEM_DEVICE)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for graph in (False, True):
    t5, t45 = t(5, graph), t(45, graph)
    print("level %d (n_p %d) %s: PCApply %.1f us at its=5, %.1f us at its=45 -> %.2f us per "
          "extra Chebyshev step" % (level, pb.space.n_p,
                                     "graph replay" if graph else "eager", t5 * 1e6,
                                     t45 * 1e6, (t45 - t5) / 40 * 1e6))
