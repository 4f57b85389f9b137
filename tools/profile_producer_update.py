#!/usr/bin/env python3
"""Ten operator updates of the device producer (for rocprofv3 --stats): which
kernels a nonlinear step's refresh consists of and what they cost."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import PETScOptions                                   # noqa
from fenapack_amd.device_producer import solve_steady_device            # noqa
from fenapack_amd.driver import multigrid_inner_options                 # noqa
from fenapack_amd.fem import Cavity, Cavity3D                           # noqa

geometry = sys.argv[1] if len(sys.argv) > 1 else "cavity"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
galerkin = (sys.argv[3] if len(sys.argv) > 3 else "galerkin") == "galerkin"
pb = Cavity(level, nu=0.01) if geometry == "cavity" \
    else Cavity3D(level, nu=0.01, n0=4)
PETScOptions.clear()
multigrid_inner_options(dim=pb.space.dim, galerkin_u=galerkin)
out = solve_steady_device(pb, max_newton=2)
prod, V = out["producer"], pb.space
x = out["w"].vector()
import time                                                             # noqa
t0 = time.perf_counter()
for _ in range(10):
    prod.update(x[V.is_u], x[V.is_p])
print("10 updates: %.1f ms each (host pieces included)"
      % (100 * (time.perf_counter() - t0)))
# the device-resident form of the same step: refresh + residual in one call
prod.set_time_level()
xm = np.ascontiguousarray(x)
prod.residual(xm)
t0 = time.perf_counter()
for _ in range(10):
    prod.residual(xm)
print("10 device residual calls (refresh of every level + residual): "
      "%.2f ms each" % (100 * (time.perf_counter() - t0)))
