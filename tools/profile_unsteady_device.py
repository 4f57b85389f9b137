#!/usr/bin/env python3
"""cProfile of the unsteady loop with the device producer (where does the host
time go at a small level?)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import PETScOptions                                    # noqa
from fenapack_amd.device_producer import solve_unsteady_device           # noqa
from fenapack_amd.driver import multigrid_inner_options                  # noqa
from fenapack_amd.fem import BackwardStep                                # noqa

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pb = BackwardStep(level, nu=0.02, dt=0.2, dirichlet_diag="multiplicity")
PETScOptions.clear()
multigrid_inner_options(cycles_u=2, cycles_p=2)
pr = cProfile.Profile()
pr.enable()
out = solve_unsteady_device(pb, dt=0.2, t_end=2.0, newton_rtol=1e-5,
                            gmres_rtol=1e-6)
pr.disable()
print(out["krylov_its"], out["time"], out["producer_timing"],
      out["time_gmres"])
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
