#!/usr/bin/env python3
"""cProfile of a steady Picard solve (host producer + device engine): where
the wall time of one nonlinear step goes."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd.driver import multigrid_inner_options, solve_steady  # noqa
from fenapack_amd.fem import Cavity                                    # noqa

level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pb = Cavity(level, nu=0.01)
multigrid_inner_options(dim=2)
pr = cProfile.Profile()
pr.enable()
out = solve_steady(pb, max_newton=25)
pr.disable()
print({k: v for k, v in out.items() if not hasattr(v, "shape")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
