#!/usr/bin/env python3
"""Steady incompressible Navier-Stokes with PCD preconditioning on MI355X.

Counterpart of the reference's ``demo/navier-stokes-pcd/
demo_navier-stokes-pcd.py`` (same flags ``-l --nu --pcd --nls``), on the
reference's L-shaped backward-facing step or on the lid-driven cavity that
BASELINE.json names.  Inner solvers are the device-native ones (Jacobi-CG,
Chebyshev-Jacobi); ``--ls direct`` of the reference has no counterpart."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fenapack_amd.driver import (default_inner_options,               # noqa
                                 multigrid_inner_options, solve_steady)
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D          # noqa

p = argparse.ArgumentParser(description=__doc__)
p.add_argument("-l", type=int, dest="level", default=4)
p.add_argument("--nu", type=float, dest="viscosity", default=None)
p.add_argument("--pcd", dest="pcd_variant", default="BRM1",
               choices=["BRM1", "BRM2"])
p.add_argument("--nls", default="picard", choices=["picard", "newton"])
p.add_argument("--geometry", default="lshape",
               choices=["lshape", "cavity", "cube"])
p.add_argument("--n0", type=int, default=4,
               help="cube: cells per side of the coarsest mesh")
p.add_argument("--ls", default="mg", choices=["mg", "jacobi"],
               help="inner solvers: multigrid V-cycles (counterpart of the "
                    "reference's 'iterative' AMG setting) or plain "
                    "Jacobi-CG / Chebyshev-Jacobi")
p.add_argument("--stabilize", action="store_true",
               help="SUPG-stabilised 00-block in the preconditioner matrix "
                    "(the reference's J_pc for --ls iterative)")
p.add_argument("--cycles", type=int, default=1)
p.add_argument("--smooth", type=int, default=2)
p.add_argument("--mg-coarse", default="galerkin",
               choices=["galerkin", "rediscretize"],
               help="coarse velocity operators of the multigrid cycle")
p.add_argument("--producer", default="host", choices=["host", "device"],
               help="who re-assembles the iterate-dependent operators: the "
                    "numpy producer, or the engine itself in HBM (pcd_fe_*; "
                    "Picard, --ls mg)")
p.add_argument("--a00-its", type=int, default=60)
p.add_argument("--a00-ratio", type=float, default=0.01)
args = p.parse_args()

if args.geometry == "lshape":
    pb = BackwardStep(args.level, nu=args.viscosity or 0.02,
                      variant=args.pcd_variant, nls=args.nls,
                      stabilize=args.stabilize)
    print("Reynolds number: Re = %g" % (2.0 / pb.nu))
elif args.geometry == "cavity":
    pb = Cavity(args.level, nu=args.viscosity or 0.01,
                variant=args.pcd_variant, nls=args.nls,
                stabilize=args.stabilize)
    print("Reynolds number: Re = %g" % (1.0 / pb.nu))
else:
    pb = Cavity3D(args.level, nu=args.viscosity or 0.01, n0=args.n0,
                  variant=args.pcd_variant, nls=args.nls,
                  stabilize=args.stabilize)
    print("Reynolds number: Re = %g" % (1.0 / pb.nu))
print("Dimension of the function space: %d" % pb.space.ndof)
if args.ls == "mg":
    multigrid_inner_options(cycles_u=args.cycles, cycles_p=args.cycles,
                            smooth=args.smooth, dim=pb.space.dim,
                            galerkin_u=args.mg_coarse == "galerkin")
else:
    default_inner_options(a00_its=args.a00_its, a00_ratio=args.a00_ratio,
                          dim=pb.space.dim)
if args.producer == "device":
    from fenapack_amd.device_producer import solve_steady_device
    out = solve_steady_device(pb, max_newton=25)
else:
    out = solve_steady(pb, max_newton=25)
print("Newton iterations: %d, converged: %s" % (out["newton_its"],
                                                out["converged"]))
print("GMRES iterations per Newton step:", out["krylov_per_step"])
print("residuals:", ["%.3e" % r for r in out["residuals"]])
print("solve time: %.2f s" % out["time"])
if args.producer == "device":
    print("  of which: plans %.2f s, device Picard loop %.3f s "
          "(host-driven loop: GMRES %.2f s, producer %s)"
          % (out["time_plan"], out["time_device_loop"], out["time_gmres"],
             {k: round(v, 3) for k, v in out["producer_timing"].items()}))
