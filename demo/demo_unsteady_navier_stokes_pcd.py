#!/usr/bin/env python3
"""Unsteady Navier-Stokes (backward Euler) with PCD / PCDR preconditioning.

Counterpart of the reference's ``demo/unsteady-navier-stokes-pcd/
demo_unsteady-navier-stokes-pcd.py`` and ``...-pcdr.py`` (flags ``-l --nu
--pcd --nls --dt --t_end``; ``--pcdr`` switches to the PCDR variant).  Prints
the same summary table as the reference (``:214-222``), whose published values
for level 4 are 3157 Krylov iterations (PCD) and 1686 (PCDR) over 25 steps
(``demo/unsteady-navier-stokes-pcd/documentation.rst:134-140``)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fenapack_amd import PETScOptions                                 # noqa
from fenapack_amd.driver import (default_inner_options,               # noqa
                                 multigrid_inner_options, solve_unsteady)
from fenapack_amd.fem import BackwardStep                            # noqa

p = argparse.ArgumentParser(description=__doc__)
p.add_argument("-l", type=int, dest="level", default=4)
p.add_argument("--nu", type=float, dest="viscosity", default=0.02)
p.add_argument("--pcd", dest="pcd_variant", default="BRM1",
               choices=["BRM1", "BRM2"])
p.add_argument("--pcdr", action="store_true")
p.add_argument("--nls", default="picard", choices=["picard", "newton"])
p.add_argument("--ls", default="mg", choices=["mg", "jacobi"])
p.add_argument("--cycles", type=int, default=2,
               help="multigrid cycles per inner solve (--ls mg)")
p.add_argument("--dirichlet-diag", default="multiplicity",
               choices=["unit", "multiplicity"],
               help="diagonal of Dirichlet rows: 1, or the number of cells "
                    "sharing the dof as DOLFIN's SystemAssembler produces")
p.add_argument("--producer", default="host", choices=["host", "device"],
               help="host (numpy) producer or the engine's device producer "
                    "(pcd_fe_*; Picard, --ls mg)")
p.add_argument("--mg-coarse", default=None,
               choices=["galerkin", "rediscretize", "rediscretize-supg"],
               help="coarse velocity operators (default: galerkin)")
p.add_argument("--dt", type=float, default=0.2)
p.add_argument("--t_end", type=float, default=5.0)
args = p.parse_args()

# several GPUs: start it as the reference is started under `mpirun -np N`
# (test/regression/test.py:186-195), here one process per GPU:
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
#       --master-addr 127.0.0.1 demo/demo_unsteady_navier_stokes_pcd.py ...
# torch.distributed only carries the RCCL unique id (fenapack_amd/parallel.py)
world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0"))
solver_kw = {}
if world > 1:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from fenapack_amd.parallel import Comm
    solver_kw = {"comm": Comm.world(), "device": local}
    if rank:
        sys.stdout = open(os.devnull, "w")      # rank 0 reports

if args.mg_coarse is None:
    args.mg_coarse = "galerkin"
pb = BackwardStep(args.level, nu=args.viscosity, variant=args.pcd_variant,
                  nls=args.nls, dt=args.dt, pcdr=args.pcdr,
                  dirichlet_diag=args.dirichlet_diag,
                  coarse_stabilize=args.mg_coarse == "rediscretize-supg")
print("Reynolds number: Re = %g" % (2.0 / pb.nu))
print("Dimension of the function space: %d" % pb.space.ndof)
PETScOptions.clear()
if args.ls == "mg":
    multigrid_inner_options(cycles_u=args.cycles, cycles_p=args.cycles,
                            pcdr=args.pcdr,
                            galerkin_u=args.mg_coarse == "galerkin")
else:
    default_inner_options(a00_its=30, a00_ratio=0.03, ap_rtol=1e-10,
                          pcdr=args.pcdr)
if args.producer == "device":
    from fenapack_amd.device_producer import solve_unsteady_device
    solve_unsteady = solve_unsteady_device
out = solve_unsteady(pb, dt=args.dt, t_end=args.t_end, newton_rtol=1e-5,
                     gmres_rtol=1e-6, **solver_kw)
tab = "{:^15} | {:^15} | {:^15} | {:^19} | {:^15}\n".format(
    "No. of DOF", "Steps", "Krylov its", "Krylov its (p.t.s.)", "Time (s)")
tab += "{:>9}       | {:^15} | {:^15} | {:^19.1f} | {:^15.2f}\n".format(
    out["ndof"], out["steps"], out["krylov_its"],
    float(out["krylov_its"]) / out["steps"], out["time"])
print("\nSummary of iteration counts:")
print(tab)
print("Krylov iterations per time step:", out["krylov_per_step"])
for k, (its, res) in enumerate(zip(out["krylov_per_newton"], out["residuals"])):
    print("step %2d  GMRES its per Picard iteration %s  |F| %s"
          % (k + 1, its, ["%.2e" % r for r in res]))
if world > 1:
    dist.destroy_process_group()
