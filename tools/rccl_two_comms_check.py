#!/usr/bin/env python3
"""One process, two RCCL communicators from ONE library copy: torch's own
process group (backend nccl, world size 1) and the engine's
(PCD_FORCE_COMM=1) - what every rank of `bench.py --gpus N` holds.  Runs a
small device-producer solve through the partitioned code path and exits;
the exit code is the check (two RCCL copies abort at exit)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["PCD_FORCE_COMM"] = "1"
import torch                                                       # noqa
import torch.distributed as dist                                   # noqa

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1,
                        device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda")
dist.all_reduce(t)
from fenapack_amd import PETScOptions                              # noqa
from fenapack_amd.device_producer import solve_steady_device      # noqa
from fenapack_amd.driver import multigrid_inner_options            # noqa
from fenapack_amd.fem import Cavity                                # noqa
from fenapack_amd.parallel import Comm                             # noqa

PETScOptions.clear()
multigrid_inner_options(dim=2)
out = solve_steady_device(Cavity(4, nu=0.01), max_newton=8, comm=Comm.world())
print("ranks", out["producer"].ranks, "converged", out["converged"],
      "krylov", out["krylov_per_step"])
out["solver"].linear_solver().ksp().engine.destroy()
dist.destroy_process_group()
print("clean exit")
