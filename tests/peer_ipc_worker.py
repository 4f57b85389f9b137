#!/usr/bin/env python3
"""Worker of tests/test_peer_gpu.py: TWO PROCESSES on ONE GPU.  RCCL refuses
two ranks on one device; the engine's peer-write protocol (csrc/pcd_peer.hpp)
does not need it: the processes exchange the IPC handles of their arenas over
a host transport (torch.distributed / gloo here, pcd_comm_init_host) and every
halo exchange and dot product of the solve is then one kernel that stores
into the other process's arena.  Rank 0 writes the results."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

p = argparse.ArgumentParser()
p.add_argument("--out", required=True)
p.add_argument("--level", type=int, default=4)
p.add_argument("--fail-rank", type=int, default=-1)
a = p.parse_args()

import torch                                                  # noqa: E402
import torch.distributed as dist                              # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
# the ONE GPU - or, from tools/first_contact.py on a box with several
# (PCD_KIT_ONE_GPU_PER_RANK=1), a device per rank: the same arenas, mapped
# across devices
DEV = int(os.environ.get("LOCAL_RANK", "0")) \
    if os.environ.get("PCD_KIT_ONE_GPU_PER_RANK") == "1" else 0
torch.cuda.set_device(DEV)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")

from fenapack_amd import PETScOptions, _cabi as c            # noqa: E402
from fenapack_amd.driver import multigrid_inner_options, solve_steady  # noqa
from fenapack_amd.fem import Cavity                          # noqa: E402
from fenapack_amd.fem import partition as pt                 # noqa: E402
from fenapack_amd.parallel import Comm, TorchHostTransport   # noqa: E402

os.environ.setdefault("PCD_REPLICATE_BELOW", "1500")
transport = TorchHostTransport()
comm = Comm(rank, world, host_transport=transport)
comm.host = pt.TorchHostComm()
PETScOptions.clear()
multigrid_inner_options(dim=2, galerkin_u=False)
pp = pt.partitioned(Cavity, rank, world, host=comm.host, level=a.level, nu=0.01)
out = solve_steady(pp, max_newton=3, newton_rtol=0.0, comm=comm, device=DEV)
ksp = out["solver"].linear_solver().ksp()
eng = ksp.engine
res = {"its": np.array(out["krylov_per_step"]), "x": out["w"].vector().copy(),
       "ranks": eng.info(c.INFO_RANKS)}
if rank == a.fail_rank:
    sys.stderr.write("peer_ipc_worker: rank %d leaves on purpose\n" % rank)
    os._exit(3)
# one PCApply eagerly and as a replayed hipGraph (the exchanges are kernels of
# the engine's stream: they are captured with everything else)
V = pp.space
xg = np.random.default_rng(5).standard_normal(V.ndof)
y0 = eng.fieldsplit_apply_np(xg)
eng.graph_enable(True)
y1 = eng.fieldsplit_apply_np(xg)          # first: eager, counts other traffic
y2 = eng.fieldsplit_apply_np(xg)          # captured
y3 = eng.fieldsplit_apply_np(xg)          # replayed
t0 = time.perf_counter()
for _ in range(50):
    eng.fieldsplit_apply_np(xg)
t_graph = (time.perf_counter() - t0) / 50
eng.graph_enable(False)
t0 = time.perf_counter()
for _ in range(50):
    eng.fieldsplit_apply_np(xg)
t_eager = (time.perf_counter() - t0) / 50
# the same with vectors resident on the device (this rank's row block
# [u_loc; p_loc]): what bench.py times; kernel launches per PCApply
from fenapack_amd.petsc import Vec                            # noqa: E402
nl = int(eng.info(c.INFO_N_U_LOCAL)) + int(eng.info(c.INFO_N_P_LOCAL))
xd = Vec(xg[:nl].copy(), device="cuda:%d" % DEV)
yd = xd.duplicate()


def timed(n=200):
    for _ in range(10):
        eng.fieldsplit_apply(xd.t, yd.t, c.MEM_DEVICE)
    torch.cuda.synchronize()
    dist.barrier()
    l0 = eng.info(c.INFO_LAUNCHES)
    p0 = eng.info(c.INFO_PEER_CALLS)
    b0 = eng.info(c.INFO_BOOT_CALLS)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.fieldsplit_apply(xd.t, yd.t, c.MEM_DEVICE)
    torch.cuda.synchronize()
    return ((time.perf_counter() - t0) / n,
            (eng.info(c.INFO_LAUNCHES) - l0) / n,
            (eng.info(c.INFO_PEER_CALLS) - p0) / n,
            (eng.info(c.INFO_BOOT_CALLS) - b0) / n)


td_eager, launches, exchanges, boot = timed()
eng.graph_enable(True)
td_graph, launches_graph, _, _ = timed()
eng.graph_enable(False)
res.update({"exchanges_per_pcapply": exchanges, "boot_calls_per_pcapply": boot})
res.update({"y0": y0, "y1": y1, "y2": y2, "y3": y3,
            "t_graph": t_graph, "t_eager": t_eager,
            "td_eager": td_eager, "td_graph": td_graph,
            "launches_per_pcapply": launches,
            "launches_per_pcapply_graph": launches_graph,
            "peer": float(os.environ.get("PCD_COMM_PEER", "1") != "0")})
if rank == 0:
    np.savez(a.out, **res)
eng.destroy()
dist.barrier()
dist.destroy_process_group()
