#!/usr/bin/env python3
"""Worker of tests/test_two_gpus.py: one process per GPU over real RCCL.

Every rank builds the same small problem, attaches the engine to the RCCL
communicator (unique id carried by torch.distributed), and runs the
row-partitioned fieldsplit PCApply and the full GMRES solve with HOST vectors
(global in, global out); rank 0 writes the results.  ``--fail-rank R``: that
rank exits with code 3 after set-up, while the others are inside a collective
- the launcher has to end them (no hang)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

p = argparse.ArgumentParser()
p.add_argument("--out", required=True)
p.add_argument("--fail-rank", type=int, default=-1)
p.add_argument("--handover", choices=("global", "rows"), default="global",
               help="rows: every operator and the block system go over as "
                    "this rank's rows only (pcd_set_csr_local, "
                    "pcd_set_system_local)")
a = p.parse_args()

import torch                                                  # noqa: E402
import torch.distributed as dist                              # noqa: E402

rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=torch.device("cuda", local))

from fenapack_amd import _cabi as c                           # noqa: E402
from fenapack_amd.parallel import Comm                        # noqa: E402
from helpers import flow_state, configure_engine, set_iter_cfg  # noqa: E402

comm = Comm.world()
st = flow_state("lshape", 3, dt=0.2)
V = st["V"]
e = c.Engine(c.hip_library(), "RBRM1", local)
if comm.unique_id() is None:
    raise SystemExit("two_rank_worker: no communicator (one rank needs "
                     "PCD_FORCE_COMM=1)")
e.comm_init(comm.rank, comm.size, comm.unique_id())
if a.handover == "rows":
    import scipy.sparse as sp                                  # noqa: E402
    pb = st["pb"]
    e.set_velocity_block(V.dim)
    p0, p1 = e.row_range(V.n_p)
    u0, u1 = e.row_range(V.n_u, velocity=True)
    for which, M in ((c.MAT_AP, pb.Ap), (c.MAT_MP, pb.Mp),
                     (c.MAT_KP, st["Kp"]), (c.MAT_RP, st["Rp"])):
        e.set_csr_local(which, sp.csr_matrix(M)[p0:p1], M.shape)
    e.set_bc(pb.bc_p_idx, pb.bc_p_val)
    rows = np.concatenate([V.is_u[u0:u1], V.is_p[p0:p1]])
    e.set_system_local(sp.csr_matrix(st["A"])[rows], rows, V.ndof, V.is_u,
                       V.is_p)
else:
    configure_engine(e, st)
set_iter_cfg(e)
e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
e.setup()
if rank == a.fail_rank:
    sys.stderr.write("two_rank_worker: rank %d fails on purpose\n" % rank)
    os._exit(3)
rng = np.random.default_rng(20)
xp, xs = rng.standard_normal(V.n_p), rng.standard_normal(V.ndof)
res = {"ranks": e.info(c.INFO_RANKS),
       "nu_loc": e.info(c.INFO_N_U_LOCAL),
       "Kp": e.spmv_np(c.MAT_KP, xp, V.n_p),
       "A": e.spmv_np(c.MAT_A, xs, V.ndof),
       "pcd": e.apply_np(xp),
       "fs": e.fieldsplit_apply_np(xs)}
e.set_inner(c.KSP_AP, "cg", "jacobi", 3000, 1e-10)
e.set_inner(c.KSP_MP, "cg", "jacobi", 3000, 1e-10)
e.set_inner(c.KSP_RP, "cg", "jacobi", 3000, 1e-10)
e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 40, 0.0, 0.02, 2.2)
# (a random right-hand side: the frozen state of the unsteady problem at t = 0
# has a zero residual)
x, its, rnorm = e.gmres_np(rng.standard_normal(V.ndof), rtol=1e-6,
                           restart=150, max_it=600)
res.update({"gmres_x": x, "gmres_its": its})
if rank == 0:
    np.savez(a.out, **res)
e.destroy()
dist.destroy_process_group()
