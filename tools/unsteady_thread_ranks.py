#!/usr/bin/env python3
"""BASELINE config 4 (unsteady demo, 100 time steps, 2 and 4 ranks) on ONE
GPU: the ranks are threads of this process (pcd_comm_init_threads), each with
its own engine, partition and device producer - the code path of a
`torch.distributed.run --nproc-per-node R` start minus RCCL itself (which
refuses two ranks on one device).  Prints one JSON line per R: Krylov totals
(they must agree with R = 1), wall time (NOT a performance number: halos and
all-reduces of this backend are host barriers + copies).

usage: unsteady_thread_ranks.py [level=4] [steps=100] [R ...=1 2 4]"""
import ctypes
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import PETScOptions                                  # noqa
from fenapack_amd.device_producer import solve_unsteady_device         # noqa
from fenapack_amd.driver import multigrid_inner_options                # noqa
from fenapack_amd.fem import BackwardStep                              # noqa
from fenapack_amd.parallel import Comm                                 # noqa

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ranks = [int(a) for a in sys.argv[3:]] or [1, 2, 4]
dt = 0.2
# partitioned finest levels, replicated coarse ones - as on real ranks; the
# default limit (60000 rows) would replicate everything at this size
os.environ.setdefault("PCD_REPLICATE_BELOW", "2000")
PETScOptions.clear()
multigrid_inner_options(cycles_u=2, cycles_p=2)


def solve(comm):
    pb = BackwardStep(level, nu=0.02, dt=dt, dirichlet_diag="multiplicity")
    out = solve_unsteady_device(pb, dt=dt, t_end=steps * dt, newton_rtol=1e-5,
                                gmres_rtol=1e-6, comm=comm)
    return {"ndof": out["ndof"], "steps": out["steps"],
            "krylov_its": out["krylov_its"], "picard_its": out["newton_its"],
            "krylov_per_step_first10": out["krylov_per_step"][:10],
            "checksum": float(abs(out["w"].vector()).sum())}


for R in ranks:
    # every rank thread builds the whole problem: the producer's memory check
    # (fem/multigrid._check_size) must count all of them
    os.environ["FENAPACK_AMD_CONCURRENT_BUILDS"] = str(R)
    t0 = time.time()
    if R == 1:
        res = [solve(None)]
    else:
        group = ctypes.c_void_p()
        res, errs = [None] * R, []

        def body(r):
            try:
                res[r] = solve(Comm(r, R, thread_group=group))
            except Exception as ex:
                errs.append((r, repr(ex)))

        th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            print(json.dumps({"ranks": R, "errors": errs}))
            sys.exit(1)
    same = all(r["krylov_its"] == res[0]["krylov_its"]
               and abs(r["checksum"] - res[0]["checksum"])
               <= 1e-10 * res[0]["checksum"] for r in res)
    print(json.dumps(dict(res[0], ranks=R, backend="threads on one GPU"
                          if R > 1 else "one GPU", replicas_agree=same,
                          wall_seconds=round(time.time() - t0, 2))),
          flush=True)
