#!/usr/bin/env python3
"""BASELINE config 5's shape (3-D lid-driven cavity, three velocity components,
row-partitioned over 8 ranks) on ONE GPU: the ranks are threads of this
process (pcd_comm_init_threads), each with its own engine, partition, halo
plans and device producer.  Steady Picard solve; prints one JSON line per R.
Parity of the partitioned code path (Krylov counts, solution), NOT a timing.

usage: steady_thread_ranks.py [--host] [--partitioned] [--algebraic] [--rediscretise] [--n0=N] [cube|cavity] [level] [R ...=1 8]

--partitioned: the partitioned operator producer (fem/partition.py): every
rank thread assembles its rows only (implies --host); the process's peak
resident set / R is then one rank's footprint (reported).

--host: the nonlinear steps driven from the host, as ``bench.py`` sets its
workload up (two steps, exactly), instead of the device producer; with
FENAPACK_AMD_LOCAL_HANDOVER=1 every rank hands over its own rows only - the
shape of the driver's ``bench.py --gpus 8`` start-up minus RCCL."""
import ctypes
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import PETScOptions, _guard                          # noqa
from fenapack_amd.device_producer import solve_steady_device           # noqa
from fenapack_amd.driver import multigrid_inner_options                # noqa
from fenapack_amd.fem import Cavity, Cavity3D                          # noqa
from fenapack_amd.parallel import Comm                                 # noqa

PARTITIONED = "--partitioned" in sys.argv
HOST = "--host" in sys.argv or PARTITIONED
REDISC = "--rediscretise" in sys.argv        # -pc_mg_galerkin none for A00
ALGEBRAIC = "--algebraic" in sys.argv        # -pc_type gamg: no nested hierarchy needed
N0 = ([int(a[5:]) for a in sys.argv if a.startswith("--n0=")] or [4])[0]
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
geometry = argv[0] if len(argv) > 0 else "cube"
level = int(argv[1]) if len(argv) > 1 else 2
ranks = [int(a) for a in argv[2:]] or [1, 8]
dim = 3 if geometry == "cube" else 2
# the two finest levels partitioned, the rest replicated (default limit:
# 60000 rows; lowered so that small runs exercise both kinds of level)
os.environ.setdefault("PCD_REPLICATE_BELOW", "20000")
PETScOptions.clear()
multigrid_inner_options(dim=dim, algebraic=ALGEBRAIC,
                        galerkin_u=not (REDISC or (PARTITIONED
                                                   and not ALGEBRAIC)))
if PARTITIONED:
    os.environ.setdefault("FENAPACK_AMD_MAX_CELLS", "4000000")
    # (the whole-problem estimate does not apply: a rank builds its slab)
    os.environ.setdefault("FENAPACK_AMD_IGNORE_MEMORY", "1")


def solve(comm):
    if PARTITIONED and comm is not None:
        from fenapack_amd.fem import partition as pt
        cls, kw = (Cavity3D, dict(level=level, nu=0.01, n0=N0)) \
            if geometry == "cube" else (Cavity, dict(level=level, nu=0.01))
        pb = pt.partitioned(cls, comm.rank, comm.size, host=comm.host, **kw)
    else:
        pb = Cavity3D(level, nu=0.01, n0=N0) if geometry == "cube" \
            else Cavity(level, nu=0.01)
    if HOST:
        from fenapack_amd.driver import solve_steady
        out = solve_steady(pb, gmres_rtol=1e-6, restart=150, newton_rtol=0.0,
                           max_newton=2, comm=comm)
    else:
        out = solve_steady_device(pb, max_newton=10, comm=comm)
    eng = out["solver"].linear_solver().ksp().engine
    from fenapack_amd import _cabi as c
    return {"ndof": pb.space.ndof, "picard_its": out["newton_its"],
            "converged": bool(out["converged"]),
            "krylov_per_step": out["krylov_per_step"],
            "final_residual": out["residuals"][-1],
            "rows_u_of_this_rank": int(eng.info(c.INFO_N_U_LOCAL)),
            "local_handover": bool(eng.local_handover),
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
        from fenapack_amd.fem.partition import ThreadHostComm
        hosts = ThreadHostComm.group(R)

        def body(r):
            try:
                comm = Comm(r, R, thread_group=group)
                comm.host = hosts[r]
                res[r] = solve(comm)
            except Exception as ex:
                import traceback
                errs.append((r, repr(ex), traceback.format_exc()))
                hosts[r]._sh.barrier.abort()

        th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            print(json.dumps({"ranks": R, "errors": errs}))
            sys.exit(1)
    same = all(r["krylov_per_step"] == res[0]["krylov_per_step"]
               and abs(r["checksum"] - res[0]["checksum"])
               <= 1e-10 * res[0]["checksum"] for r in res)
    line = dict(res[0], ranks=R, geometry=geometry, level=level,
                rows_u_per_rank=[r["rows_u_of_this_rank"] for r in res],
                replicas_agree=same, wall_seconds=round(time.time() - t0, 2),
                producer="partitioned" if PARTITIONED and R > 1 else "global",
                process_peak_rss_gb=round(_guard.peak_rss_bytes() / 1e9, 2),
                peak_rss_gb_per_rank=round(_guard.peak_rss_bytes() / 1e9 / R,
                                           2),
                backend="threads on one GPU" if R > 1 else "one GPU")
    line.pop("rows_u_of_this_rank")
    print(json.dumps(line), flush=True)
