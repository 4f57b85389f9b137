#!/usr/bin/env python3
"""The PARTITIONED path against the oracle at a size where one host holds the
whole problem next to the slabs: cube N on R thread ranks of the one GPU -
partitioned producer (fem/partition.py), rank-local hand-over, the algebraic
hierarchy aggregated rank by rank (amg.PartitionedSA).  The hierarchy depends
on the rank count, so the checker is handed THIS hierarchy: every rank's rows
of every operator and prolongator are put back together on rank 0
(oracle.mirror_partitioned) and one fieldsplit PCApply + one PCD apply of the
global vector through all ranks are compared with the one-thread oracle.
A process of its own (like tools/parity_large.py): the eight slabs, the whole
operators and the oracle's copies are more than the suite's session watchdog
allows.  Prints one JSON line.

    python3 tools/parity_partitioned.py --n0 48 --ranks 8
"""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

os.environ.setdefault("FENAPACK_AMD_MAX_CELLS", "4000000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                            # noqa
from fenapack_amd import PETScOptions, _guard                            # noqa
from fenapack_amd import _cabi as c                                      # noqa
from fenapack_amd.driver import multigrid_inner_options, solve_steady    # noqa
from fenapack_amd.fem import Cavity3D                                    # noqa
from fenapack_amd.fem import partition as pt                             # noqa
from fenapack_amd.parallel import Comm                                   # noqa

p = argparse.ArgumentParser()
p.add_argument("--n0", type=int, default=48)
p.add_argument("--ranks", type=int, default=8)
a = p.parse_args()
R = a.ranks
t0 = time.time()
PETScOptions.clear()
multigrid_inner_options(dim=3, algebraic=True)
kw = dict(level=0, nu=0.01, n0=a.n0)
hosts = pt.ThreadHostComm.group(R)
group = ctypes.c_void_p()
res, errs = [None] * R, []


def relerr(u, v):
    return float(np.abs(u - v).max() / np.abs(v).max())


def body(r):
    try:
        comm = Comm(r, R, thread_group=group)
        comm.host = hosts[r]
        pp = pt.partitioned(Cavity3D, r, R, host=hosts[r], **kw)
        out = solve_steady(pp, max_newton=2, newton_rtol=0.0, comm=comm)
        ksp = out["solver"].linear_solver().ksp()
        eng, V = ksp.engine, pp.space
        rng = np.random.default_rng(0)
        x = rng.standard_normal(V.ndof)
        xp = rng.standard_normal(V.n_p)
        yg = eng.fieldsplit_apply_np(x)              # collective
        zg = eng.apply_np(xp)
        o = oracle.mirror_partitioned(
            oracle.Engine(pp.variant) if r == 0 else None, pp, ksp)
        psa = ksp.pc.getFieldSplitSubKSP()[0].pc._mg_psa
        rec = {"its": list(out["krylov_per_step"]), "ndof": int(V.ndof),
               "partitioned_levels": [bool(f) for f in
                                      psa.partitioned_levels()],
               "rows_u": int(eng.info(c.INFO_N_U_LOCAL))}
        if r == 0:
            yo, zo = o.fieldsplit_apply_np(x), o.apply_np(xp)
            rec["hip_ranks_vs_oracle_rel_err"] = relerr(yg, yo)
            rec["pressure_block_rel_err"] = relerr(yg[V.is_p], yo[V.is_p])
            rec["pcd_apply_rel_err"] = relerr(zg, zo)
        res[r] = rec
    except Exception as ex:            # pragma: no cover
        import traceback
        errs.append((r, repr(ex), traceback.format_exc()))
        try:
            hosts[r]._sh.barrier.abort()
        except Exception:
            pass


th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
for t in th:
    t.start()
for t in th:
    t.join(timeout=1500)
if any(t.is_alive() for t in th):
    print(json.dumps({"error": "ranks deadlocked"}))
    os._exit(2)
if errs:
    sys.stderr.write("\n".join(e[2] for e in errs))
    print(json.dumps({"error": [e[:2] for e in errs]}))
    sys.exit(1)
rec = dict(res[0])
rec.update({"workload": "cube N=%d, partitioned producer, gamg, %d thread "
                        "ranks vs oracle.mirror_partitioned" % (a.n0, R),
            "ranks": R, "rows_u_per_rank": [q["rows_u"] for q in res],
            "replicas_agree_on_history": all(q["its"] == res[0]["its"]
                                             for q in res),
            "host_peak_rss_gb": round(_guard.peak_rss_bytes() / 1e9, 2),
            "seconds": round(time.time() - t0, 1)})
print(json.dumps(rec))
