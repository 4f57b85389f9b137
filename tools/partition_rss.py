#!/usr/bin/env python3
"""Peak host memory of ONE rank's partitioned build (fem/partition.py): the
skeleton of the whole problem + the operators of the rank's rows.  Host code
only - runs in the build container without a GPU.

  tools/partition_rss.py --geometry cube --n0 73 --level 0 --ranks 8 --rank 3
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import _guard                                      # noqa: E402
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D          # noqa: E402
from fenapack_amd.fem import partition as pt                         # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--geometry", default="cube", choices=["cube", "cavity", "lshape"])
p.add_argument("--n0", type=int, default=73)
p.add_argument("--level", type=int, default=0)
p.add_argument("--ranks", type=int, default=8)
p.add_argument("--rank", type=int, default=3)
args = p.parse_args()
os.environ.setdefault("FENAPACK_AMD_MAX_CELLS", "4000000")
# (the producer's whole-problem estimate does not apply to a skeleton)
os.environ.setdefault("FENAPACK_AMD_IGNORE_MEMORY", "1")
t0 = time.time()
stages = []


def rep(tag):
    stages.append({"stage": tag, "rss_gb": round(_guard.rss_bytes() / 1e9, 2),
                   "peak_gb": round(_guard.peak_rss_bytes() / 1e9, 2),
                   "t": round(time.time() - t0, 1)})
    sys.stderr.write("%-28s rss %.2f GB peak %.2f GB t=%.1f\n" % (
        tag, stages[-1]["rss_gb"], stages[-1]["peak_gb"], stages[-1]["t"]))


if args.geometry == "cube":
    cls, kw = Cavity3D, dict(level=args.level, nu=0.01, n0=args.n0)
elif args.geometry == "cavity":
    cls, kw = Cavity, dict(level=args.level, nu=0.01)
else:
    cls, kw = BackwardStep, dict(level=args.level, nu=0.02)
pp = pt.partitioned(cls, args.rank, args.ranks, **kw)
V = pp.space
rep("skeleton + local operators")
xu, xp = pp.initial_guess()
L = pp.linearise(xu, xp)
rep("linearise")
Kp = pp.Kp(xu)
rep("Kp")
from fenapack_amd.fem.forms import navier_stokes_forms               # noqa: E402
w, forms = navier_stokes_forms(pp)
A = forms["a"].assemble()
rep("monolithic rows")
print(json.dumps({
    "workload": "%s n0 %d level %d" % (args.geometry, args.n0, args.level),
    "ndof": int(V.ndof), "cells": int(V.mesh.num_cells),
    "ranks": args.ranks, "rank": args.rank,
    "local_cells": int(pp.fine.sub.cells.size),
    "owned_rows_u": int(pp.fine.own_u[1] - pp.fine.own_u[0]),
    "owned_rows_p": int(pp.fine.own_p[1] - pp.fine.own_p[0]),
    "nnz_local_rows_of_A": int(A.nnz),
    "peak_rss_gb": round(_guard.peak_rss_bytes() / 1e9, 2),
    "stages": stages}))
