#!/usr/bin/env python3
"""End-to-end steady nonlinear solve: host producer vs device producer
(time per nonlinear step, GMRES counts).
usage: device_producer_timing.py [cavity|cube] [level] [newton_rtol]
                                 [galerkin|rediscretised] [picard|newton]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import PETScOptions                                   # noqa
from fenapack_amd.device_producer import solve_steady_device            # noqa
from fenapack_amd.driver import multigrid_inner_options, solve_steady   # noqa
from fenapack_amd.fem import Cavity, Cavity3D                           # noqa

geometry = sys.argv[1] if len(sys.argv) > 1 else "cavity"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 6
newton_rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-8
galerkin = (sys.argv[4] if len(sys.argv) > 4 else "galerkin") == "galerkin"
nls = sys.argv[5] if len(sys.argv) > 5 else "picard"


def problem():
    return Cavity(level, nu=0.01, nls=nls) if geometry == "cavity" \
        else Cavity3D(level, nu=0.01, n0=4, nls=nls)


res = {}
for name, fn in (("host", solve_steady), ("device", solve_steady_device)):
    pb = problem()
    PETScOptions.clear()
    multigrid_inner_options(dim=pb.space.dim, galerkin_u=galerkin)
    t0 = time.time()
    out = fn(pb, max_newton=25, newton_rtol=newton_rtol)
    res[name] = {
        "ndof": pb.space.ndof, "nls": nls, "coarse_operators":
        "galerkin" if galerkin else "rediscretised",
        "newton_its": out["newton_its"],
        "converged": bool(out["converged"]),
        "krylov_per_step": out["krylov_per_step"],
        "solve_seconds": out["time"],
        "final_residual": out["residuals"][-1]}
    if name == "device":
        steps = out["newton_its"] - 1
        pt = out["producer_timing"]
        res[name].update(
            plan_seconds=out["time_plan"],
            device_steps=steps,
            seconds_per_device_step=(out["time_gmres"] + sum(pt.values())
                                     + out["time_device_loop"])
            / max(steps, 1),
            device_loop_seconds=out["time_device_loop"],
            gmres_seconds=out["time_gmres"],
            producer_timing=pt)
    else:
        res[name]["seconds_per_step"] = out["time"] / max(out["newton_its"], 1)
print(json.dumps(res, indent=1))
