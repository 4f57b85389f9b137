#!/usr/bin/env python3
"""Mesh-scaling sweep in the spirit of the reference's
``test/bench/test_pcd_scaling.py`` (levels x nls x PCD variant): prints ndofs,
outer GMRES iterations per nonlinear step and solve time as JSON lines."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd import PETScOptions                             # noqa: E402
from fenapack_amd.driver import multigrid_inner_options, solve_steady  # noqa
from fenapack_amd.fem import BackwardStep, Cavity                  # noqa: E402

geometry = sys.argv[1] if len(sys.argv) > 1 else "lshape"
levels = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,3,4,5").split(",")]
producer = sys.argv[3] if len(sys.argv) > 3 else "host"   # host | device
if producer == "device":
    # operators, residual and the Picard loop on the device (Picard only)
    from fenapack_amd.device_producer import solve_steady_device as solve_steady  # noqa
for level in levels:
    for nls in (("picard",) if producer == "device" else ("picard", "newton")):
        for variant in ("BRM1", "BRM2"):
            PETScOptions.clear()
            multigrid_inner_options(cycles_u=2, cycles_p=2)
            pb = (BackwardStep(level, nu=0.02, variant=variant, nls=nls)
                  if geometry == "lshape"
                  else Cavity(level, nu=0.01, variant=variant, nls=nls))
            t0 = time.time()
            out = solve_steady(pb, newton_rtol=1e-5, gmres_rtol=1e-6)
            extra = {} if producer == "host" else {
                "t_device_loop": round(out["time_device_loop"], 3),
                "t_plans": round(out["time_plan"], 2)}
            print(json.dumps({"geometry": geometry, "level": level,
                              "producer": producer, **extra,
                              "ndof": pb.space.ndof, "nls": nls,
                              "pcd": variant, "converged": out["converged"],
                              "newton_its": out["newton_its"],
                              "krylov_per_step": out["krylov_per_step"],
                              "t_solve": round(time.time() - t0, 2)}),
                  flush=True)
