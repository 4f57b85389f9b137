#!/usr/bin/env python3
"""What a Picard step costs AROUND the solve when the velocity hierarchy is
algebraic (-pc_type gamg): the host assembly of the operators, and the
refresh of the hierarchy for the new iterate (Galerkin products on the
aggregation kept from the first step, smoother bounds, coarse inverse, value
hand-over) - first step (aggregation + prolongators built) against the
following ones.

    python tools/gamg_refresh_cost.py cube 0 73      # config 5's own mesh
    python tools/gamg_refresh_cost.py cube 3 4       # N = 32

One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import fenapack_amd                                            # noqa: E402,F401
from fenapack_amd import PETScOptions, petsc                    # noqa: E402
from fenapack_amd._guard import peak_rss_bytes                  # noqa: E402
from fenapack_amd.driver import multigrid_inner_options, solve_steady  # noqa: E402
from fenapack_amd.fem import Cavity, Cavity3D                   # noqa: E402


def main():
    kind, level = sys.argv[1], int(sys.argv[2])
    n0 = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    pb = Cavity3D(level, nu=0.01, n0=n0) if kind == "cube" else Cavity(level, nu=0.01)
    PETScOptions.clear()
    multigrid_inner_options(dim=pb.space.dim, algebraic=True)
    rec = {"assemble": [], "mg_push": {}, "step": []}
    lin0 = pb.linearise

    def lin(*a, **k):
        t = time.perf_counter()
        out = lin0(*a, **k)
        rec["assemble"].append(time.perf_counter() - t)
        return out
    pb.linearise = lin
    push0 = petsc.KSP._push_multigrid

    def push(self):
        t = time.perf_counter()
        out = push0(self)
        rec["mg_push"].setdefault(self.getOptionsPrefix() or "?", []).append(
            time.perf_counter() - t)
        return out
    petsc.KSP._push_multigrid = push
    t0 = time.perf_counter()
    out = solve_steady(pb, max_newton=steps, newton_rtol=0.0)
    wall = time.perf_counter() - t0
    print(json.dumps({
        "tool": "gamg_refresh_cost", "geometry": kind, "level": level, "n0": n0,
        "ndof": int(pb.space.ndof), "picard_steps": steps,
        "krylov_per_step": out["krylov_per_step"],
        "host_assembly_seconds_per_call": [round(x, 3) for x in rec["assemble"]],
        "hierarchy_push_seconds_per_call": {k: [round(x, 3) for x in v]
                                            for k, v in rec["mg_push"].items()},
        "wall_seconds": round(wall, 2),
        "peak_rss_gb": round(peak_rss_bytes() / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    main()
