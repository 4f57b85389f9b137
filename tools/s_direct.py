#!/usr/bin/env python3
"""S-direct operating point on the CPU: outer GMRES iterations per
nonlinear step with EXACT inner solves (scipy splu in the role of the
reference's default LU / Cholesky, field_split.py:94-98,
preconditioners.py:42-49), for the steady cavity / backward-facing step and
for the unsteady demo whose totals the reference publishes
(documentation.rst:134-140: 3157 PCD / 1686 PCDR over 25 steps at level 4).
One JSON line.  Also what produced profiles/r02_pcd*_cpu.log's exact-solve
companions.

  python tools/s_direct.py steady cavity 5
  python tools/s_direct.py unsteady lshape 4 [--pcdr] [--dirichlet-diag unit]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D      # noqa: E402
from oracle import s_direct                                       # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("mode", choices=["steady", "unsteady"])
p.add_argument("geometry", choices=["cavity", "lshape", "cube"])
p.add_argument("level", type=int)
p.add_argument("--variant", default="BRM1")
p.add_argument("--pcdr", action="store_true")
p.add_argument("--n0", type=int, default=4)
p.add_argument("--dt", type=float, default=0.2)
p.add_argument("--t-end", type=float, default=5.0)
p.add_argument("--max-steps", type=int, default=25)
p.add_argument("--rtol", type=float, default=1e-5)
p.add_argument("--dirichlet-diag", default="multiplicity",
               choices=["unit", "multiplicity"])
a = p.parse_args()
kw = dict(variant=a.variant, pcdr=a.pcdr, dirichlet_diag=a.dirichlet_diag)
if a.mode == "unsteady":
    kw["dt"] = a.dt
if a.geometry == "cavity":
    pb = Cavity(a.level, nu=0.01, **kw)
elif a.geometry == "cube":
    pb = Cavity3D(a.level, nu=0.01, n0=a.n0, **kw)
else:
    pb = BackwardStep(a.level, nu=0.02, **kw)
if a.mode == "steady":
    r = s_direct.steady(pb, rtol=a.rtol, max_it=a.max_steps)
    r.pop("xu"), r.pop("xp")
else:
    r = s_direct.unsteady(pb, a.dt, a.t_end, rtol=a.rtol)
r.update({"case": "%s %s level %d %s%s" % (a.mode, a.geometry, a.level,
                                            "PCDR " if a.pcdr else "PCD ",
                                            a.variant),
          "ndof": int(pb.space.ndof),
          "inner": "exact (scipy splu) for A00, Ap, Mp%s"
                   % (", Rp" if a.pcdr else ""),
          "dirichlet_diag": a.dirichlet_diag})
print(json.dumps(r))
