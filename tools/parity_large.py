#!/usr/bin/env python3
"""One-off parity record at a size too large for the test-suite (host set-up
of several minutes): the benchmark state of `bench.py --geometry G --level L
--n0 N` on the HIP engine, the same operators mirrored into the C oracle, one
fieldsplit PCApply of each on a seeded vector.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

# (config 5's own mesh: 2.33 M tetrahedra, above the host assembler's default
# limit; the resident-set watchdog of the repository's scripts stays on)
os.environ.setdefault("FENAPACK_AMD_MAX_CELLS", "4000000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle                                                            # noqa
from fenapack_amd import PETScOptions                                    # noqa
from fenapack_amd import _cabi as c                                      # noqa
from fenapack_amd.driver import make_solver, multigrid_inner_options     # noqa
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D             # noqa

p = argparse.ArgumentParser()
p.add_argument("--geometry", default="cube")
p.add_argument("--level", type=int, default=4)
p.add_argument("--n0", type=int, default=5)
p.add_argument("--algebraic", action="store_true",
               help="-pc_type gamg: hierarchy from the matrix (meshes without "
                    "a nested one, e.g. --level 0 --n0 73)")
a = p.parse_args()
t0 = time.time()
pb = {"cube": lambda: Cavity3D(a.level, nu=0.01, n0=a.n0),
      "cavity": lambda: Cavity(a.level, nu=0.01),
      "lshape": lambda: BackwardStep(a.level, nu=0.02)}[a.geometry]()
V = pb.space
PETScOptions.clear()
multigrid_inner_options(dim=V.dim, algebraic=a.algebraic)
w, nls, nlp = make_solver(pb, gmres_rtol=1e-6, restart=150, newton_rtol=0.0,
                          max_newton=2)
nls.parameters["absolute_tolerance"] = 0.0
nls.parameters["error_on_nonconvergence"] = False
nls.solve(nlp, w.vector(), on_update=w.touch)
ksp = nls.linear_solver().ksp()
t_setup = time.time() - t0
t0 = time.time()
o = oracle.mirror(oracle.Engine(pb.variant), pb, ksp)
t_mirror = time.time() - t0
x = np.random.default_rng(0).standard_normal(V.ndof)
yg = ksp.engine.fieldsplit_apply_np(x)
t0 = time.time()
yo = o.fieldsplit_apply_np(x)
t_oracle = time.time() - t0
err = float(np.abs(yg - yo).max() / np.abs(yo).max())
err_p = float(np.abs(yg[V.is_p] - yo[V.is_p]).max() / np.abs(yo[V.is_p]).max())
# the pressure block alone: PCDPC_BRM1.apply (preconditioners.py:98-135)
xp = np.random.default_rng(1).standard_normal(V.n_p)
zg, zo = ksp.engine.apply_np(xp), o.apply_np(xp)
err_pcd = float(np.abs(zg - zo).max() / np.abs(zo).max())
from fenapack_amd import _guard                                          # noqa
print(json.dumps({
    "workload": "%s level %d n0 %d%s" % (a.geometry, a.level, a.n0,
                                        " gamg" if a.algebraic else ""),
    "ndof": int(V.ndof), "gmres_its_per_step": list(nls.krylov_history),
    "hip_vs_oracle_rel_err": err, "pressure_block_rel_err": err_p,
    "pcd_apply_rel_err": err_pcd,
    "a00_components": int(ksp.engine.info(c.INFO_A00_COMPONENTS)),
    "host_peak_rss_gb": round(_guard.peak_rss_bytes() / 1e9, 2),
    "host_rss_watchdog_limit_gb": None if not _guard._WATCHDOG["limit"]
    else round(_guard._WATCHDOG["limit"] / 1e9, 1),
    "seconds": {"setup": t_setup, "mirror": t_mirror,
                "one_oracle_pcapply": t_oracle}}))
