#!/usr/bin/env python3
"""Workload for the two rocprofv3 --pmc passes that back `roofline.traffic`:

  1. calibration: 20 x k_scale_dinv on n_u rows (exactly 24 n_u bytes, 8-byte
     lanes like the stream kernels) - gives the FETCH_SIZE correction on this
     access width (the guide: FETCH_SIZE = 1/2 of streamed bytes on gfx950);
  2. the roofline kernel: 10 x 8 fused Chebyshev steps on the finest A00;
  3. K = 10 whole fieldsplit PCApplies of the benchmark configuration (eager
     launches), each one bracketed by k_gather ... k_scatter.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT_F -- \
        python3 $REPO/tools/pmc_pcapply.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT_W -- \
        python3 $REPO/tools/pmc_pcapply.py
    python3 $REPO/tools/pmc_roofline.py OUT_F OUT_W > profiles/rNN_pmc_roofline.json

Same options as bench.py for the workload (--level, --geometry, --variant)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fenapack_amd import PETScOptions, _cabi as c                       # noqa
from fenapack_amd.driver import make_solver, multigrid_inner_options    # noqa
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D            # noqa

p = argparse.ArgumentParser()
p.add_argument("--level", type=int, default=6)
p.add_argument("--geometry", default="cavity")
p.add_argument("--variant", default="BRM1")
p.add_argument("--n0", type=int, default=4)
p.add_argument("--algebraic", action="store_true")
p.add_argument("--inner", default="mg", choices=["mg", "jacobi"],
               help="jacobi: the north star's literal solvers (bench.py "
                    "--inner jacobi) + stage 4, fixed-count Jacobi-PCG on Ap")
p.add_argument("--a00-its", type=int, default=240)
p.add_argument("--a00-ratio", type=float, default=0.002)
p.add_argument("--cycles-p", type=int, default=1)
p.add_argument("--supg", action="store_true")
a = p.parse_args()
if a.geometry == "cavity":
    pb = Cavity(a.level, nu=0.01, variant=a.variant)
elif a.geometry == "cube":
    # (config 5's own mesh is above the host assembler's default limit)
    os.environ.setdefault("FENAPACK_AMD_MAX_CELLS", "4000000")
    pb = Cavity3D(a.level, nu=0.01, n0=a.n0, variant=a.variant)
else:
    pb = BackwardStep(a.level, nu=0.02, variant=a.variant)
V = pb.space
PETScOptions.clear()
if a.inner == "mg":
    multigrid_inner_options(dim=V.dim, algebraic=a.algebraic,
                            cycles_p=a.cycles_p)
else:
    from fenapack_amd.driver import default_inner_options              # noqa
    default_inner_options(a00_its=a.a00_its, a00_ratio=a.a00_ratio, dim=V.dim)
w, nls, nlp = make_solver(pb, gmres_rtol=1e-6, restart=150, newton_rtol=1e-5,
                          max_newton=2)
nls.parameters["error_on_nonconvergence"] = False
nls.solve(nlp, w.vector(), on_update=w.touch)
ksp = nls.linear_solver().ksp()
eng = ksp.engine
rng = np.random.default_rng(0)
bu, xu = rng.standard_normal(V.n_u), np.empty(V.n_u)
from fenapack_amd.petsc import estimate_emax                           # noqa
ksp0 = ksp.pc.getFieldSplitSubKSP()[0]
emax = 1.1 * estimate_emax(ksp0.getOperators()[1].A, iters=12)
eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 0, 0.0, 0.1 * emax, emax)
for _ in range(20):                      # (1) k_scale_dinv only
    eng.inner_solve(c.KSP_A00, bu, xu)
# an even step count lands on k_cheb_first + 8 k_cheb_step launches
eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 9, 0.0, 0.1 * emax, emax)
for _ in range(10):                      # (2) the roofline kernel
    eng.inner_solve(c.KSP_A00, bu, xu)
if a.inner == "mg":
    eng.set_inner(c.KSP_A00, "richardson", "mg", 1, 0.0)
else:
    ksp0.push_settings()
    # (4) the Jacobi-PCG iteration on Ap: 10 solves of 40 iterations, rtol 0
    pcd = ksp.pc.getFieldSplitSubKSP()[1].pc.getPythonContext()
    bp, xp = rng.standard_normal(V.n_p), np.empty(V.n_p)
    eng.set_inner(c.KSP_AP, "cg", "jacobi", 40, 0.0)
    for _ in range(10):
        eng.inner_solve(c.KSP_AP, bp, xp)
    pcd.ksp_Ap.push_settings()
x, y = rng.standard_normal(V.ndof), np.empty(V.ndof)
eng.graph_enable(False)
for _ in range(10 if a.inner == "mg" else 3):     # (3) whole PCApplies
    eng.fieldsplit_apply(x, y)
print("n_u", V.n_u, "n_p", V.n_p, "nnz_A00", int(eng.info(c.INFO_NNZ_BASE + c.MAT_A00)),
      "components", int(eng.info(c.INFO_A00_COMPONENTS)),
      "rows_per_wg", int(eng.info(c.INFO_A00_ROWS_PER_WG)))
