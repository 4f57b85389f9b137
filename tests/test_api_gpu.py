"""GPU suite, part 2: the API mirror driving the HIP engine end to end."""
import numpy as np
import pytest

import oracle
from fenapack_amd import (PCDAssembler, PCDKrylovSolver, PETScOptions,
                          PCDPC_BRM1, PCDPC_BRM2, PCDRPC_BRM1, PCDRPC_BRM2)
from fenapack_amd import _cabi as c
from fenapack_amd.driver import (default_inner_options, make_solver,
                                 solve_steady, solve_unsteady)
from fenapack_amd.fem import BackwardStep, Cavity
from fenapack_amd.fem.forms import navier_stokes_forms
from fenapack_amd.petsc import Mat, Vec
from helpers import relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _clean_options():
    PETScOptions.clear()
    yield
    PETScOptions.clear()


def _initialised_solver(pb, prefix="", cls=None):
    w, forms = navier_stokes_forms(pb)
    a = PCDAssembler(**forms)
    s = PCDKrylovSolver()
    if prefix:
        s.set_options_prefix(prefix)
    A = Mat()
    a.system_matrix(A)
    s.set_operators(A, A)
    s.init_pcd(a, cls)
    return s, a, w


def test_set_options_prefix_early_works_late_raises():
    # test/unit/test_fieldsplit.py:84-96
    s, a, w = _initialised_solver(BackwardStep(1, nu=0.1), prefix="foo_")
    k0, k1 = s.ksp().pc.getFieldSplitSubKSP()
    assert k0.getOptionsPrefix() == "foo_fieldsplit_u_"
    assert k1.getOptionsPrefix() == "foo_fieldsplit_p_"
    pcd = k1.pc.getPythonContext()
    assert pcd.ksp_Ap.getOptionsPrefix() == "foo_fieldsplit_p_PCD_Ap_"
    assert pcd.mat_Kp.getOptionsPrefix() == "foo_fieldsplit_p_PCD_Kp_"
    with pytest.raises(RuntimeError):
        s.set_options_prefix("bar_")
    with pytest.raises(RuntimeError):
        s.init_pcd(a)                          # only one call allowed
    with pytest.raises(RuntimeError):
        pcd.init_pcd(None)                     # PCDPC re-initialisation


@pytest.mark.parametrize("cls,pcdr", [(PCDPC_BRM1, False), (PCDPC_BRM2, False),
                                      (PCDRPC_BRM1, True),
                                      (PCDRPC_BRM2, True)])
def test_fused_apply_equals_apply_by_parts(cls, pcdr):
    variant = "BRM1" if cls.__name__.endswith("1") else "BRM2"
    pb = BackwardStep(3, nu=0.1, variant=variant, dt=0.2, pcdr=pcdr)
    pb.t = 1.0
    PETScOptions.set("fieldsplit_p_PCD_Ap_ksp_max_it", 9)
    PETScOptions.set("fieldsplit_p_PCD_Ap_ksp_norm_type", "none")
    PETScOptions.set("fieldsplit_p_PCD_Mp_ksp_type", "chebyshev")
    PETScOptions.set("fieldsplit_p_PCD_Mp_ksp_max_it", 5)
    PETScOptions.set("fieldsplit_p_PCD_Mp_ksp_chebyshev_eigenvalues", "0.5,2")
    PETScOptions.set("fieldsplit_p_PCD_Rp_ksp_max_it", 7)
    PETScOptions.set("fieldsplit_p_PCD_Rp_ksp_norm_type", "none")
    s, a, w = _initialised_solver(pb, cls=cls)
    pc1 = s.ksp().pc.getFieldSplitSubKSP()[1].pc
    ctx = pc1.getPythonContext()
    assert isinstance(ctx, cls)
    x = Vec(np.random.default_rng(0).standard_normal(pb.space.n_p))
    x0 = x.getArray().copy()
    y, y2 = x.duplicate(), x.duplicate()
    ctx.apply(pc1, x, y)
    ctx.apply_by_parts(pc1, x, y2)
    assert np.array_equal(x.getArray(), x0)             # x is borrowed
    assert relerr(y.getArray(), y2.getArray()) < 1e-13
    with pytest.raises(ValueError):
        ctx.get_work_vecs(x, 3)                        # count is frozen


def _run_on(lib, monkeypatch, make_problem, **opts):
    """Drive the same Python stack on the HIP engine or (test only) on the
    oracle library injected in its place."""
    if lib is not None:
        monkeypatch.setattr(c, "hip_library", lambda: lib)
    PETScOptions.clear()
    default_inner_options(**opts)
    return solve_steady(make_problem(), newton_rtol=1e-5, gmres_rtol=1e-6)


def test_steady_solve_same_krylov_counts_as_cpu_restatement(monkeypatch):
    mk = lambda: BackwardStep(2, nu=0.1)
    opts = dict(a00_its=30, a00_ratio=0.03, ap_rtol=1e-12)
    gpu = _run_on(None, monkeypatch, mk, **opts)
    cpu = _run_on(oracle.library(), monkeypatch, mk, **opts)
    assert gpu["converged"] and cpu["converged"]
    assert gpu["newton_its"] == cpu["newton_its"]
    # identical counts; +-1 tolerated where a residual sits on the threshold
    diff = [abs(a - b) for a, b in zip(gpu["krylov_per_step"],
                                       cpu["krylov_per_step"])]
    assert max(diff) <= 1, (gpu["krylov_per_step"], cpu["krylov_per_step"])
    assert relerr(gpu["w"].vector(), cpu["w"].vector()) < 1e-6
    # outlet flux equals inlet flux for the converged solution: q = 1 is a
    # P1 test function, so the discrete continuity equation balances the two
    # to solver accuracy; the inflow 4y(1-y) is a P2 function, its flux is
    # integrated exactly by Simpson's rule on every edge: 2/3
    pb = mk()
    V = pb.space
    for run in (gpu, cpu):
        U = run["w"].split()[0].reshape(-1, 2)
        flux = {}
        for name, edges in (("in", pb.inlet_edges), ("out", pb.outlet_edges)):
            pl = V.robin_plan(edges)
            nodes = pl["nodes"]
            wmean = (U[nodes[:, 0]] + U[nodes[:, 1]] + 4.0 * U[nodes[:, 2]]) / 6.0
            flux[name] = float(((wmean * pl["normal"]).sum(axis=1)
                                * pl["length"]).sum())
        assert abs(flux["in"] + 2.0 / 3.0) < 1e-12, flux     # outward normal
        assert abs(flux["out"] - 2.0 / 3.0) < 1e-4, flux


def test_cavity_steady_converges():
    PETScOptions.clear()
    default_inner_options(a00_its=40, a00_ratio=0.02, ap_rtol=1e-10)
    out = solve_steady(Cavity(2, nu=0.05))
    assert out["converged"]
    assert max(out["krylov_per_step"]) < 150


def test_unsteady_pcd_and_pcdr_run():
    for pcdr in (False, True):
        PETScOptions.clear()
        default_inner_options(a00_its=10, a00_ratio=0.1, ap_rtol=1e-10,
                              pcdr=pcdr)
        pb = BackwardStep(2, nu=0.02, dt=0.2, pcdr=pcdr)
        out = solve_unsteady(pb, dt=0.2, t_end=0.6, newton_rtol=1e-5)
        assert out["steps"] == 3
        assert out["krylov_its"] > 0


def test_multigrid_options_same_counts_as_cpu_restatement(monkeypatch):
    from fenapack_amd.driver import multigrid_inner_options

    def run(lib):
        if lib is not None:
            monkeypatch.setattr(c, "hip_library", lambda: lib)
        PETScOptions.clear()
        multigrid_inner_options()
        return solve_steady(Cavity(3, nu=0.01), newton_rtol=1e-5,
                            gmres_rtol=1e-6)
    gpu = run(None)
    cpu = run(oracle.library())
    assert gpu["converged"] and cpu["converged"]
    assert gpu["krylov_per_step"] == cpu["krylov_per_step"]
    assert relerr(gpu["w"].vector(), cpu["w"].vector()) < 1e-6
    assert max(gpu["krylov_per_step"]) < 60


def test_algebraic_multigrid_options_of_the_reference_demo(monkeypatch):
    """The reference demo's "iterative" option strings
    (demo_navier-stokes-pcd.py:152-165: Richardson + hypre BoomerAMG for A00
    and Ap, Chebyshev + Jacobi for Mp) run unchanged: hypre is served by the
    engine's algebraic hierarchy (smoothed aggregation from the matrix alone,
    no mesh).  Same GMRES counts on the HIP engine and on the oracle."""
    def run(lib):
        if lib is not None:
            monkeypatch.setattr(c, "hip_library", lambda: lib)
        PETScOptions.clear()
        S = PETScOptions.set
        for key in ("fieldsplit_u_", "fieldsplit_p_PCD_Ap_"):
            S(key + "ksp_type", "richardson")
            S(key + "ksp_max_it", 1 if key.startswith("fieldsplit_u") else 2)
            S(key + "pc_type", "hypre")
            S(key + "pc_hypre_type", "boomeramg")
        S("fieldsplit_p_PCD_Mp_ksp_type", "chebyshev")
        S("fieldsplit_p_PCD_Mp_ksp_max_it", 5)
        S("fieldsplit_p_PCD_Mp_ksp_chebyshev_eigenvalues", "0.5, 2.0")
        S("fieldsplit_p_PCD_Mp_pc_type", "jacobi")
        out = solve_steady(BackwardStep(3, nu=0.02, variant="BRM2"),
                           newton_rtol=1e-5, gmres_rtol=1e-6)
        PETScOptions.clear()
        return out
    gpu = run(None)
    cpu = run(oracle.library())
    assert gpu["converged"] and cpu["converged"]
    assert gpu["krylov_per_step"] == cpu["krylov_per_step"]
    assert relerr(gpu["w"].vector(), cpu["w"].vector()) < 1e-6
    assert max(gpu["krylov_per_step"]) < 120
    ksp0 = gpu["solver"].linear_solver().ksp().pc.getFieldSplitSubKSP()[0]
    assert ksp0.pc.mg_algebraic and len(ksp0.pc.mg_data["ops"]) >= 2


def test_algebraic_hierarchy_in_three_dimensions(monkeypatch):
    """-pc_type gamg on a 3-D operator (three components per node, every
    boundary node a Dirichlet row - the rows that used to stall the
    aggregation): the hierarchy goes below the coarse limit, the HIP engine
    and the oracle count the same outer iterations.  Config 5's own mesh
    (N = 73, no nested hierarchy) runs this path:
    profiles/r03_gamg_cube_n73_config5_size.json."""
    from fenapack_amd.driver import multigrid_inner_options
    from fenapack_amd.fem import Cavity3D

    def run(lib):
        if lib is not None:
            monkeypatch.setattr(c, "hip_library", lambda: lib)
        PETScOptions.clear()
        multigrid_inner_options(dim=3, algebraic=True)
        out = solve_steady(Cavity3D(0, nu=0.01, n0=10), newton_rtol=0.0,
                           max_newton=2, gmres_rtol=1e-6)
        PETScOptions.clear()
        return out
    gpu = run(None)
    cpu = run(oracle.library())
    assert gpu["krylov_per_step"] == cpu["krylov_per_step"]
    assert max(gpu["krylov_per_step"]) < 80
    assert relerr(gpu["w"].vector(), cpu["w"].vector()) < 1e-6
    ksp0 = gpu["solver"].linear_solver().ksp().pc.getFieldSplitSubKSP()[0]
    ops = ksp0.pc.mg_data["ops"]
    assert ksp0.pc.mg_algebraic and len(ops) >= 2
    assert ops[0].shape[0] <= ksp0.pc.mg_coarse_eq_limit


def test_bench_line_of_the_re1000_supg_configuration():
    """``bench.py --re 1000 --supg --rediscretise-u`` - BASELINE configs[2]'s
    flow (Re = 1000, SUPG-stabilised preconditioner matrix,
    fenapack/stabilization.py:66-67) through the bench contract, in small
    (level 4): one JSON line that names the workload, carries the oracle's
    parity of this very state, and withholds the whole-apply PMC traffic
    (no counter pass exists for this inner configuration).  --supg without
    --rediscretise-u is refused."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--level", "4",
           "--re", "1000", "--supg", "--steps", "5", "--warmup", "2",
           "--cycles-u", "2", "--cycles-p", "2", "--smooth", "3",
           "--no-producer", "--cpu-seconds", "1"]
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "--rediscretise-u" in bad.stderr
    run = subprocess.run(cmd + ["--rediscretise-u"], capture_output=True,
                         text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"] == "fieldsplit PCApply calls/sec (2D cavity Re=1000, P2/P1)"
    assert d["config"]["workload"] == \
        "cavity level 4, Re=1000, P2/P1, PCD BRM1 + SUPG"
    assert d["value"] > 0 and len(d["gmres_its_per_newton_step"]) == 2
    assert d["cpu_baseline"]["gpu_vs_oracle_rel_err"] < 1e-11
    assert d["pcapply_roofline"]["traffic"] is None
    assert d["pcapply_roofline"]["traffic_stale"]


def test_bench_line_of_the_north_stars_literal_solvers():
    """``bench.py --inner jacobi`` - the solvers `north_star` names: a
    Jacobi-preconditioned CG (wave64 reductions) for the pressure Laplacian
    (fenapack/preconditioners.py:42-49, 130: `ksp_Ap.solve`), Chebyshev-Jacobi
    for the mass matrix (:133) and a Chebyshev / Jacobi sweep for the velocity
    block - through the bench contract in small (level 3): the line carries
    the `cg` block (two launches per iteration on one rank, B_cg of SURVEY
    8(d) over the measured time), the executed k_A, and the oracle's parity of
    this very apply (a tolerance-driven CG: the iteration counts agree, the
    results to cond(Ap) * rtol - not the fixed-count paths' 1e-11)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--level", "3",
           "--inner", "jacobi", "--steps", "5", "--warmup", "2",
           "--a00-its", "60", "--a00-ratio", "0.01", "--cpu-seconds", "1"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] > 0 and len(d["gmres_its_per_newton_step"]) == 2
    assert d["config"]["inner"]["Ap"].startswith("cg+jacobi")
    cg = d["cg"]
    assert "error" not in cg, cg
    assert cg["launches_per_iteration"] == 2.0, cg
    assert cg["executed_k_A_last_apply"] > 10
    n_p, = [d["config"]["n_p"]]
    assert cg["algorithmic_bytes_per_iteration"] > 148 * n_p
    assert 0.0 < cg["frac"] < 1.0 and cg["us_per_iteration"] > 1.0
    # the CG stops by its tolerance: engine and oracle execute the same number
    # of iterations, and their results agree to cond(Ap) * rtol, not to
    # rounding (level 6: 1931 iterations either side, 1.8e-4)
    ke, ko = d["cpu_baseline"]["k_A_engine_and_oracle"]
    assert abs(ke - ko) <= 1 and ke == cg["executed_k_A_last_apply"], (ke, ko)
    assert d["cpu_baseline"]["gpu_vs_oracle_rel_err"] < 1e-4
