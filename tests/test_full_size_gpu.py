"""GPU suite, part 5: parity at the sizes BASELINE.json names.

The other GPU tests compare the HIP engine with the oracle on meshes the
oracle's set-up finishes in a blink; here the workloads are the benchmark's own
(`bench.py`): the Python stack drives the HIP engine through two Picard steps
from w = 0 (so the operators carry real convection), `oracle.mirror` hands the
very same operators, hierarchies and smoother bounds to the C oracle, and one
fieldsplit PCApply - the benchmark's "step" - of each is compared on a seeded
vector.  Fixed-iteration inner solvers: tolerance 1e-11 (summation order is
the only difference).  The GMRES counts asserted are the engine's own history
at these settings: a change means the preconditioner changed; at
the headline size the oracle's GMRES runs next to the engine's on one
right-hand side and the counts must be identical.
"""
import numpy as np
import pytest

import oracle
from fenapack_amd import PETScOptions
from fenapack_amd import _cabi as c
from fenapack_amd.driver import make_solver, multigrid_inner_options
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D
from helpers import relerr

pytestmark = pytest.mark.gpu


def frozen_state(pb, picard_steps=2, exactly=False, **mg):
    """``exactly``: take ``picard_steps`` nonlinear iterations whatever the
    residual (on fine 3-D meshes the first, Stokes-like step already meets the
    demo's 1e-5 reduction, and the frozen operators would carry no
    convection: bench.py does the same)."""
    PETScOptions.clear()
    multigrid_inner_options(dim=pb.space.dim, **mg)
    w, nls, nlp = make_solver(pb, gmres_rtol=1e-6, restart=150,
                              newton_rtol=0.0 if exactly else 1e-5,
                              max_newton=picard_steps)
    if exactly:
        nls.parameters["absolute_tolerance"] = 0.0
    nls.parameters["error_on_nonconvergence"] = False
    nls.solve(nlp, w.vector(), on_update=w.touch)
    PETScOptions.clear()
    return nls.linear_solver().ksp(), list(nls.krylov_history)


def compare_with_oracle(pb, ksp, tol=1e-11, gmres=False):
    eng = ksp.engine
    V = pb.space
    o = oracle.mirror(oracle.Engine(("R" if pb.pcdr else "") + pb.variant),
                      pb, ksp)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(V.ndof)
    yg, yo = eng.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)
    assert np.isfinite(yo).all() and np.abs(yo).max() > 0
    err_fs = relerr(yg, yo)
    xp = rng.standard_normal(V.n_p)
    err_pcd = relerr(eng.apply_np(xp), o.apply_np(xp))
    # the two halves of the result separately: a wrong pressure block would
    # hide behind the (larger) velocity entries in a max-norm
    err_p = relerr(yg[V.is_p], yo[V.is_p])
    assert max(err_fs, err_pcd, err_p) < tol, (err_fs, err_pcd, err_p)
    if gmres:
        # the outer solve itself on both sides: right-preconditioned
        # GMRES(150), rtol 1e-6 (demo_navier-stokes-pcd.py:146-148) on a seeded
        # right-hand side in the operator's range (an enclosed flow's matrix
        # has the hydrostatic pressure mode: a random vector is not) -
        # IDENTICAL iteration counts, GPU engine vs CPU oracle
        b = ksp.getOperators()[0].A @ rng.standard_normal(V.ndof)
        xe, ie, _ = eng.gmres_np(b, rtol=1e-6, restart=150, max_it=300)
        xo, io, _ = o.gmres_np(b, rtol=1e-6, restart=150, max_it=300)
        assert ie == io and 0 < ie < 150, (ie, io)
        assert relerr(xe, xo) < 1e-7
    return err_fs


@pytest.mark.parametrize("variant,its", [("BRM1", [11, 30]),
                                         ("BRM2", [11, 28])])
def test_cavity_level6_headline_workload(variant, its):
    """BASELINE configs[1]: 924 803 DOF, Re = 100, bench.py's settings."""
    pb = Cavity(6, nu=0.01, variant=variant)
    assert pb.space.ndof == 924803
    ksp, hist = frozen_state(pb)
    assert [abs(a - b) <= 1 for a, b in zip(hist, its)] == [True, True], hist
    compare_with_oracle(pb, ksp, gmres=True)
    # hipGraph replay (what bench.py times) is bitwise the eager result
    x = np.random.default_rng(1).standard_normal(pb.space.ndof)
    y0 = ksp.engine.fieldsplit_apply_np(x)
    ksp.engine.graph_enable(True)
    y1 = ksp.engine.fieldsplit_apply_np(x)
    y2 = ksp.engine.fieldsplit_apply_np(x)
    ksp.engine.graph_enable(False)
    assert np.array_equal(y0, y1) and np.array_equal(y1, y2)


@pytest.mark.heavy(1)
@pytest.mark.rss_gb(20)
def test_cavity_level7_re1000_supg():
    """BASELINE configs[2]: 3 692 803 DOF, Re = 1000, SUPG-stabilised
    preconditioner matrix (stabilization.py), re-discretised coarse levels,
    2 x V(3,3) per inner solve (profiles/HISTORY_design_diary_rounds_1-3.md 9)."""
    pb = Cavity(7, nu=0.001, stabilize=True)
    assert pb.space.ndof == 3692803
    ksp, hist = frozen_state(pb, cycles_u=2, cycles_p=2, smooth=3,
                             galerkin_u=False)
    assert hist[0] <= 10 and hist[1] <= 90, hist
    A, P = ksp.getOperators()
    assert P is not A and P.isAssembled()       # J_pc differs from J
    compare_with_oracle(pb, ksp)


@pytest.mark.heavy(5)
@pytest.mark.rss_gb(12)
def test_cube_n32_three_components():
    """Config 5's geometry at the size one host can assemble: N = 32,
    859 812 DOF, F (x) I_3 kernels."""
    pb = Cavity3D(3, nu=0.01, n0=4)
    assert pb.space.ndof == 859812
    ksp, hist = frozen_state(pb)
    assert hist[0] <= 12 and hist[1] <= 50, hist
    assert int(ksp.engine.info(c.INFO_A00_COMPONENTS)) == 3
    compare_with_oracle(pb, ksp)


@pytest.mark.heavy(2)
@pytest.mark.rss_gb(26)
def test_cube_n48_config5_class():
    """BASELINE config 5's class on ONE GPU inside the suite's host-memory
    budget (32 GB, tests/conftest.py): the unit cube with N = 48 per side,
    2 855 668 DOF, F (x) I_3 kernels, TWO Picard steps from w = 0 so the frozen
    operators carry convection.  One fieldsplit PCApply and one PCD apply
    against the oracle.  Always runs: there is no skip for memory - the
    producer's estimate or the resident-set watchdog fail it instead.

    Config 5's own mesh (N = 73, 9.93 M DOF) has no nested hierarchy; it runs
    through the algebraic one, tools/parity_large.py --algebraic
    (profiles/r03_parity_cube73_config5_size_gamg.json, 3.2e-16; ~100 GB of
    host memory on one rank - outside this suite's budget); N = 64 (6.7 M
    DOF) is profiles/r03_i_parity_cube64.json."""
    pb = Cavity3D(3, nu=0.01, n0=6)
    assert pb.space.ndof == 2855668
    ksp, hist = frozen_state(pb, picard_steps=2, exactly=True)
    # (engine history at these settings: profiles/r03_i_parity_cube48.json)
    assert len(hist) == 2 and hist[0] <= 10 and hist[1] <= 46, hist
    assert int(ksp.engine.info(c.INFO_A00_COMPONENTS)) == 3
    compare_with_oracle(pb, ksp)


@pytest.mark.heavy(8)
@pytest.mark.timeout(1200)
def test_cube_n73_config5_own_mesh():
    """BASELINE config 5's OWN mesh on one GPU: the unit cube with N = 73 per
    side, 9 934 793 DOF.  No nested hierarchy exists for N = 73: the velocity
    and pressure solves run through the algebraic one (-pc_type gamg).  TWO
    Picard steps from w = 0, then one fieldsplit PCApply and one PCD apply
    against the oracle at 1e-11, and the engine's GMRES history.

    A process of its own (tools/parity_large.py): the producer, the hierarchy
    and the oracle's mirror hold tens of GB of host memory, which this
    session's 32 GB watchdog would refuse; the child runs under the watchdog
    of the repository's scripts (half of the memory the control group gives,
    fenapack_amd/_guard.py), so the budget is declared and enforced there."""
    import json
    import os
    import subprocess
    import sys
    from fenapack_amd import _guard
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = _guard.host_memory_available()
    need = 90e9        # (measured peak: 57 GB; the children run side by side only
    #                    where the host holds all of them: helpers.n73_children)
    assert have is None or have >= need, (
        "config 5's own mesh needs a host with %.0f GB available to this "
        "control group, %.0f GB here" % (need / 1e9, (have or 0) / 1e9))
    # (started together with the 8-thread-rank run of the same mesh -
    # test_configs_thread_ranks_gpu.py - so that the two share their wall time)
    from helpers import n73_result
    rc, so, se = n73_result("one_gpu")
    assert rc == 0, so[-2000:] + se[-4000:]
    rec = json.loads(so.strip().splitlines()[-1])
    print("cube N = 73:", rec)
    assert rec["ndof"] == 9934793 and rec["a00_components"] == 3
    # (engine history at these settings since round 3:
    # profiles/r03_parity_cube73_config5_size_gamg.json)
    hist = rec["gmres_its_per_step"]
    assert len(hist) == 2 and abs(hist[0] - 10) <= 1 and abs(hist[1] - 69) <= 2, hist
    assert rec["hip_vs_oracle_rel_err"] < 1e-11
    assert rec["pressure_block_rel_err"] < 1e-11
    assert rec["pcd_apply_rel_err"] < 1e-11
    assert rec["host_peak_rss_gb"] < 90.0
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_cube_n73_config5_own_mesh.json"), "w") as f:
            f.write(json.dumps(rec) + "\n")


@pytest.mark.parametrize("variant", ["BRM1", "BRM2"])
def test_channel_in_space_inflow_outflow(variant):
    """A 3-D inflow / outflow problem (square duct, N = 16, 112 724 DOF):
    pressure operators pinned on the inlet (BRM1) or the outlet (BRM2), and
    for BRM2 the Robin term over the inflow FACES in Kp
    (demo_navier-stokes-pcd.py:131-135; the form is dimension-free) - the
    whole PCApply and the PCD apply against the oracle, GMRES counts equal."""
    from fenapack_amd.fem import Channel3D
    pb = Channel3D(2, nu=0.02, n0=4, variant=variant)
    assert pb.space.ndof == 3 * 33 ** 3 + 17 ** 3
    ksp, hist = frozen_state(pb, picard_steps=3)
    assert max(hist) <= 60, hist
    if variant == "BRM2":
        V = pb.space
        Kp = ksp.pc.getFieldSplitSubKSP()[1].pc.getPythonContext().mat_Kp.A
        assert len(pb.robin_edges) == 2 * 16 * 16
        # (the boundary term is in the operator the engine holds)
        xu = np.zeros(V.n_u)
        assert Kp.shape == (V.n_p, V.n_p) and abs(Kp - pb.Kp(xu)).max() > 1e-3
    compare_with_oracle(pb, ksp, gmres=True)


def test_lshape_level6_reference_geometry():
    """The reference demo's own mesh at config-2 size (408 067 DOF), BRM2 with
    the Robin term in Kp (demo_navier-stokes-pcd.py:131-135)."""
    pb = BackwardStep(6, nu=0.02, variant="BRM2")
    assert pb.space.ndof == 408067
    ksp, hist = frozen_state(pb)
    assert hist[1] <= 45, hist
    compare_with_oracle(pb, ksp)


@pytest.mark.parametrize("pcdr,published,cycles,band", [
    (False, 3157, 2, (0.95, 1.25)), (True, 1686, 2, (0.95, 1.40)),
    # eight cycles per inner solve ~ the reference's exact factorisations:
    # "a few %" of the published totals; what is left is the
    # fifth Picard iteration this driver takes from step 6 on
    (False, 3157, 8, (0.97, 1.08)), (True, 1686, 8, (0.97, 1.12))])
def test_unsteady_anchor_against_the_published_table(pcdr, published, cycles,
                                                     band):
    """SURVEY 8c G4, the reference's only published numbers
    (demo/unsteady-navier-stokes-pcd/documentation.rst:134-140): L-shape
    level 4 (25 987 DOF), dt = 0.2, 25 steps, Picard, BRM1 -> 3157 Krylov
    iterations with PCD, 1686 with PCDR, both with EXACT inner solves (LU /
    Cholesky).  This engine replaces the factorisations by two multigrid
    cycles per inner solve and its own driver decides on the Picard
    iteration count; the bands state what that costs (DESIGN.md 5 explains
    both excesses: a fifth Picard iteration from step 6 on, and inexact
    R_p / A_p solves in PCDR - with 8 cycles the totals are 3266 / 1808 on the
    CPU restatement, profiles/r02_pcd*_cpu.log)."""
    from fenapack_amd.device_producer import solve_unsteady_device
    pb = BackwardStep(4, nu=0.02, variant="BRM1", dt=0.2, pcdr=pcdr,
                      dirichlet_diag="multiplicity")
    assert pb.space.ndof == 25987
    PETScOptions.clear()
    multigrid_inner_options(cycles_u=cycles, cycles_p=cycles, pcdr=pcdr)
    out = solve_unsteady_device(pb, dt=0.2, t_end=5.0, newton_rtol=1e-5,
                                gmres_rtol=1e-6)
    PETScOptions.clear()
    assert out["steps"] == 25
    lo, hi = band
    assert lo * published <= out["krylov_its"] <= hi * published, \
        (out["krylov_its"], out["krylov_per_step"])
    # per Picard iteration the count stays at the exact-solve level
    per_picard = [k for step in out["krylov_per_newton"] for k in step[1:]]
    assert max(per_picard) <= (24 if pcdr else 36), per_picard
    if cycles >= 8:
        # the published averages: 126.3 / 4 = 31.6 and 67.4 / 4 = 16.9
        assert max(per_picard) <= (19 if pcdr else 33), per_picard
        # EXACT inner solves under this driver (oracle/s_direct.py, scipy splu
        # in the role of the reference's LU / Cholesky; tools/s_direct.py,
        # profiles/r03_s_direct_unsteady_level4.jsonl): 3263 / 1801 - eight
        # cycles per inner solve are that operating point to 1.5 %
        exact = 1801 if pcdr else 3263
        assert abs(out["krylov_its"] - exact) <= 0.015 * exact, \
            (out["krylov_its"], exact)


@pytest.mark.heavy(6)
def test_cavity_level6_newton_block_on_the_device():
    """`--nls newton` at the headline size: the coupled velocity block
    F x I + N assembled on the DEVICE (pcd_fe_set_newton), as matrix and as
    the engine applies it, against the HOST producer at the same iterate."""
    from fenapack_amd.device_producer import solve_steady_device
    PETScOptions.clear()
    multigrid_inner_options(dim=2)
    pb = Cavity(6, nu=0.01, nls="newton")
    out = solve_steady_device(pb, max_newton=3, newton_rtol=0.0)
    PETScOptions.clear()
    assert out["krylov_per_step"] == [11, 39, 39]
    prod, V = out["producer"], pb.space
    assert prod.newton
    # the host producer at the device's final iterate
    x = out["w"].vector()
    lin = pb.linearise(x[V.is_u], x[V.is_p])
    import scipy.sparse.linalg as spla
    A00 = prod.level_matrix(prod.nlev - 1)
    assert abs(A00[0::2, 1::2]).max() > 1e-5                     # coupled
    assert spla.norm(A00 - lin["A00"]) < 1e-12 * spla.norm(lin["A00"])
    eng = out["solver"].linear_solver().ksp().engine
    rng = np.random.default_rng(1)
    xu = rng.standard_normal(V.n_u)
    assert relerr(eng.spmv_np(c.MAT_A00, xu, V.n_u), lin["A00"] @ xu) < 1e-12
    # (the GMRES counts are those of the host-driven Newton solve at these
    # settings: profiles/r02_ac_newton_level6_host_vs_device.json)


@pytest.mark.heavy(6)
@pytest.mark.rss_gb(14)
def test_cube_n32_newton_on_an_algebraic_hierarchy():
    """cube N = 32 (859 812 DOF), --nls newton through -pc_type gamg (the
    reference's bench sweeps nls x ls: test/bench/test_pcd_scaling.py:194-223).
    The chain prolongates every component alike (aggregates of the scalar
    stencil), the coarse operators are those of the COUPLED block.  (1) three
    host-driven Newton steps, then one fieldsplit PCApply and one PCD apply of
    the engine against the oracle mirrored from the same stack: 1e-11.  (2) the
    same three steps with the hierarchy refreshed ON THE DEVICE (numeric sparse
    products of the coupled block, d*d + 1 per level): the same Krylov history
    and iterate."""
    from fenapack_amd.device_producer import DevicePicardSolver
    pb = Cavity3D(3, nu=0.01, n0=4, nls="newton")
    assert pb.space.ndof == 859812
    ksp, hist = frozen_state(pb, picard_steps=3, exactly=True, algebraic=True)
    assert len(hist) == 3 and max(hist) <= 90, hist
    assert int(ksp.engine.info(c.INFO_A00_COMPONENTS)) == 0     # coupled block
    compare_with_oracle(pb, ksp)
    x_host = ksp.getOperators()[0]            # (keep the stack alive)
    PETScOptions.clear()
    multigrid_inner_options(dim=3, algebraic=True)
    s = DevicePicardSolver(Cavity3D(3, nu=0.01, n0=4, nls="newton"),
                           max_newton=3, newton_rtol=0.0)
    s.nls.parameters["absolute_tolerance"] = 0.0
    it, _ = s.solve()
    PETScOptions.clear()
    assert it == 3 and s.producer.newton and s.producer.algebraic
    assert s.producer.device_loop and s.producer.galerkin_mode == "product"
    dev = list(s.krylov_history)
    assert len(dev) == 3 and all(abs(a - b) <= 1 for a, b in zip(dev, hist)), (
        dev, hist)
    del x_host
