"""GPU suite, part 3: the row-partitioned multi-rank path, exercised on ONE
GPU through the in-process threaded comm backend (RCCL refuses two ranks on a
single device).  R engine handles = R ranks, one thread each; host-pointer
calls carry global vectors, so results compare directly with one engine."""
import ctypes
import threading

import numpy as np
import pytest

import oracle
from fenapack_amd import _cabi as c
from helpers import (flow_state, configure_engine, relerr, set_iter_cfg,
                     push_multigrid)

pytestmark = pytest.mark.gpu


def run_ranks(hip_lib, R, variant, work):
    """``work(engine, rank)`` on R ranks in lockstep; returns the results."""
    group = ctypes.c_void_p()
    engines = []
    for r in range(R):
        e = c.Engine(hip_lib, variant, 0)
        e.comm_init_threads(r, R, group)
        engines.append(e)
    out, errs = [None] * R, []

    def body(r):
        try:
            out[r] = work(engines[r], r)
        except Exception as ex:            # pragma: no cover
            errs.append((r, ex))

    threads = [threading.Thread(target=body, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "ranks deadlocked"
    assert not errs, errs
    return out


@pytest.mark.parametrize("R", [2, 3])
def test_spmv_and_apply_partitioned_vs_single(hip_lib, R):
    st = flow_state("lshape", 3, dt=0.2)
    V = st["V"]
    rng = np.random.default_rng(20)
    xp, xu, xs = (rng.standard_normal(V.n_p), rng.standard_normal(V.n_u),
                  rng.standard_normal(V.ndof))

    def work(e, rank):
        configure_engine(e, st)
        set_iter_cfg(e)
        e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
        e.setup()
        res = {"np_loc": e.info(c.INFO_N_P_LOCAL),
               "nu_loc": e.info(c.INFO_N_U_LOCAL)}
        res["Kp"] = e.spmv_np(c.MAT_KP, xp, V.n_p)
        res["A00"] = e.spmv_np(c.MAT_A00, xu, V.n_u)
        res["A01"] = e.spmv_np(c.MAT_A01, xp, V.n_u)
        res["A"] = e.spmv_np(c.MAT_A, xs, V.ndof)
        res["cg"] = e.inner_solve_np(c.KSP_AP, xp)
        res["cheb"] = e.inner_solve_np(c.KSP_A00, xu)
        res["pcd"] = e.apply_np(xp)
        res["fs"] = e.fieldsplit_apply_np(xs)
        e.set_inner(c.KSP_AP, "cg", "jacobi", 3000, 1e-10)
        res["cg_tol"] = e.inner_solve_np(c.KSP_AP, xp)
        res["cg_its"] = e.info(c.INFO_ITS_AP)
        # -ksp_cg_single_reduction: one 2-double all-reduce per iteration
        e.set_inner(c.KSP_AP, "cgsr", "jacobi", 8, 0.0)
        res["cgsr"] = e.inner_solve_np(c.KSP_AP, xp)
        e.set_inner(c.KSP_AP, "cgsr", "jacobi", 3000, 1e-10)
        res["cgsr_tol"] = e.inner_solve_np(c.KSP_AP, xp)
        res["cgsr_its"] = e.info(c.INFO_ITS_AP)
        return res

    outs = run_ranks(hip_lib, R, "RBRM1", work)
    ref = work(oracle.Engine("RBRM1"), 0)
    assert sum(o["np_loc"] for o in outs) == V.n_p
    assert sum(o["nu_loc"] for o in outs) == V.n_u
    assert all(o["nu_loc"] % 2 == 0 for o in outs)
    for o in outs:                       # every rank returns the full vector
        for key in ("Kp", "A00", "A01", "A"):
            assert relerr(o[key], ref[key]) < 1e-13, key
        for key in ("cg", "cheb", "pcd", "fs"):
            assert relerr(o[key], ref[key]) < 1e-11, key
        assert abs(o["cg_its"] - ref["cg_its"]) <= 1
        assert relerr(o["cg_tol"], ref["cg_tol"]) < 1e-8
        assert relerr(o["cgsr"], ref["cgsr"]) < 1e-11
        assert abs(o["cgsr_its"] - ref["cgsr_its"]) <= 1
        assert relerr(o["cgsr_tol"], ref["cgsr_tol"]) < 1e-8


@pytest.mark.parametrize("replicate_below", ["60000", "0", "700"])
def test_gmres_with_multigrid_partitioned(hip_lib, monkeypatch,
                                          replicate_below):
    # coarse levels replicated on every rank (default), fully distributed
    # (0), or mixed (700 rows: both kinds of level and the transition)
    monkeypatch.setenv("PCD_REPLICATE_BELOW", replicate_below)
    st = flow_state("cavity", 3)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()

    def work(e, rank):
        configure_engine(e, st)
        push_multigrid(e, c.KSP_AP, pb.Ap, I.chain("p"))
        push_multigrid(e, c.KSP_A00, L["A00"], I.chain("u"))
        e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
        e.setup()
        x, its, rn = e.gmres_np(st["b"], rtol=1e-8, restart=60, max_it=200)
        # values refreshed between Newton steps keep working
        e.update_system(st["A"].data)
        y = e.fieldsplit_apply_np(st["b"])
        return x, its, y

    outs = run_ranks(hip_lib, 2, "BRM1", work)
    xr, ir, yr = work(oracle.Engine("BRM1"), 0)
    for x, its, y in outs:
        assert its == ir
        assert relerr(x, xr) < 1e-7
        assert relerr(y, yr) < 1e-11
    assert relerr(st["A"] @ outs[0][0], st["b"]) < 1e-6


def test_rccl_backend_single_rank_smoke(hip_lib, monkeypatch):
    """The RCCL backend itself (dlopen, unique id, ncclCommInitRank,
    ncclAllReduce on the engine's stream) with a ONE-rank communicator: the
    multi-rank code path (localised operators, slot reductions, global-vector
    staging) must reproduce the plain single-GPU results."""
    monkeypatch.setenv("PCD_FORCE_COMM", "1")
    st = flow_state("cavity", 3)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()

    def run(e):
        configure_engine(e, st)
        push_multigrid(e, c.KSP_A00, L["A00"], I.chain("u"))
        e.set_inner(c.KSP_AP, "cg", "jacobi", 2000, 1e-10)
        e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
        e.setup()
        return e.gmres_np(st["b"], rtol=1e-8, restart=60, max_it=200)

    e = c.Engine(hip_lib, "BRM1", 0)
    e.comm_init(0, 1, c.comm_unique_id())
    x1, its1, _ = run(e)
    monkeypatch.delenv("PCD_FORCE_COMM")
    x0, its0, _ = run(c.Engine(hip_lib, "BRM1", 0))
    assert its1 == its0
    assert relerr(x1, x0) < 1e-9


def test_three_dimensional_problem_partitioned(hip_lib):
    """3D (three velocity components per node): single engine and 2 / 3
    threaded ranks against the oracle; row cuts must fall on node boundaries."""
    st = flow_state("cube", 1, nu=0.1)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()

    def work(e, rank):
        configure_engine(e, st)
        push_multigrid(e, c.KSP_AP, pb.Ap, I.chain("p"))
        push_multigrid(e, c.KSP_A00, L["A00"], I.chain("u"))
        e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 8, 0.0, 0.3, 2.6)
        e.setup()
        x, its, _ = e.gmres_np(st["b"], rtol=1e-8, restart=80, max_it=200)
        return x, its, e.info(c.INFO_N_U_LOCAL)

    xr, ir, _ = work(oracle.Engine("BRM1"), 0)
    x1, i1, _ = work(c.Engine(hip_lib, "BRM1", 0), 0)
    assert i1 == ir and relerr(x1, xr) < 1e-7
    for R in (2, 3):
        for x, its, nu_loc in run_ranks(hip_lib, R, "BRM1", work):
            assert its == ir and relerr(x, xr) < 1e-7
            assert nu_loc % 3 == 0


@pytest.mark.parametrize("form", ["2", "2staged"])
@pytest.mark.parametrize("kind,level,nu,R", [("cavity", 3, 0.01, 3),
                                             ("cube", 1, 0.1, 2)])
def test_tile_kernels_read_their_ghost_columns(hip_lib, monkeypatch, form,
                                               kind, level, nu, R):
    """The LDS-staged vector-tile kernels on row-partitioned operators: a
    block's tile holds nodes of OTHER ranks (tile sources >= nloc live in the
    ghost buffer).  By default the tile kernels start at 80 000 node rows per
    rank - beyond the suite's multi-rank sizes - so they are forced here
    (direct and staged form), every level distributed."""
    monkeypatch.setenv("PCD_VEC_TILE", "2")
    if form == "2staged":
        monkeypatch.setenv("PCD_NT_BYTES", "0")
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "0")
    st = flow_state(kind, level, nu=nu)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()
    rng = np.random.default_rng(5)
    xu = rng.standard_normal(V.n_u)

    def work(e, rank):
        configure_engine(e, st)
        push_multigrid(e, c.KSP_AP, pb.Ap, I.chain("p"))
        push_multigrid(e, c.KSP_A00, L["A00"], I.chain("u"))
        e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.3, 2.6)
        e.setup()
        x, its, _ = e.gmres_np(st["b"], rtol=1e-8, restart=80, max_it=200)
        return (x, its, e.spmv_np(c.MAT_A00, xu, V.n_u),
                e.inner_solve_np(c.KSP_A00, xu))

    xr, ir, yr, zr = work(oracle.Engine("BRM1"), 0)
    for x, its, y, z in run_ranks(hip_lib, R, "BRM1", work):
        assert relerr(y, yr) < 1e-13
        assert relerr(z, zr) < 1e-11
        assert its == ir and relerr(x, xr) < 1e-7


@pytest.mark.parametrize("R,dt", [(2, None), (3, None), (8, None),
                                  (2, 0.2), (3, 0.2)])
def test_device_producer_plans_cut_by_rows(hip_lib, monkeypatch, R, dt):
    """Several ranks, Picard, re-discretised coarse levels: every rank hands
    the engine the cells and contribution lists of ITS node rows of each
    partitioned level only (pcd_fe_set_rows) - element work and plan memory
    of a rank are 1 / R of the level's; the nonlinear side's VECTORS stay
    replicated, the unconstrained operator of the residual is applied by rows
    through the engine's own partitioned A00.  Against the replicated plans
    (FENAPACK_AMD_FE_ROWS=0) on the same ranks: the same nonlinear history, the
    same Krylov counts, the same solution; operators of the owned rows equal."""
    from fenapack_amd import PETScOptions
    from fenapack_amd.device_producer import (solve_steady_device,
                                              solve_unsteady_device)
    from fenapack_amd.driver import multigrid_inner_options
    from fenapack_amd.fem import BackwardStep, Cavity
    from fenapack_amd.parallel import Comm
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "300")

    def run(rows):
        monkeypatch.setenv("FENAPACK_AMD_FE_ROWS", "1" if rows else "0")
        PETScOptions.clear()
        multigrid_inner_options(dim=2, galerkin_u=False)
        PETScOptions.set("fieldsplit_u_pc_mg_coarse_eq_limit", 300)
        group = ctypes.c_void_p()
        comms = [Comm(r, R, thread_group=group) for r in range(R)]
        out, errs = [None] * R, []

        def body(r):
            try:
                if dt:
                    pb = BackwardStep(3, nu=0.02, dt=dt)
                    o = solve_unsteady_device(pb, dt=dt, t_end=3 * dt,
                                              newton_rtol=1e-5, comm=comms[r])
                    hist = o["krylov_per_newton"]
                else:
                    # (eight rank threads each hold the global host problem:
                    # a level less keeps the test inside its memory budget)
                    pb = Cavity(4 if R == 8 else 5, nu=0.01)
                    o = solve_steady_device(pb, max_newton=8, comm=comms[r])
                    hist = [o["krylov_per_step"]]
                prod = o["producer"]
                top = prod.nlev - 1
                out[r] = {"its": o["newton_its"], "hist": hist,
                          "x": o["w"].vector().copy(), "rows": prod.rows,
                          "plan": list(prod.plan_entries),
                          "F": prod._scalar(top), "cut": prod._cut[top],
                          "cells": pb.space.mesh.num_cells}
            except Exception as ex:            # pragma: no cover
                import traceback
                errs.append((r, repr(ex), traceback.format_exc()))

        th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in th), "ranks deadlocked"
        assert not errs, errs
        return out

    rep, cut = run(False), run(True)
    PETScOptions.clear()
    assert all(not o["rows"] for o in rep) and all(o["rows"] for o in cut)
    for a, b in zip(cut, rep):
        assert a["its"] == b["its"] and a["hist"] == b["hist"]
        assert relerr(a["x"], b["x"]) < 1e-10
        # this rank's rows of the finest operator: what the replicated plan
        # assembled for the same rows (at iterates that agree to round-off: the
        # residual's slices are summed across the ranks in another order)
        a0, a1 = a["cut"][0], a["cut"][1]
        Fa, Fb = a["F"][a0:a1], b["F"][a0:a1]
        assert np.array_equal(Fa.indices, Fb.indices)
        assert relerr(Fa.data, Fb.data) < 1e-9
        assert a["F"].nnz == Fa.nnz                      # nothing outside its rows
    # plan memory: cells and entries of the finest level per rank ~ 1 / R
    ncell = cut[0]["cells"]
    tot_cells = sum(o["plan"][-1][0] for o in cut)
    tot_ent = sum(o["plan"][-1][1] for o in cut)
    assert tot_ent == rep[0]["plan"][-1][1]               # the rows partition the operator
    assert ncell <= tot_cells <= 1.35 * ncell             # (+ one layer of cells per cut)
    assert max(o["plan"][-1][0] for o in cut) <= 1.4 * ncell / R
    print("\nfinest level: %d cells / %d entries replicated; per rank %s"
          % (ncell, rep[0]["plan"][-1][1], [o["plan"][-1] for o in cut]))


@pytest.mark.parametrize("R,galerkin,dt,nls", [
    (2, True, None, "picard"), (3, False, None, "picard"),
    (2, True, 0.2, "picard"), (2, True, None, "newton"),
    (3, False, 0.2, "newton")])
def test_device_producer_on_several_ranks(hip_lib, R, galerkin, dt, nls):
    """Config 4 shape without the host producer: the nonlinear side
    (assembly of every level, residual) replicated on every rank, the linear
    solve partitioned; R ranks = R threads on this one GPU.  Compared with
    the one-GPU device solve."""
    from fenapack_amd import PETScOptions
    from fenapack_amd.device_producer import (solve_steady_device,
                                              solve_unsteady_device)
    from fenapack_amd.driver import multigrid_inner_options
    from fenapack_amd.fem import BackwardStep, Cavity
    from fenapack_amd.parallel import Comm

    def problem():
        if dt:
            return BackwardStep(3, nu=0.02, dt=dt, nls=nls)
        return Cavity(5, nu=0.01, nls=nls)

    def solve(comm):
        pb = problem()
        if dt:
            out = solve_unsteady_device(pb, dt=dt, t_end=3 * dt,
                                        newton_rtol=1e-5, comm=comm)
            return out["newton_its"], out["krylov_per_newton"], \
                out["w"].vector().copy()
        out = solve_steady_device(pb, max_newton=8, comm=comm)
        assert out["converged"]
        eng = out["solver"].linear_solver().ksp().engine
        rows.append((eng.info(c.INFO_N_U_LOCAL), pb.space.n_u,
                     out["producer"].ranks))
        return out["newton_its"], [out["krylov_per_step"]], \
            out["w"].vector().copy()

    rows = []
    PETScOptions.clear()
    multigrid_inner_options(dim=2, galerkin_u=galerkin)
    PETScOptions.set("fieldsplit_u_pc_mg_coarse_eq_limit", 300)
    # both kinds of coarse level: replicated below 300 rows (the coarsest one,
    # which every rank inverts), partitioned above
    import os
    os.environ["PCD_REPLICATE_BELOW"] = "300"
    try:
        ref = solve(None)
        group = ctypes.c_void_p()
        comms = [Comm(r, R, thread_group=group) for r in range(R)]
        out, errs = [None] * R, []

        def body(r):
            try:
                out[r] = solve(comms[r])
            except Exception as ex:            # pragma: no cover
                errs.append((r, repr(ex)))

        threads = [threading.Thread(target=body, args=(r,)) for r in range(R)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in threads), "ranks deadlocked"
        assert not errs, errs
    finally:
        del os.environ["PCD_REPLICATE_BELOW"]
    for its, krylov, x in out:
        assert its == ref[0]
        for a, b in zip(krylov, ref[1]):
            assert len(a) == len(b)
            for i, j in zip(a, b):
                assert abs(i - j) <= max(1, 0.05 * j), (krylov, ref[1])
        assert relerr(x, ref[2]) < 1e-5
    if not dt:
        # the ranks really owned row blocks (rows[0] is the one-GPU run)
        assert rows[0][0] == rows[0][1] and rows[0][2] == 0
        assert sum(r[0] for r in rows[1:]) == rows[0][1]
        assert all(0 < r[0] < r[1] and r[2] == R for r in rows[1:])
    # every rank holds the same replicated iterate
    for its, krylov, x in out[1:]:
        assert relerr(x, out[0][2]) < 1e-12


@pytest.mark.parametrize("R,kind", [(2, "cavity"), (3, "cube")])
def test_device_producer_with_an_algebraic_hierarchy_on_several_ranks(hip_lib, R,
                                                                      kind):
    """-pc_type gamg with ranks and the global hand-over: the aggregation is
    the same on every rank, so prolongators and coarse patterns are replicated
    and the device producer refreshes the Galerkin products of every level in
    HBM as on one GPU (the reference re-runs hypre's set-up per outer
    iteration: demo_navier-stokes-pcd.py:153-160); the solve is partitioned.
    Same nonlinear history, Krylov counts and solution as the one-GPU device
    solve; the ranks own row blocks."""
    import os
    from fenapack_amd import PETScOptions
    from fenapack_amd.device_producer import solve_steady_device
    from fenapack_amd.driver import multigrid_inner_options
    from fenapack_amd.fem import Cavity, Cavity3D
    from fenapack_amd.parallel import Comm

    def problem():
        return Cavity(4, nu=0.01) if kind == "cavity" \
            else Cavity3D(1, nu=0.02, n0=6)

    rows = []

    def solve(comm):
        out = solve_steady_device(problem(), max_newton=8, comm=comm)
        assert out["converged"] and out["producer"].algebraic
        assert out["producer"].device_loop
        eng = out["solver"].linear_solver().ksp().engine
        rows.append((eng.info(c.INFO_N_U_LOCAL), out["producer"].ranks))
        return out["newton_its"], out["krylov_per_step"], \
            out["w"].vector().copy()

    PETScOptions.clear()
    multigrid_inner_options(dim=2 if kind == "cavity" else 3, algebraic=True)
    PETScOptions.set("fieldsplit_u_pc_mg_coarse_eq_limit", 100)
    os.environ["PCD_REPLICATE_BELOW"] = "300"
    try:
        ref = solve(None)
        group = ctypes.c_void_p()
        comms = [Comm(r, R, thread_group=group) for r in range(R)]
        out, errs = [None] * R, []

        def body(r):
            try:
                out[r] = solve(comms[r])
            except Exception as ex:            # pragma: no cover
                import traceback
                errs.append((r, repr(ex), traceback.format_exc()))

        threads = [threading.Thread(target=body, args=(r,)) for r in range(R)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=600)
        assert not any(t.is_alive() for t in threads), "ranks deadlocked"
        assert not errs, errs
    finally:
        del os.environ["PCD_REPLICATE_BELOW"]
        PETScOptions.clear()
    for its, krylov, x in out:
        assert its == ref[0]
        assert len(krylov) == len(ref[1])
        for i, j in zip(krylov, ref[1]):
            assert abs(i - j) <= max(1, 0.05 * j), (krylov, ref[1])
        assert relerr(x, ref[2]) < 1e-5
        assert relerr(x, out[0][2]) < 1e-12
    n_u = rows[0][0]
    assert rows[0][1] == 0 and sum(r[0] for r in rows[1:]) == n_u
    assert all(0 < r[0] < n_u and r[1] == R for r in rows[1:])


def test_device_producer_over_rccl_single_rank(hip_lib, monkeypatch):
    """The same replicated-producer path over the RCCL backend itself, with
    the one-rank communicator this box allows (all-reduces of whole vectors,
    slices through the rank-local permutation)."""
    from fenapack_amd import PETScOptions
    from fenapack_amd.device_producer import solve_steady_device
    from fenapack_amd.driver import multigrid_inner_options
    from fenapack_amd.fem import Cavity
    from fenapack_amd.parallel import Comm
    PETScOptions.clear()
    multigrid_inner_options(dim=2)
    ref = solve_steady_device(Cavity(4, nu=0.01), max_newton=8)
    monkeypatch.setenv("PCD_FORCE_COMM", "1")
    out = solve_steady_device(Cavity(4, nu=0.01), max_newton=8,
                              comm=Comm.world())
    # (an RCCL communicator left to interpreter shutdown after another one
    # has come and gone aborts in the library's own teardown: free it here)
    out["solver"].linear_solver().ksp().engine.destroy()
    assert out["producer"].ranks == 1 and ref["producer"].ranks == 0
    assert out["converged"] and out["newton_its"] == ref["newton_its"]
    for i, j in zip(out["krylov_per_step"], ref["krylov_per_step"]):
        assert abs(i - j) <= max(1, 0.05 * j)
    assert relerr(out["w"].vector(), ref["w"].vector()) < 1e-5


@pytest.mark.parametrize("R", [2, 3])
def test_rank_local_handover_equals_global_handover(hip_lib, R):
    """pcd_set_csr_local: every rank hands over ONLY its own rows (global
    column ids); ghost columns and send lists come from the set-up handshake.
    Same results as the hand-over of the global matrices, and value updates
    carry the rank's entries only."""
    import scipy.sparse as sp
    st = flow_state("lshape", 3, dt=0.2)
    pb, V = st["pb"], st["V"]
    L = st["L"]
    rng = np.random.default_rng(31)
    xp, xu = rng.standard_normal(V.n_p), rng.standard_normal(V.n_u)
    mats_p = {c.MAT_AP: pb.Ap, c.MAT_MP: pb.Mp, c.MAT_KP: st["Kp"],
              c.MAT_RP: st["Rp"]}

    def work(local):
        def body(e, rank):
            e.set_velocity_block(V.dim)
            if local:
                p0, p1 = e.row_range(V.n_p)
                u0, u1 = e.row_range(V.n_u, velocity=True)
                assert u0 % V.dim == 0
                # (a hand-over that is not this rank's block is refused)
                with pytest.raises(c.EngineError, match="owns"):
                    e.set_csr_local(c.MAT_AP, sp.csr_matrix(pb.Ap)[p0:p1 - 1]
                                    if p1 - p0 > 1 else
                                    sp.csr_matrix(pb.Ap)[p0:p0], pb.Ap.shape)
                for which, M in mats_p.items():
                    e.set_csr_local(which, sp.csr_matrix(M)[p0:p1], M.shape)
                e.set_csr_local(c.MAT_A00, sp.csr_matrix(L["A00"])[u0:u1],
                                L["A00"].shape)
                e.set_csr_local(c.MAT_A01, sp.csr_matrix(L["A01"])[u0:u1],
                                L["A01"].shape)
            else:
                for which, M in mats_p.items():
                    e.set_csr(which, M)
                e.set_csr(c.MAT_A00, L["A00"])
                e.set_csr(c.MAT_A01, L["A01"])
            e.set_bc(pb.bc_p_idx, pb.bc_p_val)
            set_iter_cfg(e)
            e.setup()
            res = {"Kp": e.spmv_np(c.MAT_KP, xp, V.n_p),
                   "A00": e.spmv_np(c.MAT_A00, xu, V.n_u),
                   "A01": e.spmv_np(c.MAT_A01, xp, V.n_u),
                   "pcd": e.apply_np(xp)}
            # new values: the rank's own entries (local) / everything (global)
            K2 = sp.csr_matrix(st["Kp"]) * 1.5
            if local:
                p0, p1 = e.row_range(V.n_p)
                e.update_values(c.MAT_KP, K2[p0:p1].data)
            else:
                e.update_values(c.MAT_KP, K2.data)
            res["Kp2"] = e.spmv_np(c.MAT_KP, xp, V.n_p)
            return res
        return body

    glob = run_ranks(hip_lib, R, "RBRM1", work(False))
    loc = run_ranks(hip_lib, R, "RBRM1", work(True))
    for r in range(R):
        for k in glob[r]:
            assert np.array_equal(loc[r][k], glob[r][k]), (r, k)
    assert relerr(loc[0]["Kp2"], 1.5 * (st["Kp"] @ xp)) < 1e-13


def _push_multigrid_local(e, slot, A, chain, velocity, limit, nu=2, ratio=0.1):
    """helpers.push_multigrid with every PARTITIONED level handed over as this
    rank's rows only (pcd_mg_set_level_local); replicated levels - at most
    ``limit`` rows - go over whole, as they live on every rank."""
    import scipy.sparse as sp
    from fenapack_amd.fem.multigrid import galerkin_chain, coarse_inverse
    from fenapack_amd.petsc import estimate_emax
    ops = galerkin_chain(A, chain)
    L = len(ops)
    e.mg_begin(slot, L, nu, nu)
    e.mg_set_level(slot, 0, coarse_inverse(ops[0]))
    n_local = 0
    for l in range(1, L):
        emax = 1.1 * estimate_emax(ops[l], iters=12)
        n, nc = ops[l].shape[0], ops[l - 1].shape[0]
        if l < L - 1 and n <= limit:
            # (a replicated level lives whole on every rank: its rows alone
            # are refused, before anything is exchanged)
            r0, r1 = e.row_range(n, velocity=velocity)
            with pytest.raises(c.EngineError, match="replicated"):
                e.mg_set_level_local(slot, l, n, sp.csr_matrix(ops[l])[r0:r1],
                                     sp.csr_matrix(chain[l])[r0:r1], None,
                                     ratio * emax, emax)
            e.mg_set_level(slot, l, ops[l], chain[l], ratio * emax, emax)
            continue
        r0, r1 = e.row_range(n, velocity=velocity)
        P = sp.csr_matrix(chain[l])
        R_rows = None
        if nc > limit:
            c0, c1 = e.row_range(nc, velocity=velocity)
            R_rows = sp.csr_matrix(P.T)[c0:c1]
        A_rows = sp.csr_matrix(ops[l])[r0:r1] if l < L - 1 else None
        if R_rows is not None:
            # (the level below is partitioned too: its restriction rows are
            # part of the hand-over; a block that is not this rank's is refused)
            with pytest.raises(c.EngineError, match="restriction"):
                e.mg_set_level_local(slot, l, n, A_rows, P[r0:r1], None,
                                     ratio * emax, emax)
        with pytest.raises(c.EngineError, match="owns"):
            e.mg_set_level_local(slot, l, n, None if A_rows is None
                                 else A_rows[:-1], P[r0:r1 - 1], R_rows,
                                 ratio * emax, emax)
        e.mg_set_level_local(slot, l, n, A_rows, P[r0:r1], R_rows,
                             ratio * emax, emax)
        n_local += 1
    e.set_inner(slot, "richardson", "mg", 1, 0.0)
    return n_local


@pytest.mark.parametrize("R,replicate_below", [(2, "0"), (3, "700"),
                                               (2, "60000")])
def test_rank_local_system_and_multigrid_equal_the_global_handover(
        hip_lib, monkeypatch, R, replicate_below):
    """pcd_set_system_local / pcd_mg_set_level_local: a rank hands over its
    own rows of the monolithic matrix and of every partitioned multigrid
    level; nothing global but the index sets crosses the boundary.  Same
    GMRES history and PCApply as the global hand-over, also after a value
    refresh with the rank's own values."""
    import scipy.sparse as sp
    monkeypatch.setenv("PCD_REPLICATE_BELOW", replicate_below)
    limit = int(replicate_below)
    st = flow_state("cavity", 3)
    pb, V, Lz = st["pb"], st["V"], st["L"]
    I = pb.interpolations()
    A = sp.csr_matrix(st["A"])
    n = A.shape[0]
    A2 = A.copy()
    A2.data = A2.data * (1.0 + 0.01 * np.sin(np.arange(A2.nnz)))
    mats_p = {c.MAT_AP: pb.Ap, c.MAT_MP: pb.Mp, c.MAT_KP: st["Kp"]}

    def work(local):
        def body(e, rank):
            e.set_velocity_block(V.dim)
            if local:
                p0, p1 = e.row_range(V.n_p)
                u0, u1 = e.row_range(V.n_u, velocity=True)
                for which, M in mats_p.items():
                    e.set_csr_local(which, sp.csr_matrix(M)[p0:p1], M.shape)
                e.set_bc(pb.bc_p_idx, pb.bc_p_val)
                rows = np.concatenate([V.is_u[u0:u1], V.is_p[p0:p1]])
                # (rows that are not this rank's blocks are refused)
                with pytest.raises(c.EngineError, match="expected|owns"):
                    e.set_system_local(A[rows[::-1]], rows[::-1], n, V.is_u,
                                       V.is_p)
                e.set_system_local(A[rows], rows, n, V.is_u, V.is_p)
                nl = _push_multigrid_local(e, c.KSP_AP, pb.Ap, I.chain("p"),
                                           False, limit)
                nl += _push_multigrid_local(e, c.KSP_A00, Lz["A00"],
                                            I.chain("u"), True, limit)
                # (the finest level of a hierarchy is always partitioned)
                assert nl >= 2 and (limit > 0 or nl > 2)
            else:
                configure_engine(e, st)
                push_multigrid(e, c.KSP_AP, pb.Ap, I.chain("p"))
                push_multigrid(e, c.KSP_A00, Lz["A00"], I.chain("u"))
            e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
            e.setup()
            x, its, rn = e.gmres_np(st["b"], rtol=1e-8, restart=60,
                                    max_it=200)
            y = e.fieldsplit_apply_np(st["b"])
            if local:
                e.update_system(A2[rows].data)
            else:
                e.update_system(A2.data)
            x2, its2, _ = e.gmres_np(st["b"], rtol=1e-8, restart=60,
                                     max_it=200)
            return x, its, y, x2, its2
        return body

    glob = run_ranks(hip_lib, R, "BRM1", work(False))
    loc = run_ranks(hip_lib, R, "BRM1", work(True))
    for g, l in zip(glob, loc):
        assert l[1] == g[1] and l[4] == g[4]
        for a, b in ((l[0], g[0]), (l[2], g[2]), (l[3], g[3])):
            assert np.array_equal(a, b)
    assert relerr(A @ loc[0][0], st["b"]) < 1e-6
    assert relerr(A2 @ loc[0][3], st["b"]) < 1e-6
