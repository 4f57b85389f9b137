"""GPU suite: the RANK-LOCAL device operator producer
(fenapack_amd/device_producer_rows.py) - config 5's path with the operator
refresh in HBM.

In the reference every rank re-assembles fp / kp for ITS rows each outer
iteration (fenapack/assembling.py:98-106; field_split_backend.py:79-83,
285-291; owned rows only: SubfieldBC.h:136-155) and re-runs the AMG set-up
(demo_navier-stokes-pcd.py:153-160).  Here, ranks as threads on the one GPU:
every rank feeds the engine from its slab, the partitioned algebraic hierarchy
(amg.PartitionedSA) is refreshed by numeric sparse products BY ROWS with the
terms of other ranks' coarse rows delivered over one all-reduced wire buffer.

* operators of every level, K_p and the residual against the HOST refresh of
  the same partitioned producer at the same iterate: 1e-12;
* the whole Picard solve against the host-driven one on the same ranks:
  identical Krylov history, same solution.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from fenapack_amd import PETScOptions, _host
from fenapack_amd.device_producer import DevicePicardSolver
from fenapack_amd.driver import multigrid_inner_options, solve_steady
from fenapack_amd.fem import Cavity3D, Channel3D
from fenapack_amd.fem import partition as pt

from test_partitioned_gpu import on_thread_ranks

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("N,R,limit", [(16, 2, 1500), (16, 3, 1500),
                                       (16, 8, 1500), (16, 2, 60000),
                                       # a duct with inflow / outflow, BRM2: the
                                       # Robin term of K_p over the inflow
                                       # faces of every rank's slab
                                       ("duct", 3, 1500)])
def test_operators_of_the_rank_local_producer_equal_the_host_refresh(
        hip_lib, monkeypatch, N, R, limit):
    """cube N = 16 on 2, 3 and 8 thread ranks.  PCD_REPLICATE_BELOW = 1500: two
    coarsenings run rank by rank (one coarse level stays partitioned, cut where
    the aggregates fall; the next is gathered); 60000: the first coarse level
    is gathered at once."""
    monkeypatch.setenv("PCD_REPLICATE_BELOW", str(limit))
    PETScOptions.clear()
    multigrid_inner_options(dim=3, algebraic=True)
    duct = N == "duct"
    cls = Channel3D if duct else Cavity3D
    kw = dict(level=0, nu=0.02, n0=16, variant="BRM2") if duct \
        else dict(level=0, nu=0.01, n0=N)

    def body(r, comm, host):
        pp = pt.partitioned(cls, r, R, host=host, **kw)
        V, d = pp.space, pp.space.dim
        s = DevicePicardSolver(pp, max_newton=2, newton_rtol=0.0, comm=comm)
        s.solve()                           # host step, then one device step
        prod = s.producer
        x = s.w.vector().copy()
        # a perturbed iterate (the same on every rank): operators carry real
        # convection, nothing is symmetric
        rng = np.random.default_rng(5)
        x = x + 0.05 * np.abs(x).max() * rng.standard_normal(x.size)
        b = prod.residual(x)                # refreshes every operator at x
        xu, xp = x[V.is_u], x[V.is_p]
        lin = pp.linearise(xu, xp)
        own_u, own_p = pp.fine.own_u, pp.fine.own_p
        out = {"levels": prod.nlev, "npart": len(prod.psa.part),
               "wire": list(prod.wire_doubles)}
        out["bu"] = _relerr(b[V.is_u][own_u[0]:own_u[1]],
                            lin["bu"][own_u[0]:own_u[1]])
        out["bp"] = _relerr(b[V.is_p][own_p[0]:own_p[1]],
                            lin["bp"][own_p[0]:own_p[1]])
        # K_p, my rows
        Kd, Kh = prod.kp_matrix(), sp.csr_matrix(pp.Kp(xu))
        out["kp"] = float(spla.norm(Kd - Kh) / spla.norm(Kh))
        # every level against the host refresh of the same hierarchy
        F = _host.kron_factor(sp.csr_matrix(lin["A00"]), d)
        ref = prod.psa.operators(F)[prod.lvl_off:]
        errs = []
        for l in range(prod.nlev):
            got, want = prod._scalar(l), sp.csr_matrix(ref[l])
            assert got.shape == want.shape, (l, got.shape, want.shape)
            errs.append(float(spla.norm(got - want)
                              / max(spla.norm(want), 1e-300)))
        out["ops"] = errs
        out["rows"] = [None if o is None else int(o[0].size - 1)
                       for o in prod._own]
        return out

    runs = on_thread_ranks(R, body)
    PETScOptions.clear()
    r0 = runs[0]
    # (limit 60000 at this size: the one coarse level is the gathered one)
    assert r0["levels"] >= (3 if limit == 1500 else 2)
    assert r0["npart"] == (2 if limit == 1500 else 1), r0
    if limit == 1500:
        # a partitioned coarse level exists, and something crosses the wire
        assert any(n is not None for n in r0["rows"][:-1]), r0
        assert sum(r0["wire"]) > 0
    for r in runs:
        assert r["bu"] < 1e-11 and r["bp"] < 1e-11, r
        assert r["kp"] < 1e-13, r
        assert max(r["ops"]) < 1e-12, r["ops"]


@pytest.mark.parametrize("R,limit", [(3, 1500), (8, 60000)])
def test_picard_solve_of_the_rank_local_producer_equals_the_host_driven_one(
        hip_lib, monkeypatch, R, limit):
    """The nonlinear loop on the device (pcd_fe_picard_solve) over the
    rank-local producer against the host-driven loop of the same partitioned
    producer on the same ranks: the same Krylov history (a count may move by
    one: smoother bounds come from a device power iteration) and solution."""
    monkeypatch.setenv("PCD_REPLICATE_BELOW", str(limit))
    PETScOptions.clear()
    multigrid_inner_options(dim=3, algebraic=True)
    kw = dict(level=0, nu=0.01, n0=16)

    def host_driven(r, comm, host):
        pp = pt.partitioned(Cavity3D, r, R, host=host, **kw)
        out = solve_steady(pp, max_newton=4, newton_rtol=0.0, comm=comm)
        return {"its": out["krylov_per_step"], "x": out["w"].vector().copy()}

    def on_device(r, comm, host):
        pp = pt.partitioned(Cavity3D, r, R, host=host, **kw)
        s = DevicePicardSolver(pp, max_newton=4, newton_rtol=0.0, comm=comm)
        s.solve()
        return {"its": list(s.krylov_history), "x": s.w.vector().copy(),
                "loop": bool(s.producer.device_loop)}

    ref = on_thread_ranks(R, host_driven)
    dev = on_thread_ranks(R, on_device)
    PETScOptions.clear()
    x1 = ref[0]["x"]
    for r in dev:
        assert r["loop"]
        assert np.array_equal(r["x"], dev[0]["x"])          # replicas identical
        assert len(r["its"]) == len(ref[0]["its"]) == 4
        assert all(abs(a - b) <= 1 for a, b in zip(r["its"], ref[0]["its"])), (
            r["its"], ref[0]["its"])
        assert np.abs(r["x"] - x1).max() <= 1e-6 * np.abs(x1).max()


def test_unsteady_loop_of_the_rank_local_producer(hip_lib, monkeypatch):
    """The unsteady demo's loop (demo_unsteady-navier-stokes-pcd.py:188-208:
    backward Euler, the mass term in F and K_p, the previous velocity in the
    residual) over the rank-local device producer: L-shape level 3, dt 0.2,
    three time steps on 2 thread ranks through -pc_type gamg, against the
    host-driven loop of the same partitioned producer."""
    from fenapack_amd.device_producer import solve_unsteady_device
    from fenapack_amd.driver import solve_unsteady
    from fenapack_amd.fem import BackwardStep
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "400")
    R, dt = 2, 0.2
    kw = dict(level=3, nu=0.02, dt=dt)

    def options():
        PETScOptions.clear()
        multigrid_inner_options(dim=2, algebraic=True, cycles_u=2, cycles_p=2)

    def host_driven(r, comm, host):
        pp = pt.partitioned(BackwardStep, r, R, host=host, **kw)
        out = solve_unsteady(pp, dt=dt, t_end=3 * dt, newton_rtol=1e-5,
                             comm=comm)
        return {"its": out["krylov_per_newton"], "x": out["w"].vector().copy()}

    def on_device(r, comm, host):
        pp = pt.partitioned(BackwardStep, r, R, host=host, **kw)
        out = solve_unsteady_device(pp, dt=dt, t_end=3 * dt, newton_rtol=1e-5,
                                    comm=comm)
        assert out["producer"].device_loop
        return {"its": out["krylov_per_newton"], "x": out["w"].vector().copy()}

    options()
    ref = on_thread_ranks(R, host_driven)
    options()
    dev = on_thread_ranks(R, on_device)
    PETScOptions.clear()
    x1 = ref[0]["x"]
    flat = lambda h: [k for step in h for k in step]
    for r in dev:
        assert np.array_equal(r["x"], dev[0]["x"])
        assert len(flat(r["its"])) == len(flat(ref[0]["its"])), (
            r["its"], ref[0]["its"])
        assert all(abs(a - b) <= 2 for a, b in
                   zip(flat(r["its"]), flat(ref[0]["its"]))), (
            r["its"], ref[0]["its"])
        assert np.abs(r["x"] - x1).max() <= 1e-5 * np.abs(x1).max()
