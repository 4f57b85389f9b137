"""GPU suite, part 8: BASELINE.json's multi-rank configurations at their own
shape, with the ranks as threads on the ONE GPU of this box (the code path of
a ``torch.distributed.run --nproc-per-node R`` start minus RCCL itself, which
refuses two ranks on one device; real ranks: test_two_gpus.py).

* config 4: unsteady demo, L-shape level 4, dt 0.2, 100 time steps, device
  producer - on 1, 2 and 4 ranks: identical Picard counts, Krylov totals that
  agree to a handful of iterations (round-off of the partitioned reductions),
  replicas identical.  Round 2's one-off run (tools/unsteady_thread_ranks.py,
  profiles/r02_ae_*): 13339 / 13340 / 13340 Krylov iterations, 421 Picard
  iterations.
* config 5's shape: 3-D lid-driven cube, three components per node, the two
  finest levels partitioned and the rest replicated, on 8 ranks (N = 32 per
  side; the full 10 M DOF are parity-checked on one GPU by
  tools/parity_large.py: profiles/r03_*parity_cube*.json).
"""
import ctypes
import os
import threading

import pytest

from fenapack_amd import PETScOptions
from fenapack_amd.device_producer import (solve_steady_device,
                                          solve_unsteady_device)
from fenapack_amd.driver import multigrid_inner_options
from fenapack_amd.fem import BackwardStep, Cavity3D
from fenapack_amd.parallel import Comm

pytestmark = pytest.mark.gpu


def _on_ranks(R, solve):
    """``solve(comm)`` on R thread ranks (R = 1: plain call)."""
    if R == 1:
        return [solve(None)]
    group = ctypes.c_void_p()
    res, errs = [None] * R, []

    def body(r):
        try:
            res[r] = solve(Comm(r, R, thread_group=group))
        except Exception as ex:            # pragma: no cover
            errs.append((r, repr(ex)))

    th = [threading.Thread(target=body, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    assert not any(t.is_alive() for t in th), "ranks deadlocked"
    assert not errs, errs
    return res


@pytest.fixture
def replicate_below(monkeypatch):
    def set_limit(rows):
        monkeypatch.setenv("PCD_REPLICATE_BELOW", str(rows))
    return set_limit


@pytest.mark.heavy(4)
@pytest.mark.timeout(900)
def test_config4_unsteady_100_steps_on_1_2_4_ranks(hip_lib, replicate_below):
    """100 steps on one rank (round 2's totals), the first 25 of them (the
    length of the reference's own table) on two and on four ranks:
    R thread ranks share ONE GPU, so a run costs R times the replicated device
    work, and the suite has a time budget.  The full 100 steps on 2 and 4
    ranks are a tools/ run, repeated every round
    (profiles/r03_i_unsteady_level4_100steps_thread_ranks.jsonl: 13339 /
    13340 / 13340 Krylov iterations on 1 / 2 / 4 ranks)."""
    # partitioned finest levels, replicated coarse ones - as on real ranks; the
    # default limit (60000 rows) would replicate everything at this size
    replicate_below(2000)
    dt = 0.2
    PETScOptions.clear()
    multigrid_inner_options(cycles_u=2, cycles_p=2)

    def solver(steps):
        def solve(comm):
            pb = BackwardStep(4, nu=0.02, dt=dt, dirichlet_diag="multiplicity")
            out = solve_unsteady_device(pb, dt=dt, t_end=steps * dt,
                                        newton_rtol=1e-5, gmres_rtol=1e-6,
                                        comm=comm)
            return {"ndof": out["ndof"], "steps": out["steps"],
                    "krylov": out["krylov_its"], "picard": out["newton_its"],
                    "per_step": out["krylov_per_step"],
                    "picard_per_step": [len(k) for k in
                                        out["krylov_per_newton"]],
                    "checksum": float(abs(out["w"].vector()).sum())}
        return solve

    one = _on_ranks(1, solver(100))[0]
    assert one["ndof"] == 25987 and one["steps"] == 100
    # round 2's totals (13339 Krylov / 421 Picard iterations); the producer's
    # element matrices changed by round-off since: a small band
    assert abs(one["krylov"] - 13339) <= 70 and abs(one["picard"] - 421) <= 2
    for R, steps in ((2, 25), (4, 25)):
        runs = _on_ranks(R, solver(steps))
        ref_k = sum(one["per_step"][:steps])
        ref_p = sum(one["picard_per_step"][:steps])
        for r in runs:                          # every replica
            assert r["steps"] == steps and r["picard"] == ref_p
            assert abs(r["krylov"] - ref_k) <= 8, (R, r["krylov"], ref_k)
        assert len({r["krylov"] for r in runs}) == 1
        assert len({round(r["checksum"], 6) for r in runs}) == 1
    PETScOptions.clear()


@pytest.mark.heavy(3)
@pytest.mark.rss_gb(26)
@pytest.mark.timeout(900)
def test_config5_shape_cube_n32_on_8_ranks(hip_lib, replicate_below):
    replicate_below(20000)
    PETScOptions.clear()
    multigrid_inner_options(dim=3)

    # one problem object for all ranks (read-only here; its pattern caches are
    # warm after the one-rank run): the R threads would otherwise build it R
    # times under the interpreter lock
    pb = Cavity3D(3, nu=0.01, n0=4)                      # N = 32: 859 812 DOF

    def solve(comm):
        from fenapack_amd import _cabi as c
        out = solve_steady_device(pb, max_newton=10, comm=comm)
        eng = out["solver"].linear_solver().ksp().engine
        return {"ndof": pb.space.ndof, "its": out["krylov_per_step"],
                "rows_u": int(eng.info(c.INFO_N_U_LOCAL)),
                "checksum": float(abs(out["w"].vector()).sum())}

    one = _on_ranks(1, solve)[0]
    eight = _on_ranks(8, solve)
    assert one["ndof"] == 859812
    assert one["its"][0] <= 12 and one["its"][1] <= 50
    for r in eight:
        assert r["its"] == one["its"]
        assert abs(r["checksum"] - one["checksum"]) <= 1e-9 * one["checksum"]
    rows = [r["rows_u"] for r in eight]
    assert sum(rows) == one["rows_u"] and max(rows) - min(rows) <= 3
    PETScOptions.clear()


@pytest.mark.parametrize("R", [2, 3])
def test_host_driven_solve_with_rank_local_handover(hip_lib, replicate_below,
                                                    monkeypatch, R):
    """FENAPACK_AMD_LOCAL_HANDOVER=1: every operator, the block system and
    every partitioned multigrid level cross the C ABI as this rank's rows
    only (pcd_set_csr_local / pcd_set_system_local / pcd_mg_set_level_local),
    through the whole Python stack, value refreshes between the nonlinear
    steps included.  Same Krylov history and the same solution as the global
    hand-over on the same ranks."""
    import numpy as np
    from fenapack_amd.driver import solve_steady
    from fenapack_amd.fem import Cavity
    replicate_below(700)        # finest levels partitioned, the rest replicated

    def run(local):
        if local:
            monkeypatch.setenv("FENAPACK_AMD_LOCAL_HANDOVER", "1")
        else:
            monkeypatch.delenv("FENAPACK_AMD_LOCAL_HANDOVER", raising=False)
        PETScOptions.clear()
        multigrid_inner_options()
        pb = Cavity(3, nu=0.01)

        def solve(comm):
            out = solve_steady(pb, gmres_rtol=1e-6, max_newton=4, comm=comm)
            eng = out["solver"].linear_solver().ksp().engine
            assert eng.local_handover == local
            return out["krylov_per_step"], np.array(out["w"].vector())
        res = _on_ranks(R, solve)
        PETScOptions.clear()
        return res

    glob, loc = run(False), run(True)
    for (kg, xg), (kl, xl) in zip(glob, loc):
        assert kl == kg and len(kg) == 4
        assert np.abs(xl - xg).max() <= 1e-12 * np.abs(xg).max()
