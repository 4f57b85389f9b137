"""GPU suite, part 7: REAL ranks - one process per GPU over RCCL.  Skipped on
a box with fewer than two GPUs (RCCL refuses two ranks on one device; there the
same code path runs on thread ranks: test_multi_gpu_threads.py).  Counterpart
of the reference's ``mpirun -np 3`` pass over its demos
(test/regression/test.py:186-190)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from fenapack_amd import _cabi as c
from helpers import flow_state, configure_engine, relerr, set_iter_cfg, free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
WORKER = os.path.join(ROOT, "tests", "two_rank_worker.py")


def _gpus():
    import torch
    return torch.cuda.device_count()      # (does not initialise the GPU)


pytestmark = pytest.mark.gpu
needs2 = pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _launch(args, port, timeout, nproc=2, extra_env=None):
    env = _env()
    env.update(extra_env or {})
    return subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
         "--master-port", str(port)] + args, env=env, capture_output=True,
        text=True, timeout=timeout)


@pytest.mark.parametrize("local", [False, True])
@pytest.mark.timeout(600)
def test_worker_on_one_rank_over_a_real_rccl_communicator(hip_lib, tmp_path,
                                                          local):
    """The worker of the two-GPU tests below, started by the same launcher
    with ONE rank and the communicator forced on (PCD_FORCE_COMM): everything
    of the multi-rank path except a second GPU - process group, unique id,
    ncclCommInitRank, the set-up handshake, halos and all-reduces over RCCL -
    runs on every box, so the script cannot be broken when two GPUs show up.
    ``local``: the rank-local hand-over (pcd_set_csr_local /
    pcd_set_system_local)."""
    out = str(tmp_path / "one.npz")
    run = _launch([WORKER, "--out", out]
                  + (["--handover=rows"] if local else []),
                  free_port(), 420, nproc=1,
                  extra_env={"PCD_FORCE_COMM": "1"})
    assert run.returncode == 0, run.stderr[-3000:]
    one = np.load(out)
    assert one["ranks"] == 1
    st = flow_state("lshape", 3, dt=0.2)
    V = st["V"]
    e = c.Engine(hip_lib, "RBRM1", 0)
    configure_engine(e, st)
    set_iter_cfg(e)
    e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
    e.setup()
    rng = np.random.default_rng(20)
    xp, xs = rng.standard_normal(V.n_p), rng.standard_normal(V.ndof)
    assert relerr(one["A"], e.spmv_np(c.MAT_A, xs, V.ndof)) < 1e-13
    assert relerr(one["fs"], e.fieldsplit_apply_np(xs)) < 1e-11
    for s_ in (c.KSP_AP, c.KSP_MP, c.KSP_RP):
        e.set_inner(s_, "cg", "jacobi", 3000, 1e-10)
    e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 40, 0.0, 0.02, 2.2)
    x, its, _ = e.gmres_np(rng.standard_normal(V.ndof), rtol=1e-6,
                           restart=150, max_it=600)
    assert int(one["gmres_its"]) == its and its > 0
    assert relerr(one["gmres_x"], x) < 1e-7


@needs2
@pytest.mark.timeout(900)
def test_bench_on_two_gpus_over_rccl():
    """``bench.py --gpus 2`` as invoked (self-launched ranks, real RCCL): one
    JSON line, the same outer GMRES counts as one GPU."""
    recs = {}
    for n in (1, 2):
        out = subprocess.run(
            [sys.executable, BENCH, "--gpus", str(n), "--level", "4",
             "--steps", "3", "--warmup", "1", "--no-producer"]
            + (["--no-cpu-baseline"] if n == 1 else []),
            env=_env(), capture_output=True, text=True, timeout=420)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout
        recs[n] = json.loads(lines[0])
    assert recs[2]["n_gpus"] == 2 and recs[2]["value"] > 0
    assert recs[2]["config"]["parallelism"] == "row partition x2"
    assert recs[2]["gmres_its_per_newton_step"] == \
        recs[1]["gmres_its_per_newton_step"]
    # the line's own parity check (one PCApply through both GPUs against the
    # oracle on rank 0) and what the ranks exchanged per PCApply
    assert recs[2]["parity_with_ranks"]["hip_ranks_vs_oracle_rel_err"] < 1e-11, \
        recs[2]["parity_with_ranks"]
    cm = recs[2]["comm"]
    assert "error" not in cm, cm
    assert cm["peer_write_exchanges_and_reductions_per_pcapply"] \
        + cm["rccl_or_host_transport_calls_per_pcapply"] >= 4


@needs2
@pytest.mark.parametrize("local", [False, True])
@pytest.mark.timeout(900)
def test_two_processes_match_one_gpu(hip_lib, tmp_path, local):
    """Partitioned SpMV, PCD apply, fieldsplit PCApply and a full GMRES solve
    on two RCCL ranks against one engine: 1e-11, identical iteration count."""
    out = str(tmp_path / "two.npz")
    run = _launch([WORKER, "--out", out]
                  + (["--handover=rows"] if local else []),
                  free_port(), 600)
    assert run.returncode == 0, run.stderr[-3000:]
    two = np.load(out)
    assert two["ranks"] == 2 and 0 < two["nu_loc"] < 2 * 10 ** 9
    st = flow_state("lshape", 3, dt=0.2)
    V = st["V"]
    assert two["nu_loc"] < V.n_u
    e = c.Engine(hip_lib, "RBRM1", 0)
    configure_engine(e, st)
    set_iter_cfg(e)
    e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
    e.setup()
    rng = np.random.default_rng(20)
    xp, xs = rng.standard_normal(V.n_p), rng.standard_normal(V.ndof)
    assert relerr(two["Kp"], e.spmv_np(c.MAT_KP, xp, V.n_p)) < 1e-13
    assert relerr(two["A"], e.spmv_np(c.MAT_A, xs, V.ndof)) < 1e-13
    assert relerr(two["pcd"], e.apply_np(xp)) < 1e-11
    assert relerr(two["fs"], e.fieldsplit_apply_np(xs)) < 1e-11
    for s in (c.KSP_AP, c.KSP_MP, c.KSP_RP):
        e.set_inner(s, "cg", "jacobi", 3000, 1e-10)
    e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 40, 0.0, 0.02, 2.2)
    x, its, _ = e.gmres_np(rng.standard_normal(V.ndof), rtol=1e-6,
                           restart=150, max_it=600)
    assert int(two["gmres_its"]) == its and its > 0
    assert relerr(two["gmres_x"], x) < 1e-7


@needs2
@pytest.mark.timeout(600)
def test_failing_rank_does_not_hang_the_job(tmp_path):
    """One rank dies after set-up while the other waits inside an RCCL
    collective: the launcher must end the job with a non-zero code."""
    run = _launch([WORKER, "--out", str(tmp_path / "x.npz"), "--fail-rank",
                   "1"], free_port(), 420)
    assert run.returncode != 0
    assert not os.path.exists(str(tmp_path / "x.npz"))
