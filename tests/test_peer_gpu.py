"""GPU suite, part 10: the one-shot peer-write protocol (csrc/pcd_peer.hpp) -
halo exchange and dot products as single kernels that store into the
neighbour's arena, capturable into a hipGraph.

* thread ranks, one stream each (PCD_COMM_PEER=1): the same solve as over the
  host-barrier thread backend, and hipGraph replay with ranks;
* TWO PROCESSES ON ONE GPU over HIP IPC, bootstrapped by a host transport
  (torch.distributed / gloo, pcd_comm_init_host): RCCL cannot build a
  communicator there - the reference's ``mpirun -np 2`` on one box
  (test/regression/test.py:186-190) - against one engine."""
import ctypes
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

from fenapack_amd import PETScOptions
from fenapack_amd.driver import multigrid_inner_options, solve_steady
from fenapack_amd.fem import Cavity, Cavity3D
from fenapack_amd.fem import partition as pt
from fenapack_amd.parallel import Comm
from helpers import free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "peer_ipc_worker.py")


def on_thread_ranks(R, body, own_streams):
    group = ctypes.c_void_p()
    hosts = pt.ThreadHostComm.group(R)
    res, errs = [None] * R, []

    def run(r):
        try:
            comm = Comm(r, R, thread_group=group)
            comm.host = hosts[r]
            if own_streams:
                s = torch.cuda.Stream()
                comm.stream = s.cuda_stream
                with torch.cuda.stream(s):
                    res[r] = body(r, comm, hosts[r])
                    s.synchronize()
            else:
                res[r] = body(r, comm, hosts[r])
        except Exception as ex:                       # pragma: no cover
            import traceback
            errs.append((r, repr(ex), traceback.format_exc()))
            try:
                hosts[r]._sh.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in th), "ranks deadlocked"
    assert not errs, errs
    return res


# Opt-in (FENAPACK_AMD_THREAD_PEER_TEST=1) - the protocol's place in the suite is
# taken by the two-process tests below.  Thread ranks: R = 2 only.  A kernel that waits for another rank's kernel
# needs both in flight at once; the streams of ONE process share a few hardware
# queues (GPU_MAX_HW_QUEUES, default 4), and with three rank streams two of
# them can land in one queue - measured: R = 3 timed out.  More ranks on one
# GPU are separate PROCESSES below, each with its own queues.)
@pytest.mark.skipif(os.environ.get("FENAPACK_AMD_THREAD_PEER_TEST") != "1",
                    reason="opt-in: kernels of two rank threads that wait for "
                           "each other need a hardware queue each; in a long "
                           "test session the streams of one process do not "
                           "always get one (seen: one abort in four runs)")
@pytest.mark.parametrize("cls,kw,dim,R", [
    (Cavity, dict(level=4, nu=0.01), 2, 2),
    (Cavity3D, dict(level=2, nu=0.01, n0=4), 3, 2)])
def test_peer_protocol_on_thread_ranks(hip_lib, monkeypatch, request, cls, kw, dim, R):
    if os.environ.get("GPU_MAX_HW_QUEUES") != "8":
        # the HIP runtime reads the queue count when it initialises: this test
        # alone runs with eight hardware queues, in a process of its own - the
        # rest of the suite keeps the default the product runs with
        env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
        run = subprocess.run(
            [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
             "-p", "no:cacheprovider", request.node.nodeid],
            cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert run.returncode == 0, run.stdout[-4000:] + run.stderr[-2000:]
        return
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "1500")
    monkeypatch.setenv("PCD_PEER_TIMEOUT_S", "10")
    monkeypatch.setenv("PCD_THREAD_BARRIER_TIMEOUT_S", "30")
    PETScOptions.clear()
    multigrid_inner_options(dim=dim, galerkin_u=False)

    def body(r, comm, host):
        from fenapack_amd import _cabi as c
        pp = pt.partitioned(cls, r, R, host=host, **kw)
        out = solve_steady(pp, max_newton=3, newton_rtol=0.0, comm=comm)
        eng = out["solver"].linear_solver().ksp().engine
        V = pp.space
        xg = np.random.default_rng(5).standard_normal(V.ndof)
        y0 = eng.fieldsplit_apply_np(xg)
        eng.graph_enable(True)
        ys = [eng.fieldsplit_apply_np(xg) for _ in range(4)]
        eng.graph_enable(False)
        res = {"its": out["krylov_per_step"], "x": out["w"].vector().copy(),
               "y0": y0, "ys": ys}
        eng.synchronize()
        host.allgather(0)                 # every rank is done with the others
        eng.destroy()                     # streams / queues back before the next test
        return res

    monkeypatch.setenv("PCD_COMM_PEER", "0")
    ref = on_thread_ranks(R, body, own_streams=False)[0]
    monkeypatch.setenv("PCD_COMM_PEER", "1")
    runs = on_thread_ranks(R, body, own_streams=True)
    PETScOptions.clear()
    for r in runs:
        assert r["its"] == ref["its"], (r["its"], ref["its"])
        # (the peer all-reduce adds in rank order, the thread backend too)
        assert np.abs(r["x"] - ref["x"]).max() <= 1e-11 * np.abs(ref["x"]).max()
        assert np.abs(r["y0"] - ref["y0"]).max() <= 1e-12 * np.abs(ref["y0"]).max()
        # hipGraph replay with ranks: bitwise the eager result
        for y in r["ys"]:
            assert np.array_equal(y, r["y0"])
    assert all(np.array_equal(r["x"], runs[0]["x"]) for r in runs)


def _launch(args, port, timeout, nproc=2, extra_env=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(extra_env or {})
    return subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
         "--master-port", str(port)] + args, env=env, capture_output=True,
        text=True, timeout=timeout)


@pytest.mark.timeout(400)
def test_two_processes_on_one_gpu_over_hip_ipc(hip_lib, tmp_path):
    out = str(tmp_path / "two.npz")
    run = _launch([WORKER, "--out", out], free_port(), 300,
                  extra_env={"PCD_COMM_PEER": "1", "PCD_PEER_TIMEOUT_S": "15",
                             "PCD_COMM_VERBOSE": "1"})
    assert run.returncode == 0, run.stderr[-4000:]
    assert "peer protocol unavailable" not in run.stderr, run.stderr[-2000:]
    two = np.load(out)
    assert two["ranks"] == 2
    # one engine, same options
    os.environ["PCD_REPLICATE_BELOW"] = "1500"
    try:
        PETScOptions.clear()
        multigrid_inner_options(dim=2, galerkin_u=False)
        pb = Cavity(4, nu=0.01)
        one = solve_steady(pb, max_newton=3, newton_rtol=0.0)
        eng = one["solver"].linear_solver().ksp().engine
        xg = np.random.default_rng(5).standard_normal(pb.space.ndof)
        y = eng.fieldsplit_apply_np(xg)
    finally:
        os.environ.pop("PCD_REPLICATE_BELOW", None)
        PETScOptions.clear()
    assert list(two["its"]) == one["krylov_per_step"]
    x1 = one["w"].vector()
    assert np.abs(two["x"] - x1).max() <= 1e-9 * np.abs(x1).max()
    assert np.abs(two["y0"] - y).max() <= 1e-11 * np.abs(y).max()
    for k in ("y1", "y2", "y3"):                  # eager / captured / replayed
        assert np.array_equal(two[k], two["y0"]), k
    # launches: the exchanges are ONE kernel each on top of the single-GPU
    # count; a replayed graph issues none
    from fenapack_amd import _cabi as c
    from fenapack_amd.petsc import Vec
    xd = Vec(xg.copy(), device="cuda:0")
    yd = xd.duplicate()
    eng.fieldsplit_apply(xd.t, yd.t, c.MEM_DEVICE)
    l0 = eng.info(c.INFO_LAUNCHES)
    for _ in range(10):
        eng.fieldsplit_apply(xd.t, yd.t, c.MEM_DEVICE)
    single = (eng.info(c.INFO_LAUNCHES) - l0) / 10
    assert two["launches_per_pcapply_graph"] <= 2.0      # (copy in, copy out)
    print("two processes on one GPU over HIP IPC: %.0f launches per PCApply, "
          "%.0f of them exchanges / reductions (one engine: %.0f launches), "
          "%.3f ms eager, %.3f ms as a replayed hipGraph"
          % (two["launches_per_pcapply"], two["exchanges_per_pcapply"], single,
             1e3 * two["td_eager"], 1e3 * two["td_graph"]))
    # every exchange / reduction is ONE kernel, nothing goes through the host
    # transport; (with PCD_REPLICATE_BELOW lowered as here, levels that one
    # engine composes into single operators stay step by step on ranks - the
    # launch count at the default limit is in profiles/)
    assert two["boot_calls_per_pcapply"] == 0.0
    assert two["exchanges_per_pcapply"] >= 4


@pytest.mark.timeout(600)
@pytest.mark.parametrize("nproc", [2, 3])
def test_interior_boundary_overlap_is_bitwise_the_fused_exchange(hip_lib, tmp_path,
                                                                nproc):
    """PCD_OVERLAP=1 ("SpMV overlaps the offd halo transfer with
    the diag compute"): the tile kernels' blocks that read no ghost column run
    between a send kernel and a wait-and-land kernel, the boundary blocks after
    it - every block is computed as without the split, so solve and PCApply
    are BITWISE the fused exchange's; two launches more per overlapped SpMV.
    (PCD_VEC_TILE=2: tile kernels on every F (x) I operator, also at this
    size.)  What it buys can only be priced with real peers; the switch is
    there for the first multi-GPU run to A/B."""
    res = {}
    for ov in ("0", "1"):
        out = str(tmp_path / ("ov%s.npz" % ov))
        run = _launch([WORKER, "--out", out], free_port(), 300, nproc=nproc,
                      extra_env={"PCD_COMM_PEER": "1", "PCD_PEER_TIMEOUT_S": "15",
                                 "PCD_COMM_VERBOSE": "1", "PCD_VEC_TILE": "2",
                                 "PCD_OVERLAP": ov})
        assert run.returncode == 0, run.stderr[-4000:]
        assert "peer protocol unavailable" not in run.stderr, run.stderr[-2000:]
        res[ov] = np.load(out)
    a, b = res["0"], res["1"]
    assert a["ranks"] == nproc and b["ranks"] == nproc
    assert list(a["its"]) == list(b["its"])
    assert np.array_equal(a["x"], b["x"])
    for k in ("y0", "y1", "y2", "y3"):            # eager / captured / replayed
        assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(b[k], b["y0"]), k
    assert b["boot_calls_per_pcapply"] == 0.0
    # the split costs launches: a send and a wait instead of one exchange, two
    # consumer launches instead of one
    assert b["launches_per_pcapply"] > a["launches_per_pcapply"]
    assert b["launches_per_pcapply_graph"] <= 2.0
    print("%d processes on one GPU: %.0f launches per PCApply fused, %.0f with "
          "the interior / boundary split; %.3f / %.3f ms eager, %.3f / %.3f ms "
          "replayed" % (nproc, a["launches_per_pcapply"], b["launches_per_pcapply"],
                        1e3 * a["td_eager"], 1e3 * b["td_eager"],
                        1e3 * a["td_graph"], 1e3 * b["td_graph"]))


@pytest.mark.timeout(300)
def test_a_process_that_leaves_does_not_hang_the_other(hip_lib, tmp_path):
    """Rank 1 exits after set-up: rank 0's exchange kernels give up after
    PCD_PEER_TIMEOUT_S, the engine reports PCD_ERR_COMM, the launcher ends
    with a non-zero status - nothing hangs, the GPU stays usable."""
    out = str(tmp_path / "gone.npz")
    run = _launch([WORKER, "--out", out, "--fail-rank", "1"], free_port(), 240,
                  extra_env={"PCD_COMM_PEER": "1", "PCD_PEER_TIMEOUT_S": "3"})
    assert run.returncode != 0
    assert not os.path.exists(out)
    # the GPU still works
    t = torch.ones(1024, device="cuda")
    assert float(t.sum().item()) == 1024.0


@pytest.mark.timeout(400)
def test_bench_with_two_and_four_ranks_sharing_the_gpu(hip_lib):
    """bench.py --gpus 2 (and 4) through its own launcher, the ranks as
    processes on GPU 0 (--share-gpu: gloo bootstraps the peer protocol over HIP IPC):
    the contract's multi-rank path - launcher, barrier-bracketed timed region,
    max over ranks, ONE JSON line, hipGraph replay with ranks - on a box with
    one GPU.  (Real GPUs: test_two_gpus.py.)"""
    import json
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    # (the last form is config 5's start-up in small: the cube through
    # -pc_type gamg with every rank assembling its slab and the hierarchy
    # aggregated rank by rank - amg.PartitionedSA over torch.distributed)
    cube = ["--geometry", "cube", "--level", "1", "--n0", "6", "--algebraic",
            "--partitioned-producer"]
    for n, extra in ((2, []), (2, ["--partitioned-producer"]), (4, []),
                     (2, cube)):
        # (the plain two-rank form also runs the line's own parity check:
        # one PCApply through the ranks against the oracle on rank 0)
        check = n == 2 and not extra
        run = subprocess.run(
            [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n),
             "--share-gpu", "--steps", "10", "--warmup", "3"]
            + ([] if check else ["--no-cpu-baseline"])
            + (["--level", "4"] if "--geometry" not in extra else []) + extra,
            env=env, capture_output=True, text=True, timeout=360)
        assert run.returncode == 0, run.stderr[-3000:]
        lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, run.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["steps"] == 10 and d["value"] > 0
        assert d["config"]["parallelism"] == "row partition x%d" % n
        assert d["config"]["launch"] == "hipGraph replay"
        assert ("partitioned" in d["config"]["producer"]) == bool(extra)
        if check:
            assert d["parity_with_ranks"]["hip_ranks_vs_oracle_rel_err"] < 1e-11, \
                d["parity_with_ranks"]
        else:
            assert "parity_with_ranks" not in d
        if "--algebraic" in extra:
            assert "cube N=12" in d["config"]["workload"], d["config"]
        assert len(d["gmres_its_per_newton_step"]) == 2
        # what the ranks exchange per PCApply: every halo exchange / reduction
        # a one-shot peer-write kernel, none through RCCL / the host transport
        cm = d["comm"]
        assert "error" not in cm, cm
        assert cm["peer_write_exchanges_and_reductions_per_pcapply"] >= 4
        assert cm["rccl_or_host_transport_calls_per_pcapply"] == 0.0
        assert cm["halo_channels_declined_by_the_peer_arena"] == 0
        assert cm["launches_per_pcapply"] > \
            cm["peer_write_exchanges_and_reductions_per_pcapply"]
