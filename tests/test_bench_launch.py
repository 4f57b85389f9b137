"""CPU suite: the multi-rank launch path of bench.py.

``python bench.py --gpus N`` must work as invoked, with no external launcher
(the reference's counterpart is ``mpirun -np N`` over the same script:
test/regression/test.py:186-195).  The GPU work itself cannot run here, so the
``--stub-step`` mode replaces it by a no-op and everything around it is real:
child processes started before anything touches a GPU, the gloo rendezvous on
127.0.0.1, barrier + max-over-ranks timing, ONE JSON line from rank 0, failure
propagation."""
import json
import os
import subprocess
import sys

import pytest

from helpers import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(300)
def test_bench_self_launches_two_ranks():
    out = subprocess.run(
        [sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup",
         "1", "--stub-step"], env=_env(), capture_output=True, text=True,
        timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout                  # ONE line, rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert rec["config"]["parallelism"] == "row partition x2"
    assert rec["value"] > 0


@pytest.mark.timeout(300)
def test_bench_under_an_external_launcher():
    # the driver's form: python -m torch.distributed.run --nproc-per-node N
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(free_port()), BENCH, "--gpus", "2", "--steps", "2",
         "--warmup", "0", "--stub-step"], env=_env(), capture_output=True,
        text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(l) for l in out.stdout.splitlines()
            if l.startswith("{")]
    assert len(recs) == 1 and recs[0]["n_gpus"] == 2


@pytest.mark.timeout(120)
def test_world_size_mismatch_and_child_failure_exit_nonzero():
    env = _env()
    env["WORLD_SIZE"] = "3"
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-step"],
                         env=env, capture_output=True, text=True, timeout=100)
    assert out.returncode != 0 and "WORLD_SIZE=3" in out.stderr
    # a failing child (no GPU here, real workload) must fail the launcher
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline"], env=_env(),
                         capture_output=True, text=True, timeout=100)
    import torch
    if torch.cuda.device_count() < 2:
        assert out.returncode != 0
        assert out.stdout.strip() == ""


def test_first_contact_kit_runs_a_stage_and_writes_its_records(tmp_path):
    """tools/first_contact.py (the one command for the first box with several
    GPUs): every stage is a child process with a JSON of its own and a summary
    at the end - here the stage that needs no GPU (the device inventory)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "kit"
    r = subprocess.run([sys.executable, os.path.join(root, "tools",
                                                     "first_contact.py"),
                        "--out", str(out), "--stages", "00_devices"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out / "00_devices.json"))
    assert rec["status"] == "ok" and "gpus" in rec["result"]
    assert "peer_access" in rec["result"]
    summary = json.load(open(out / "summary.json"))
    assert [s["stage"] for s in summary["stages"]] == ["00_devices"]
    assert "00_devices" in r.stdout
