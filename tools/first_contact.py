#!/usr/bin/env python3
"""First-contact kit for the first box with more than one GPU.

Nothing of DESIGN.md 6 has run on two or more GPUs (RCCL with N > 1 ranks,
cross-device HIP IPC, a scaling curve): this pool's boxes have one.  The
reference runs every demo under ``mpirun -np 3``
(test/regression/test.py:186-195); the day a multi-GPU box shows up, this is
the one command to run on it:

    python3 tools/first_contact.py [--out gpurun_out/first_contact] [--share-gpu]

Every STAGE runs in a fresh child process (a dead stage - a hang cut by its
time-out, a crash, a refused peer mapping - exits non-zero, is recorded, and
the kit goes on to the next one) and writes ONE JSON file:

  00_devices        GPUs, peer-access matrix, link types as the runtime reports
                    them, HSA_ENABLE_IPC_MODE_LEGACY
  01_arena          two processes map each other's uncached arena (HIP IPC) and
                    exchange through it: tests/test_peer_gpu.py's worker
  02_exchange       latency of one SpMV + halo exchange: peer-write kernel vs
                    PCD_COMM_PEER=0 (RCCL with real GPUs; the host transport
                    under --share-gpu)
  03_two_gpus       tests/test_two_gpus.py (skips itself below two GPUs)
  04_bench_l6_N     bench.py --gpus N at the headline size, N = 2, 4, 8
  05_bench_n73_N_ov bench.py --gpus N on config 5's own mesh (cube N = 73,
                    rank-local producer, algebraic hierarchy), PCD_OVERLAP=0/1

``--share-gpu``: every multi-rank stage puts its ranks on ONE device (HIP IPC
between processes) - the rehearsal this pool allows; RCCL refuses two ranks on
one device, so stages that need it record that and the kit goes on.
``--quick``: small sizes (level 4, cube N = 16), for the suite.
A summary (stage, status, seconds, headline number) is printed at the end and
written to ``summary.json``."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (the resident-set watchdog of this repository's scripts; the kit itself only
# starts child processes and never touches the GPU)
from fenapack_amd import _guard                                      # noqa: E402
_guard.start_rss_watchdog()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def base_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.setdefault("PCD_PEER_TIMEOUT_S", "60")
    return env


DEVICES = r"""
import json, os, sys
import torch
n = torch.cuda.device_count()
out = {"gpus": n, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
       "names": [torch.cuda.get_device_name(i) for i in range(n)],
       "peer_access": [[bool(i == j or torch.cuda.can_device_access_peer(i, j))
                        for j in range(n)] for i in range(n)]}
try:
    import subprocess
    out["rocm_smi_topology"] = subprocess.run(
        ["rocm-smi", "--showtopo"], capture_output=True, text=True,
        timeout=60).stdout[-4000:]
except Exception as ex:
    out["rocm_smi_topology"] = "unavailable: %r" % (ex,)
print(json.dumps(out))
"""


def run_stage(name, cmd, out_dir, timeout, env=None, json_from="stdout"):
    """One stage in a child process; its JSON (last line starting with '{' of
    stdout, or what the command wrote to ``json_from``) goes to NAME.json."""
    t0 = time.time()
    rec = {"stage": name, "command": " ".join(cmd)}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout,
                           env=env or base_env(), cwd=ROOT)
        rec["returncode"] = r.returncode
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if lines:
            try:
                rec["result"] = json.loads(lines[-1])
            except ValueError:
                rec["result_raw"] = lines[-1][:2000]
        rec["stdout_tail"] = r.stdout[-1500:]
        rec["stderr_tail"] = r.stderr[-2500:]
        rec["status"] = "ok" if r.returncode == 0 else "failed"
    except subprocess.TimeoutExpired as ex:
        rec["status"] = "timeout"
        rec["returncode"] = None
        rec["stderr_tail"] = (ex.stderr or b"")[-2500:].decode("utf-8", "replace") \
            if isinstance(ex.stderr, bytes) else str(ex.stderr)[-2500:]
    rec["seconds"] = round(time.time() - t0, 1)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(rec, f, indent=1)
    return rec


def torchrun(nproc, script_and_args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port())] + script_and_args


def headline(rec):
    r = rec.get("result") or {}
    if "value" in r:
        return "%.1f %s, GMRES %s" % (r["value"], r.get("unit", ""),
                                      r.get("gmres_its_per_newton_step"))
    if "gpus" in r:
        return "%d GPU(s)" % r["gpus"]
    for k, v in r.items():
        if isinstance(v, dict) and "us_per_spmv_with_exchange" in v:
            return "%s %.1f us" % (k, v["us_per_spmv_with_exchange"])
    return ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out",
                                                  "first_contact"))
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--stages", default="", help="comma list of stage name "
                    "prefixes to run (default: all)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    want = [s for s in a.stages.split(",") if s]
    on = lambda name: not want or any(name.startswith(w) for w in want)
    ranks = [int(r) for r in a.ranks.split(",") if r]
    recs = []
    share = ["--share-gpu"] if a.share_gpu else []

    if on("00_devices"):
        recs.append(run_stage("00_devices", [sys.executable, "-c", DEVICES],
                              a.out, 300))
    ngpu = (recs[0].get("result") or {}).get("gpus", 1) if recs else 1
    real = ngpu >= 2 and not a.share_gpu

    if on("01_arena"):
        # two processes, each maps the other's arena and exchanges through it
        env = base_env()
        if real:
            env["PCD_KIT_ONE_GPU_PER_RANK"] = "1"
        recs.append(run_stage(
            "01_arena", [sys.executable, "-m", "pytest", "-x", "-q",
                         "tests/test_peer_gpu.py", "-m", "gpu", "-k",
                         "two_processes_on_one_gpu_over_hip_ipc or leaves_does_not_hang"], a.out, 900, env))
    if on("02_exchange"):
        env = base_env()
        if real:
            env["PCD_KIT_ONE_GPU_PER_RANK"] = "1"
        recs.append(run_stage(
            "02_exchange", torchrun(2, ["tools/exchange_latency.py"]), a.out,
            600, env))
    if on("03_two_gpus"):
        recs.append(run_stage(
            "03_two_gpus", [sys.executable, "-m", "pytest", "-q",
                            "tests/test_two_gpus.py", "-m", "gpu", "-rs"],
            a.out, 1800))
    lvl = ["--level", "4"] if a.quick else []
    for n in ranks:
        name = "04_bench_l6_%d" % n
        if on(name) and (a.share_gpu or n <= ngpu):
            recs.append(run_stage(
                name, [sys.executable, "bench.py", "--gpus", str(n), "--steps",
                       "20", "--warmup", "5"] + lvl + share, a.out, 1500))
    cube = ["--geometry", "cube", "--level", "0", "--n0",
            "16" if a.quick else "73", "--algebraic"]
    for n in ranks:
        for ov in ("0", "1"):
            name = "05_bench_n73_%d_overlap%s" % (n, ov)
            if not on(name) or not (a.share_gpu or n <= ngpu):
                continue
            env = base_env()
            env["PCD_OVERLAP"] = ov
            if a.quick:
                env["PCD_REPLICATE_BELOW"] = "1500"
            recs.append(run_stage(
                name, [sys.executable, "bench.py", "--gpus", str(n), "--steps",
                       "10", "--warmup", "3", "--partitioned-producer"] + cube
                + share, a.out, 2400, env))
    summary = [{"stage": r["stage"], "status": r["status"],
                "seconds": r["seconds"], "headline": headline(r)}
               for r in recs]
    with open(os.path.join(a.out, "summary.json"), "w") as f:
        json.dump({"gpus": ngpu, "share_gpu": a.share_gpu, "quick": a.quick,
                   "stages": summary}, f, indent=1)
    for s_ in summary:
        print("%-28s %-8s %7.1f s  %s" % (s_["stage"], s_["status"],
                                          s_["seconds"], s_["headline"]))
    # the kit itself succeeds when it ran to the end; stages speak for themselves
    return 0


if __name__ == "__main__":
    sys.exit(main())
