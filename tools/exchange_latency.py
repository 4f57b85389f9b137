#!/usr/bin/env python3
"""Latency of one halo exchange per backend, two ranks on ONE GPU.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \\
      --master-addr 127.0.0.1 --master-port 29681 tools/exchange_latency.py

Two processes share the GPU (HIP IPC between them); the operator is a small
banded matrix partitioned by rows, so an SpMV is one exchange plus a ~4 us
kernel.  PCD_COMM_PEER=1: the exchange is one kernel of the engine's stream
that stores into the other process's arena (csrc/pcd_peer.hpp); =0: it goes
through the host transport (device -> host -> gloo -> host -> device), the
stand-in on this box for a host-driven exchange (RCCL refuses two ranks on
one device).  Prints one JSON line on rank 0."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                    # noqa: E402
import torch.distributed as dist                                # noqa: E402
from fenapack_amd import _cabi as c                             # noqa: E402
from fenapack_amd.parallel import TorchHostTransport            # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
# PCD_KIT_ONE_GPU_PER_RANK=1 (tools/first_contact.py on a box with several
# GPUs): every rank drives its own device and the bootstrap backend is RCCL -
# PCD_COMM_PEER=0 then measures the RCCL exchange, =1 the peer-write kernel
# over xGMI.  Default: both ranks on device 0, host transport as bootstrap.
REAL = os.environ.get("PCD_KIT_ONE_GPU_PER_RANK") == "1"
DEV = int(os.environ.get("LOCAL_RANK", "0")) if REAL else 0
torch.cuda.set_device(DEV)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
if REAL:
    dist.init_process_group("nccl", device_id=torch.device("cuda", DEV))
    from fenapack_amd.parallel import Comm                      # noqa: E402
    WORLD_COMM = Comm.world()
else:
    dist.init_process_group("gloo")
out = {}
for n in (4096, 262144):
    A = sp.diags([1.0, -2.0, 5.0, -2.0, 1.0], [-40, -1, 0, 1, 40],
                 shape=(n, n), format="csr")
    for peer in ("1", "0"):
        os.environ["PCD_COMM_PEER"] = peer
        e = c.Engine(c.hip_library(), "BRM1", DEV)
        if REAL:
            e.comm_init(rank, world, WORLD_COMM.unique_id())
        else:
            e.comm_init_host(rank, world, TorchHostTransport())
        r0, r1 = e.row_range(n)
        e.set_csr_local(c.MAT_KP, sp.csr_matrix(A[r0:r1]), A.shape)
        x = torch.randn(r1 - r0, dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        reps = 2000 if peer == "1" else 200
        for _ in range(20):
            e.spmv(c.MAT_KP, x, y, c.MEM_DEVICE)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.spmv(c.MAT_KP, x, y, c.MEM_DEVICE)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["n%d_%s" % (n, "peer_write" if peer == "1" else
                        ("rccl" if REAL else "host_transport"))] = \
            {"us_per_spmv_with_exchange": round(1e6 * dt, 2),
             "halo_doubles_per_rank": 40 + 0,
             "peer_calls": e.info(c.INFO_PEER_CALLS),
             "boot_calls": e.info(c.INFO_BOOT_CALLS)}
        dist.barrier()
        e.destroy()
    # the same kernel without any exchange: one engine, the rank's row block
    if rank == 0:
        e = c.Engine(c.hip_library(), "BRM1", DEV)
        B = sp.csr_matrix(A[:n // 2, :n // 2])
        e.set_csr(c.MAT_KP, B)
        x = torch.randn(n // 2, dtype=torch.float64, device="cuda")
        y = torch.empty_like(x)
        for _ in range(20):
            e.spmv(c.MAT_KP, x, y, c.MEM_DEVICE)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            e.spmv(c.MAT_KP, x, y, c.MEM_DEVICE)
        torch.cuda.synchronize()
        out["n%d_kernel_alone" % n] = {"us_per_spmv": round(
            1e6 * (time.perf_counter() - t0) / 2000, 2)}
        e.destroy()
    dist.barrier()
if rank == 0:
    out["what"] = ("one process per GPU (RCCL bootstrap); SpMV of a 5-band "
                   "matrix partitioned by rows" if REAL else
                   "two processes on ONE MI355X; SpMV of a 5-band matrix "
                   "partitioned by rows; both processes time-share the GPU, so "
                   "the peer-write figure is an upper bound of what two GPUs see")
    print(json.dumps(out))
dist.destroy_process_group()
