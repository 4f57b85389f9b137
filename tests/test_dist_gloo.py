"""CPU suite, part 4: the N>1 path with two real processes (gloo, world 2).
Each rank asks the engine's own partitioning code (pcd_dist_probe: the C++ of
csrc/pcd_dist.hpp, host only) for its row block and halo plan, exchanges the
halo through torch.distributed and runs a distributed SpMV and a Jacobi-PCG
whose dot products are all-reduced - the communication pattern of the GPU
path - against the single-process numpy restatement."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _halo(dist, plan, x_owned, rank):
    """Fill the ghost segment: send my packed entries, receive the peers'."""
    ghost = np.zeros(plan["nghost"])
    reqs, bufs = [], []
    import torch
    for peer, idx in sorted(plan["send"].items()):
        t = torch.from_numpy(np.ascontiguousarray(x_owned[idx]))
        reqs.append(dist.isend(t, peer))
        bufs.append(t)
    recvs = []
    for peer, (b, e) in sorted(plan["recv"].items()):
        t = torch.zeros(e - b, dtype=torch.float64)
        reqs.append(dist.irecv(t, peer))
        recvs.append((b, e, t))
    for r in reqs:
        r.wait()
    for b, e, t in recvs:
        ghost[b:e] = t.numpy()
    return ghost


def _worker(rank, world, port, queue):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fenapack_amd import _cabi as c
        from helpers import flow_state
        from oracle import reference_numpy as rn
        st = flow_state("lshape", 2)
        pb, L = st["pb"], st["L"]
        rng = np.random.default_rng(0)
        out = {}
        for name, A, even in (("Ap", pb.Ap, False), ("A00", L["A00"], True)):
            plan = c.dist_probe(A, rank, world, even, even)
            x = rng.standard_normal(A.shape[0])
            r0, nl = plan["row0"], plan["local"].shape[0]
            assert plan["col0"] == r0 and plan["ncols_owned"] == nl
            xo = x[r0:r0 + nl]
            y = plan["local"] @ np.concatenate([xo, _halo(dist, plan, xo,
                                                          rank)])
            out[name + "_spmv"] = float(np.abs(y - (A @ x)[r0:r0 + nl]).max())
            out[name + "_rows"] = nl
        # distributed Jacobi-PCG on Ap: 2 all-reduced dots per iteration
        A = pb.Ap
        plan = c.dist_probe(A, rank, world, False, False)
        r0, nl = plan["row0"], plan["local"].shape[0]
        b = rng.standard_normal(A.shape[0])
        dinv = 1.0 / A.diagonal()[r0:r0 + nl]

        def gdot(u, v):
            t = torch.tensor([float(u @ v)], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t[0])
        x = np.zeros(nl)
        r = b[r0:r0 + nl].copy()
        z = dinv * r
        rz = gdot(r, z)
        p = z.copy()
        for it in range(10):
            if it:
                p = z + (rz / rz_old) * p
            q = plan["local"] @ np.concatenate([p, _halo(dist, plan, p, rank)])
            alpha = rz / gdot(p, q)
            x += alpha * p
            r -= alpha * q
            z = dinv * r
            rz_old, rz = rz, gdot(r, z)
        xr, _ = rn.cg(A, b, 10)
        out["cg"] = float(np.abs(x - xr[r0:r0 + nl]).max() / np.abs(xr).max())
        queue.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_partition_halo_and_allreduce_two_processes():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q))
             for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from helpers import flow_state
    st = flow_state("lshape", 2)
    assert res[0]["Ap_rows"] + res[1]["Ap_rows"] == st["V"].n_p
    assert res[0]["A00_rows"] + res[1]["A00_rows"] == st["V"].n_u
    assert res[0]["A00_rows"] % 2 == 0
    for r in (0, 1):
        assert res[r]["Ap_spmv"] < 1e-13 and res[r]["A00_spmv"] < 1e-12
        assert res[r]["cg"] < 1e-12
