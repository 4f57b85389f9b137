"""Shared test utilities: golden loading and engine configuration."""
import glob
import os

import numpy as np
import scipy.sparse as sp

from fenapack_amd import _cabi as c

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VARIANTS = ("BRM1", "BRM2", "RBRM1", "RBRM2")
# must match ITER_CFG of tests/golden/make_goldens.py
ITER_CFG = {c.KSP_AP: ("cg", 8, 0.0, 0.5, 2.0),
            c.KSP_MP: ("chebyshev", 5, 0.0, 0.5, 2.0),
            c.KSP_RP: ("cg", 6, 0.0, 0.5, 2.0)}


def golden_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def csr_from(d, p):
    return sp.csr_matrix((d[p + "_data"], d[p + "_indices"], d[p + "_indptr"]),
                         shape=tuple(d[p + "_shape"]))


def relerr(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def load_pcd_operators(e, d):
    e.set_csr(c.MAT_AP, csr_from(d, "Ap"))
    e.set_csr(c.MAT_MP, csr_from(d, "Mp"))
    e.set_csr(c.MAT_KP, csr_from(d, "Kp"))
    e.set_csr(c.MAT_RP, csr_from(d, "Rp"))
    e.set_bc(d["bc_idx"], d["bc_val"])


def set_iter_cfg(e, cfg=ITER_CFG):
    for slot, (ksp, its, rtol, lo, hi) in cfg.items():
        e.set_inner(slot, ksp, "jacobi", its, rtol, lo, hi)


def set_tight_cg(e, slots=(c.KSP_AP, c.KSP_MP, c.KSP_RP), rtol=1e-14):
    for s in slots:
        e.set_inner(s, "cg", "jacobi", 100000, rtol)


_state_cache = {}


def flow_state(kind, level, variant="BRM1", picard_steps=2, **kw):
    """Problem + matrices frozen at Picard iterate 2."""
    key = (kind, level, variant, picard_steps, tuple(sorted(kw.items())))
    if key in _state_cache:
        return _state_cache[key]
    import scipy.sparse.linalg as spla
    from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D
    cls = {"lshape": BackwardStep, "cavity": Cavity, "cube": Cavity3D}[kind]
    pb = cls(level, variant=variant, **kw)
    V = pb.space
    xu, xp = pb.initial_guess()
    for _ in range(picard_steps):
        L = pb.linearise(xu, xp)
        A = sp.bmat([[L["A00"], L["A01"]], [L["A10"], None]]).tocsc()
        if kind != "lshape":
            A = A + 1e-10 * sp.identity(A.shape[0], format="csc")
        dx = spla.spsolve(A, np.concatenate([L["bu"], L["bp"]]))
        xu, xp = xu - dx[:V.n_u], xp - dx[V.n_u:]
    L = pb.linearise(xu, xp)
    st = {"pb": pb, "V": V, "xu": xu, "xp": xp, "L": L,
          "A": V.monolithic(L["A00"], L["A01"], L["A10"]),
          "b": V.to_mixed(L["bu"], L["bp"]),
          "Kp": pb.Kp(xu), "Rp": pb.Rp() if pb.idt else None}
    _state_cache[key] = st
    return st


def configure_engine(e, st, with_system=True):
    pb = st["pb"]
    e.set_velocity_block(st["V"].dim)
    e.set_csr(c.MAT_AP, pb.Ap)
    e.set_csr(c.MAT_MP, pb.Mp)
    e.set_csr(c.MAT_KP, st["Kp"])
    if st["Rp"] is not None:
        e.set_csr(c.MAT_RP, st["Rp"])
    e.set_bc(pb.bc_p_idx, pb.bc_p_val)
    if with_system:
        e.set_system(st["A"], st["V"].is_u, st["V"].is_p)


def push_multigrid(e, slot, A, chain, nu=2, ratio=0.1, cycles=1):
    """Galerkin hierarchy of ``A`` handed to an engine (HIP or oracle);
    returns (ops, bounds, coarse_inverse) for the numpy restatement."""
    from fenapack_amd.fem.multigrid import galerkin_chain, coarse_inverse
    from fenapack_amd.petsc import estimate_emax
    ops = galerkin_chain(A, chain)
    bounds = [None]
    for l in range(1, len(ops)):
        emax = 1.1 * estimate_emax(ops[l], iters=12)
        bounds.append((ratio * emax, emax))
    Cs = coarse_inverse(ops[0])
    C = Cs.toarray()
    e.mg_begin(slot, len(ops), nu, nu)
    e.mg_set_level(slot, 0, Cs)
    for l in range(1, len(ops)):
        e.mg_set_level(slot, l, ops[l] if l < len(ops) - 1 else None,
                       chain[l], *bounds[l])
    e.set_inner(slot, "richardson", "mg", cycles, 0.0)
    return ops, bounds, C


def free_port():
    """A TCP port nobody listens on right now (rendezvous of the multi-process
    tests: a fixed number can be taken on a shared host)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ---- config 5's own mesh: the child processes of the suite --------------------
_N73 = {}
#: measured host peaks of the three children (GB) - what running them side by
#: side needs of the control group
_N73_PEAK_GB = {"n48_ranks8": 30.0, "one_gpu": 57.0, "ranks8": 75.0}


def _n73_commands():
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return root, {
        # (the partitioned path against the oracle, cube N = 48 on 8 thread
        # ranks - a child as well, beside the other two)
        "n48_ranks8": ([sys.executable,
                        os.path.join(root, "tools", "parity_partitioned.py"),
                        "--n0", "48", "--ranks", "8"], {}),
        "one_gpu": ([sys.executable, os.path.join(root, "tools", "parity_large.py"),
                     "--geometry", "cube", "--level", "0", "--n0", "73",
                     "--algebraic"], {}),
        "ranks8": ([sys.executable,
                    os.path.join(root, "tools", "steady_thread_ranks.py"),
                    "--partitioned", "--algebraic", "--n0=73", "cube", "0", "8"],
                   {"FENAPACK_AMD_LOCAL_HANDOVER": "1"}),
    }


def n73_children(only=None):
    """The suite's runs on config 5's own mesh (cube N = 73, 9 934 793 DOF) -
    the one-GPU parity run (tools/parity_large.py) and the 8-thread-rank run
    (tools/steady_thread_ranks.py) - and the partitioned parity run at cube
    N = 48 (tools/parity_partitioned.py) are processes of their own (tens of GB
    of host memory each, under the scripts' own watchdog).  Where the control
    group has the memory for all of them (their measured peaks add up to
    162 GB: 200 GB available or more) they are started TOGETHER by whichever
    test asks first and run side by side - 58 s instead of 54 + 66 + 27 s one
    after the other; on a smaller host every test starts its own child when
    it asks (``only``).  Returns ``{name: {"proc", "out", "err", "t0"}}``;
    :func:`n73_result` waits for one of them."""
    import atexit
    import os
    import subprocess
    import tempfile
    import time
    from fenapack_amd import _guard
    root, cmds = _n73_commands()
    have = _guard.host_memory_available()
    together = have is None or have >= 200e9
    # (FENAPACK_AMD_SUITE_CHILDREN=sequential | together: the A/B switch)
    mode = os.environ.get("FENAPACK_AMD_SUITE_CHILDREN")
    if mode in ("sequential", "together"):
        together = mode == "together"
    want = list(cmds) if together else [only]
    env = dict(os.environ)
    for k in ("FENAPACK_AMD_NO_WATCHDOG", "PCD_REPLICATE_BELOW"):
        env.pop(k, None)
    env["FENAPACK_AMD_WATCHDOG"] = "1"
    if together:
        # (the builds thread their host work: share the cores)
        env.setdefault("FENAPACK_AMD_HOST_THREADS",
                       str(max(4, (os.cpu_count() or 8) // 3)))
    if not _N73:
        # (a session that ends before it collected them - a selection, `-x`
        # after a failure - does not leave GPU processes behind)
        def _reap():
            for rec in _N73.values():
                if rec["proc"].poll() is None:
                    rec["proc"].kill()
        atexit.register(_reap)
    for name in want:
        if name is None or name in _N73:
            continue
        cmd, extra = cmds[name]
        out = tempfile.TemporaryFile(mode="w+")
        err = tempfile.TemporaryFile(mode="w+")
        proc = subprocess.Popen(cmd, cwd=root, env=dict(env, **extra),
                                stdout=out, stderr=err, text=True)
        _N73[name] = {"proc": proc, "out": out, "err": err, "t0": time.time()}
    return _N73


def n73_result(name, timeout=1100):
    """(returncode, stdout, stderr) of one of :func:`n73_children`."""
    import subprocess
    rec = n73_children(only=name)[name]
    try:
        rc = rec["proc"].wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        rec["proc"].kill()
        rc = -9
    rec["out"].seek(0)
    rec["err"].seek(0)
    return rc, rec["out"].read(), rec["err"].read()
