"""CPU parity oracle for the PCD / fieldsplit apply path.

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, from
``__graft_entry__.smoke()`` and from ``bench.py``'s ``cpu_baseline`` leg.  The
product package ``fenapack_amd`` never imports this module.

``oracle/pcd_oracle.c`` is a plain-C restatement (same C ABI as
``include/pcd_engine.h`` under the prefix ``pcdo_``); ``oracle.reference_numpy``
is an independent numpy/scipy restatement used to generate and to check the
golden fixtures.  Parity status: the four ``apply`` bodies are pinned against
the reference's own ``fenapack/preconditioners.py`` (tests/golden); everything
that PETSc computes for the reference is "parity unpinned" (PETSc is absent
and its version is not pinned by the reference) - see DESIGN.md.
"""

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIBRARY_PATH = os.path.join(_HERE, "_build", "libpcd_oracle.so")
_library = None


def build(force=False):
    """Compile pcd_oracle.c with gcc (seconds)."""
    src = os.path.join(_HERE, "pcd_oracle.c")
    if (force or not os.path.exists(ORACLE_LIBRARY_PATH)
            or os.path.getmtime(ORACLE_LIBRARY_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return ORACLE_LIBRARY_PATH


def _oracle_library_class():
    """The product binding (``fenapack_amd._cabi.Library``) binds the HIP
    library only; the oracle exports the common part of the same ABI under
    the prefix ``pcdo_``, so that one test body can drive both.  The switch
    lives HERE, in test infrastructure."""
    from fenapack_amd._cabi import Library

    class OracleLibrary(Library):
        prefix = "pcdo_"
        hip = False                      # no pcd_fe_* / comm / graph symbols
    return OracleLibrary


def library():
    """PCD_ORACLE_LIB overrides the path (e.g. the -fsanitize build made by
    ``make -C oracle asan``)."""
    global _library
    if _library is None:
        path = os.environ.get("PCD_ORACLE_LIB") or build()
        _library = _oracle_library_class()(path)
    return _library


def Engine(variant="BRM1"):
    """An oracle engine with the same Python surface as the HIP engine."""
    from fenapack_amd._cabi import Engine as _Engine
    return _Engine(library(), variant, 0)


OMP_LIBRARY_PATH = os.path.join(_HERE, "_build", "libpcd_oracle_omp.so")


def omp_engine(variant="BRM1", threads=None):
    """Multi-core TIMING variant (OpenMP, reduction order not fixed): for
    bench.py's cpu_baseline only, never for parity.  Returns (engine,
    threads in effect)."""
    import ctypes
    from fenapack_amd._cabi import Engine as _Engine
    build()
    lib = _oracle_library_class()(OMP_LIBRARY_PATH)
    f = lib.lib.pcdo_set_threads
    f.argtypes, f.restype = [ctypes.c_int], ctypes.c_int
    n = f(int(threads or 0))
    eng = _Engine(lib, variant, 0)
    eng.set_threads = lambda k: f(int(k))
    # the TEAM port (one parallel region per PCApply, first-touch placement)
    prep = lib.lib.pcdo_team_prepare
    prep.argtypes, prep.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
    app = lib.lib.pcdo_team_fieldsplit_apply
    app.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    app.restype = ctypes.c_int

    def team_prepare(k):
        if prep(eng._h, int(k)):
            raise RuntimeError(lib.error())

    def team_fieldsplit_apply(x, y):
        import numpy as np
        assert x.dtype == np.float64 and y.dtype == np.float64
        assert x.flags.c_contiguous and y.flags.c_contiguous
        if app(eng._h, x.ctypes.data, y.ctypes.data):
            raise RuntimeError(lib.error())
    eng.team_prepare, eng.team_fieldsplit_apply = team_prepare, \
        team_fieldsplit_apply
    tri = lib.lib.pcdo_stream_triad
    tri.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    tri.restype = ctypes.c_double
    eng.stream_triad = lambda n=1 << 26, reps=3, k=0: tri(int(n), int(reps),
                                                          int(k))
    return eng, n


def mirror(o, pb, ksp):
    """Configure the oracle engine ``o`` with exactly what the Python stack
    handed to the HIP engine behind ``ksp`` (a set-up ``PCDKSP``): the three
    (four) pressure operators, the subfield BC, the system with its index
    sets (and the preconditioner matrix when it differs), every inner solver
    with its multigrid hierarchy and smoother bounds.  Used by the full-size
    parity tests and by bench.py's cpu_baseline leg - the checker sees the
    same inputs as the product, at the sizes the product runs."""
    from fenapack_amd import _cabi as c
    V = pb.space
    ksp0, ksp1 = ksp.pc.getFieldSplitSubKSP()
    pcd = ksp1.pc.getPythonContext()
    A, P = ksp.getOperators()
    o.set_velocity_block(V.dim)
    o.set_csr(c.MAT_AP, pcd.ksp_Ap.getOperators()[0].A)
    o.set_csr(c.MAT_MP, pcd.ksp_Mp.getOperators()[0].A)
    o.set_csr(c.MAT_KP, pcd.mat_Kp.A)
    slots = [(ksp0, c.KSP_A00), (pcd.ksp_Ap, c.KSP_AP), (pcd.ksp_Mp, c.KSP_MP)]
    if hasattr(pcd, "ksp_Rp"):
        o.set_csr(c.MAT_RP, pcd.ksp_Rp.getOperators()[0].A)
        slots.append((pcd.ksp_Rp, c.KSP_RP))
    o.set_bc(pb.bc_p_idx, pb.bc_p_val)
    pmat = None if (P is None or P is A or not P.isAssembled()) else P.A
    o.set_system(A.A, V.is_u, V.is_p, pmat)
    for k, slot in slots:
        if k.pc.type == "mg":
            d = k.pc.mg_data
            L = len(d["ops"])
            o.mg_begin(slot, L, d["nu"], d.get("nu_post", d["nu"]))
            o.mg_set_level(slot, 0, d["C"])
            for l in range(1, L):
                o.mg_set_level(slot, l, d["ops"][l] if l < L - 1 else None,
                               d["chain"][l], *d["bounds"][l])
            o.set_inner(slot, k.type, "mg", k.max_it, 0.0)
        else:
            # the interval the PRODUCT was handed (petsc.KSP.push_settings),
            # not a fresh estimate: the power iteration is warm-started, a
            # second run ends elsewhere and 240 Chebyshev steps on another
            # interval are another operator (round 6: 0.5 relative at cube
            # N = 48 between the engine and a checker mirrored that way)
            lo, hi = (getattr(k, "cheb_bounds_pushed", None)
                      or k._chebyshev_bounds()) if k.type == "chebyshev" \
                else (0.5, 2.0)
            o.set_inner(slot, k.engine_type, "jacobi", k.max_it,
                        k.rtol if k.type == "cg" else 0.0, lo, hi)
    o.setup()
    return o


def mirror_partitioned(o, pp, ksp, root=0):
    """:func:`mirror` for a PARTITIONED run (``fem.partition.PartitionedProblem``
    with the rank-local hand-over): every rank holds its rows of every
    partitioned operator; their rows are put back together on rank ``root``
    (``pp.host.allgather``: references between thread ranks, pickles between
    processes) and the oracle engine ``o`` of that rank is configured with the
    whole operators, the whole hierarchy (prolongation rows of the owners - a
    partitioned algebraic chain carries halo rows too), the same smoother
    bounds and coarse inverses.  COLLECTIVE: every rank calls it (``o`` may be
    ``None`` off the root); returns ``o`` on the root, ``None`` elsewhere.
    The checker then sees one global problem - what the ranks together were
    handed - at sizes where that fits one host."""
    import numpy as np
    import scipy.sparse as sp
    from fenapack_amd import _cabi as c
    host, me = pp.host, pp.rank
    V = pp.space

    def whole(M, rows=None):
        """Sum over ranks of row-sparse, global-shaped ``M`` (``rows``: keep
        only my rows ``[r0, r1)`` first - matrices that carry halo rows)."""
        M = sp.csr_matrix(M)
        if rows is not None:
            r0, r1 = rows
            ip = np.zeros(M.shape[0] + 1, dtype=np.int64)
            lo, hi = int(M.indptr[r0]), int(M.indptr[r1])
            ip[r0 + 1:r1 + 1] = M.indptr[r0 + 1:r1 + 1] - lo
            ip[r1 + 1:] = hi - lo
            M = sp.csr_matrix((M.data[lo:hi], M.indices[lo:hi], ip),
                              shape=M.shape)
        parts = host.allgather(M)
        if me != root:
            return None
        out = parts[0]
        for q in parts[1:]:
            out = out + q
        out = sp.csr_matrix(out)
        out.sort_indices()
        return out

    ksp0, ksp1 = ksp.pc.getFieldSplitSubKSP()
    pcd = ksp1.pc.getPythonContext()
    A, P = ksp.getOperators()
    Ap = whole(pcd.ksp_Ap.getOperators()[0].A)
    Mp = whole(pcd.ksp_Mp.getOperators()[0].A)
    Kp = whole(pcd.mat_Kp.A)
    Aw = whole(A.A)
    pmat = None if (P is None or P is A or not P.isAssembled()) else whole(P.A)
    if me == root:
        o.set_velocity_block(V.dim)
        o.set_csr(c.MAT_AP, Ap)
        o.set_csr(c.MAT_MP, Mp)
        o.set_csr(c.MAT_KP, Kp)
        o.set_bc(pp.bc_p_idx, pp.bc_p_val)
        o.set_system(Aw, V.is_u, V.is_p, pmat)
    for k, slot in ((ksp0, c.KSP_A00), (pcd.ksp_Ap, c.KSP_AP),
                    (pcd.ksp_Mp, c.KSP_MP)):
        if k.pc.type == "mg":
            d = k.pc.mg_data
            L = len(d["ops"])
            rows = d.get("rows") or [None] * L
            ops, chain = [], [None]
            for l in range(1, L):
                part = rows[l] is not None
                ops.append(None if l == L - 1 else
                           (whole(d["ops"][l]) if part else d["ops"][l]))
                # prolongation rows live with the level's rows
                chain.append(whole(d["chain"][l], rows[l]) if part
                             else d["chain"][l])
            if me == root:
                o.mg_begin(slot, L, d["nu"], d.get("nu_post", d["nu"]))
                o.mg_set_level(slot, 0, d["C"])
                for l in range(1, L):
                    o.mg_set_level(slot, l, ops[l - 1], chain[l],
                                   *d["bounds"][l])
                o.set_inner(slot, k.type, "mg", k.max_it, 0.0)
        elif me == root:
            lo, hi = (getattr(k, "cheb_bounds_pushed", None)
                      or k._chebyshev_bounds()) if k.type == "chebyshev" \
                else (0.5, 2.0)
            o.set_inner(slot, k.engine_type, "jacobi", k.max_it,
                        k.rtol if k.type == "cg" else 0.0, lo, hi)
        elif k.type == "chebyshev" and \
                getattr(k, "cheb_bounds_pushed", None) is None:
            k._chebyshev_bounds()              # (collective: reduces over ranks)
    if me != root:
        return None
    o.setup()
    return o
