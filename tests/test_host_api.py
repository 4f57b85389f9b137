"""CPU suite, part 3: host logic of the API mirror (no engine calls)."""
import numpy as np
import pytest
import scipy.sparse as sp

from fenapack_amd import (PCDAssembler, PCDForm, PCDKSP, PCDKrylovSolver,
                          PETScOptions)
from fenapack_amd.utils import allow_only_one_call
from fenapack_amd.petsc import IS, KSP, Mat, Options, Vec, estimate_emax
from fenapack_amd.field_split_backend import PCDInterface, SubfieldBC
from fenapack_amd.fem import BackwardStep
from fenapack_amd.fem.forms import navier_stokes_forms, DirichletBC
from helpers import golden_files, csr_from, relerr


def test_allow_only_one_call():
    # known answers restated from the reference's test/unit/test_utils.py:24-57
    class C(object):
        @allow_only_one_call
        def foo(self, *args, **kwargs):
            """Foo"""
            return args, kwargs

        @allow_only_one_call
        def bar(self, *args, **kwargs):
            """Bar"""
            return args, kwargs

        def baz(self, *args, **kwargs):
            """Baz"""
            return args, kwargs

    o = C()
    assert (o.foo.__doc__, o.bar.__doc__, o.baz.__doc__) == ("Foo", "Bar",
                                                             "Baz")
    for f in (o.foo, o.bar, o.baz):
        assert f(1, 2, 3, four=5) == ((1, 2, 3), {"four": 5})
    for f in (o.foo, o.bar):
        with pytest.raises(RuntimeError):
            f(1, 2, 3, four=5)
    assert o.baz(1, 2, 3, four=5) == ((1, 2, 3), {"four": 5})
    assert C().foo(7) == ((7,), {})       # per instance, not per class


def test_subksp_prefixes_follow_an_early_prefix():
    # test/unit/test_fieldsplit.py:84-96: a prefix set before the PC set-up
    # propagates to foo_fieldsplit_u_/p_
    ksp = PCDKSP()
    ksp.setOptionsPrefix("foo_")
    ksp.pc.setFieldSplitIS(["u", IS([0, 1])], ["p", IS([2])])
    with pytest.raises(RuntimeError):
        PCDKSP().pc.getFieldSplitSubKSP()       # before setUp (issue #160)
    ksp.pc.setUp()
    k0, k1 = ksp.pc.getFieldSplitSubKSP()
    assert k0.getOptionsPrefix() == "foo_fieldsplit_u_"
    assert k1.getOptionsPrefix() == "foo_fieldsplit_p_"


def test_options_db_and_unsupported_solvers_are_rejected():
    PETScOptions.clear()
    PETScOptions.set("t_ksp_type", "chebyshev")
    PETScOptions.set("t_ksp_max_it", 7)
    PETScOptions.set("t_ksp_chebyshev_eigenvalues", "0.5, 2.0")
    PETScOptions.set("t_pc_type", "jacobi")
    k = KSP()
    k.setOptionsPrefix("t_")
    k.setFromOptions()
    assert (k.type, k.max_it, k.cheb_eigs, k.pc.type) == \
        ("chebyshev", 7, (0.5, 2.0), "jacobi")
    assert Options("t_").getString("pc_python_type", "") == ""
    PETScOptions.set("t_pc_type", "ilu")
    with pytest.raises(ValueError):
        k.setFromOptions()
    # -ksp_cg_single_reduction [ext PETSc] selects the engine's fused variant
    PETScOptions.set("t_pc_type", "jacobi")
    PETScOptions.set("t_ksp_type", "cg")
    PETScOptions.set("t_ksp_cg_single_reduction", "true")
    k.setFromOptions()
    assert (k.type, k.engine_type) == ("cg", "cgsr")
    PETScOptions.set("t_ksp_cg_single_reduction", "false")
    k.setFromOptions()
    assert k.engine_type == "cg"
    PETScOptions.set("t_ksp_type", "chebyshev")
    PETScOptions.set("t_pc_type", "jacobi")
    PETScOptions.set("t_ksp_type", "pipecr")
    with pytest.raises(ValueError):
        k.setFromOptions()
    PETScOptions.clear()


def test_pcd_form_defaults_and_missing_forms():
    pb = BackwardStep(0)
    w, forms = navier_stokes_forms(pb)
    a = PCDAssembler(**forms)
    # assembling.py:98-106
    assert a.get_pcd_form("ap").is_constant()
    assert a.get_pcd_form("mp").is_constant()
    assert not a.get_pcd_form("kp").is_constant()
    assert a.get_pcd_form("gp").is_phantom()
    with pytest.raises(AttributeError):
        a.get_pcd_form("fp")
    with pytest.raises(AttributeError):
        a.get_pcd_form("nonsense")
    assert isinstance(a.get_pcd_form("kp"), PCDForm)
    assert a.function_space() is pb.space


def test_forms_assemble_on_the_mixed_space_and_split_back():
    pb = BackwardStep(1)
    V = pb.space
    w, forms = navier_stokes_forms(pb)
    a = PCDAssembler(**forms)
    M = Mat()
    a.mp(M)
    assert M.getSize() == (V.ndof, V.ndof)
    sub = M.createSubMatrix(IS(V.is_p))
    assert abs(sub.A - pb.Mp).max() == 0.0
    A = Mat()
    a.system_matrix(A)
    L = pb.linearise(*w.split())
    assert abs(A.createSubMatrix(IS(V.is_u), IS(V.is_p)).A - L["A01"]).max() \
        == 0.0
    state = A.state
    a.system_matrix(A)
    assert A.state == state + 1            # re-assembly is noticed


def test_subfield_bc_maps_mixed_dofs_to_subfield_indices():
    pb = BackwardStep(2, variant="BRM1")
    V = pb.space
    bc = DirichletBC(V, V.is_p[pb.bc_p_idx], pb.bc_p_val + 3.0)
    sbc = SubfieldBC(bc, IS(V.is_p))
    assert np.array_equal(np.sort(sbc.indices), np.sort(pb.bc_p_idx))
    assert np.all(sbc.values == 3.0) and not sbc.is_homogeneous()
    # a velocity BC has no image in the pressure subfield
    bcu = DirichletBC(V, V.is_u[pb.bc_u_idx], pb.bc_u_values(0.0))
    assert SubfieldBC(bcu, IS(V.is_p)).indices.size == 0
    # inlet of the L-shape at level l has 2^l + 1 vertices
    assert pb.bc_p_idx.size == 2 ** 2 + 1


@pytest.mark.parametrize("path", golden_files()[:3],
                         ids=lambda p: p.split("/")[-1][:-4])
def test_rp_builder_matches_the_reference_built_golden(path):
    d = np.load(path)
    A01 = csr_from(d, "A01")
    Mu = sp.diags(d["Mu_diag"]).tocsr()
    iface = object.__new__(PCDInterface)
    R = iface._build_approx_Ap(Mat(Mu), Mat(A01)).A
    assert relerr(R.toarray(), csr_from(d, "Rp").toarray()) < 1e-14


def test_emax_estimate_is_a_tight_lower_bound():
    pb = BackwardStep(2)
    e = estimate_emax(pb.Mp, iters=40)
    true = np.linalg.eigvals((sp.diags(1 / pb.Mp.diagonal()) @ pb.Mp)
                             .toarray()).real.max()
    assert 0.9 * true < e <= true * (1 + 1e-12)
    assert true <= 2.0 + 1e-12             # the documented 2D P1 bound


def test_krylov_solver_requires_the_hip_library(monkeypatch, tmp_path):
    from fenapack_amd import _cabi
    monkeypatch.setattr(_cabi, "_hip_library", None)
    monkeypatch.setattr(_cabi, "HIP_LIBRARY_PATH", str(tmp_path / "x.so"))
    pb = BackwardStep(0)
    w, forms = navier_stokes_forms(pb)
    a = PCDAssembler(**forms)
    s = PCDKrylovSolver()
    A = Mat()
    a.system_matrix(A)
    s.set_operators(A, A)
    with pytest.raises(_cabi.EngineError):
        s.init_pcd(a)                       # no silent CPU fallback


def test_pcd_context_guards_and_timers():
    # preconditioners.py:59-60 (work-vec count is frozen) and :66-67
    # (re-initialisation); timer names of preconditioners.py:98,148,219,264
    from fenapack_amd import (PCDPC_BRM1, PCDPC_BRM2, PCDRPC_BRM1,
                              PCDRPC_BRM2, timings, list_timings)

    class FakeVec(object):
        def duplicate(self):
            return FakeVec()

    for cls in (PCDPC_BRM1, PCDPC_BRM2, PCDRPC_BRM1, PCDRPC_BRM2):
        ctx = cls()
        v = FakeVec()
        w1 = ctx.get_work_vecs(v, 2)
        assert len(w1) == 2 and ctx.get_work_vecs(v, 2) is w1    # cached
        with pytest.raises(ValueError, match="Changing number of work vecs"):
            ctx.get_work_vecs(v, 1)
        marker = object()
        ctx.init_pcd(marker)
        assert ctx.interface is marker
        with pytest.raises(RuntimeError, match="Reinitialization"):
            ctx.init_pcd(marker)

    class FakeEngine(object):
        calls = 0

        def apply(self, x, y, mem):
            FakeEngine.calls += 1

    class FakeInterface(object):
        engine = FakeEngine()

    timings(clear=True)
    for cls in (PCDPC_BRM1, PCDPC_BRM2, PCDRPC_BRM1, PCDRPC_BRM2):
        ctx = cls()
        ctx.init_pcd(FakeInterface())
        ctx.apply(None, 1, 2)
        ctx.apply(None, 1, 2)
    t = timings()
    for cls in ("PCDPC_BRM1", "PCDPC_BRM2", "PCDRPC_BRM1", "PCDRPC_BRM2"):
        assert t["FENaPack: %s apply" % cls][0] == 2
    assert FakeEngine.calls == 8
    assert "FENaPack: PCDPC_BRM1 apply" in list_timings(file=open("/dev/null", "w"))


def test_setup_timers_carry_the_reference_names():
    # field_split.py:89 - the one set-up that needs no engine
    from fenapack_amd import timings
    timings(clear=True)
    ksp = PCDKSP()
    ksp.setOptionsPrefix("foo_")

    class A(object):
        def function_space(self):
            class V(object):
                is_u, is_p = [0, 1], [2]
            return V()
    with pytest.raises(Exception):
        ksp.init_pcd(A())          # dies at the engine; the PC was set up
    assert "FENaPack: PCDKSP PC foo_ setup" in timings()


def test_output_vectors_are_validated_before_the_engine_writes():
    from fenapack_amd import _cabi
    ok = np.zeros(5)
    assert _cabi._out(ok, 5) == ok.ctypes.data
    for bad in (np.zeros(5, dtype=np.float32), np.zeros(10)[::2],
                np.zeros((5, 1))):
        with pytest.raises(_cabi.EngineError):
            _cabi._out(bad, 5)
    with pytest.raises(_cabi.EngineError, match="holds 4 entries"):
        _cabi._out(np.zeros(4), 5)
    ro = np.zeros(5)
    ro.setflags(write=False)
    with pytest.raises(_cabi.EngineError):
        _cabi._out(ro, 5)
    with pytest.raises(_cabi.EngineError):
        _cabi._out(None, 5)


def test_stabilization_parameter_in_three_dimensions():
    # the public wrapper must shape the nodal wind by the space dimension
    from fenapack_amd import StabilizationParameterSD
    from fenapack_amd.fem import Cavity3D
    pb = Cavity3D(0, nu=1e-3, n0=2)
    V = pb.space
    U = np.random.default_rng(3).standard_normal((V.n_u // 3, 3))

    class Wind(object):                       # nodal velocity array + space
        def __init__(self):
            self.V = V

        def __array__(self, *args, **kwargs):
            return U.ravel()
    delta = StabilizationParameterSD(Wind(), 1e-3)()
    ref = V.supg_delta(U, 1e-3, 1.0)
    assert delta.shape == ref.shape and np.array_equal(delta, ref)
    assert (delta > 0).any()


def test_algebraic_hierarchy_from_the_matrix_alone():
    """-pc_type gamg (fenapack_amd/amg.py): a prolongation chain built from
    the operator, no mesh - the role hypre BoomerAMG plays for the reference
    (demo_navier-stokes-pcd.py:153-160).  Properties: constants are
    reproduced away from Dirichlet rows, a velocity block F (x) I_2 keeps its
    Kronecker structure on every level, and a V(2,2) cycle on the pressure
    Laplacian is a good preconditioner."""
    import scipy.sparse.linalg as spla
    from fenapack_amd.amg import (smoothed_aggregation_chain, aggregate,
                                  _strength, scalar_stencil)
    from fenapack_amd.fem import Cavity
    from fenapack_amd.fem.multigrid import galerkin_chain
    pb = Cavity(3, nu=0.01)
    Ap = sp.csr_matrix(pb.Ap)
    S = _strength(Ap, 0.02)
    agg, nagg = aggregate(S)
    assert agg.max() == nagg - 1 and nagg < Ap.shape[0] / 3
    # Dirichlet rows - nothing couples to them - get no coarse representative
    # (kept as singletons they would survive on every level: a 3-D cavity has
    # 6 N^2 of them and its hierarchy stalled), everything else has one
    isolated = np.diff(S.indptr) == 0
    assert isolated.any() and np.all(agg[isolated] == -1)
    assert np.all(agg[~isolated] >= 0)
    assert set(np.unique(agg[~isolated])) == set(range(nagg))
    chain = smoothed_aggregation_chain(Ap, coarse_rows=300)
    assert chain[0] is None and chain[-1].shape[0] == Ap.shape[0]
    for a, b in zip(chain[1:-1], chain[2:]):
        assert a.shape[0] == b.shape[1]
    # smoothed prolongator of a Laplacian: constants stay constants wherever
    # the row sums of A vanish (interior rows)
    P = chain[-1]
    interior = np.abs(Ap @ np.ones(Ap.shape[0])) < 1e-12
    img = P @ (P.T @ np.ones(P.shape[0]))
    assert interior.sum() > 0.9 * Ap.shape[0]
    # two-level preconditioner for CG
    ops = galerkin_chain(Ap, chain)
    Pf, Ac = chain[-1], ops[-2]
    lu = spla.splu(sp.csc_matrix(Ac)) if len(chain) == 2 else None
    dinv = 1.0 / Ap.diagonal()

    def vcycle(b):
        x = 0.7 * dinv * b
        x = x + 0.7 * dinv * (b - Ap @ x)
        r = Pf.T @ (b - Ap @ x)
        e = lu.solve(r) if lu is not None else spla.spsolve(sp.csc_matrix(Ac), r)
        x = x + Pf @ e
        x = x + 0.7 * dinv * (b - Ap @ x)
        return x + 0.7 * dinv * (b - Ap @ x)
    cnt = [0]
    b = np.random.default_rng(0).standard_normal(Ap.shape[0])
    M = spla.LinearOperator(Ap.shape, matvec=vcycle)
    x, info = spla.gmres(Ap, b, M=M, rtol=1e-8, restart=60,
                         callback=lambda r: cnt.__setitem__(0, cnt[0] + 1),
                         callback_type="pr_norm")
    assert info == 0 and cnt[0] <= 20, cnt
    # velocity block: aggregation on the scalar stencil, P = P_F (x) I_2
    st_ = pb.linearise(*pb.initial_guess())
    A00 = sp.csr_matrix(st_["A00"])
    assert scalar_stencil(A00, 2) is not None
    ch = smoothed_aggregation_chain(A00, block=2, coarse_rows=500)
    for Pl in ch[1:]:
        assert scalar_stencil(sp.csr_matrix(Pl.T @ Pl), 2) is not None
    # three components, enclosed flow: every boundary node is a Dirichlet row;
    # the hierarchy reaches the coarse limit instead of stalling on them
    from fenapack_amd.fem import Cavity3D
    pb3 = Cavity3D(1, nu=0.01, n0=4)
    A3 = sp.csr_matrix(pb3.linearise(*pb3.initial_guess())["A00"])
    ch3 = smoothed_aggregation_chain(A3, block=3, coarse_rows=2000)
    assert ch3[1].shape[1] <= 2000 and ch3[1].shape[1] % 3 == 0
    zero_rows = np.diff(sp.csr_matrix(ch3[-1]).indptr) == 0
    assert zero_rows.sum() >= 3 * 6 * 15 ** 2       # the boundary nodes
    # options: gamg and the reference's hypre spelling select it
    PETScOptions.clear()
    PETScOptions.set("x_pc_type", "hypre")
    PETScOptions.set("x_pc_hypre_type", "boomeramg")
    k = KSP()
    k.setOptionsPrefix("x_")
    k.setFromOptions()
    assert k.pc.type == "mg" and k.pc.mg_algebraic
    PETScOptions.clear()


def test_apply_accepts_what_petsc_hands_a_pcpython_context():
    """PETSc calls ``ctx.apply(pc, x, y)`` with petsc4py Vecs (host arrays
    behind ``getArray``); the context must route them as HOST pointers, this
    package's device Vecs as DEVICE pointers."""
    from fenapack_amd import PCDPC_BRM1
    from fenapack_amd import _cabi
    from fenapack_amd.preconditioners import _buffer

    class PetscVec(object):                # the two methods apply needs
        def __init__(self, n):
            self.a = np.arange(float(n))

        def getArray(self, readonly=False):
            v = self.a.view()
            v.setflags(write=not readonly)
            return v

    seen = []

    class FakeEngine(object):
        def apply(self, x, y, mem):
            seen.append((type(x).__name__, x.flags.writeable,
                         y.flags.writeable, mem))
            y[:] = 2.0 * x

    class FakeInterface(object):
        engine = FakeEngine()
    ctx = PCDPC_BRM1()
    ctx.init_pcd(FakeInterface())
    x, y = PetscVec(4), PetscVec(4)
    ctx.apply(None, x, y)
    assert seen == [("ndarray", False, True, _cabi.MEM_HOST)]
    assert np.array_equal(y.a, 2.0 * x.a)
    buf, mem = _buffer(np.zeros(3))
    assert mem == _cabi.MEM_HOST


def test_subfield_bc_arrays_merge_in_application_order():
    pb = BackwardStep(1)
    w, forms = navier_stokes_forms(pb)
    a = PCDAssembler(**forms)
    V = pb.space
    iface = PCDInterface(a, Mat(), IS(V.is_u), IS(V.is_p))
    idx, val = iface.subfield_bc_arrays()
    assert idx.dtype == np.int32 and val.dtype == np.float64
    assert np.array_equal(np.sort(idx), np.sort(pb.bc_p_idx))
    assert np.all(val == 0.0)


def test_blas_pools_are_capped_but_never_resized_under_thread_binding(
        monkeypatch):
    """fenapack_amd.limit_blas_threads: the package caps the BLAS pools at
    import (small dense calls on a pool of one thread per core cost ~100 x
    their work); an explicit OPENBLAS_NUM_THREADS wins, and nothing is resized
    when an OpenMP runtime binds threads (new BLAS threads would inherit the
    bound main thread's one-core mask)."""
    import fenapack_amd
    threadpoolctl = pytest.importorskip("threadpoolctl")
    monkeypatch.setenv("OPENBLAS_NUM_THREADS", "8")
    assert fenapack_amd.limit_blas_threads() is None
    monkeypatch.delenv("OPENBLAS_NUM_THREADS")
    monkeypatch.setenv("OMP_PROC_BIND", "spread")
    assert fenapack_amd.limit_blas_threads() is None
    monkeypatch.delenv("OMP_PROC_BIND")
    monkeypatch.setenv("FENAPACK_AMD_BLAS_THREADS", "0")
    assert fenapack_amd.limit_blas_threads() is None
    before = [d["num_threads"] for d in threadpoolctl.threadpool_info()
              if d["user_api"] == "blas"]
    lim = fenapack_amd.limit_blas_threads(1)
    try:
        assert all(d["num_threads"] == 1
                   for d in threadpoolctl.threadpool_info()
                   if d["user_api"] == "blas")
    finally:
        lim.restore_original_limits()
    assert [d["num_threads"] for d in threadpoolctl.threadpool_info()
            if d["user_api"] == "blas"] == before


def test_producer_refuses_builds_beyond_the_host_memory(monkeypatch):
    """fem/multigrid._check_size: a mistyped level - or R rank threads each
    building a 10 M-DOF problem - fails with a message instead of driving the
    host out of memory."""
    from fenapack_amd.fem import multigrid as mg
    from fenapack_amd.fem import Cavity3D
    have = mg.host_memory_available()
    assert have is None or have > 0
    monkeypatch.setattr(mg, "host_memory_available", lambda: 64e9)
    monkeypatch.setenv("FENAPACK_AMD_MAX_CELLS", "3000000")
    mg._check_size(6 * 36 ** 3, 3, "cube N = 36")              # 13 GB: fine
    with pytest.raises(MemoryError, match="concurrent build"):
        mg._check_size(6 * 73 ** 3, 3, "cube N = 73")          # 112 GB
    monkeypatch.setenv("FENAPACK_AMD_CONCURRENT_BUILDS", "8")
    with pytest.raises(MemoryError, match="8 concurrent"):
        mg._check_size(6 * 36 ** 3, 3, "cube N = 36 on 8 rank threads")
    monkeypatch.setenv("FENAPACK_AMD_IGNORE_MEMORY", "1")
    mg._check_size(6 * 73 ** 3, 3, "cube N = 73")
    monkeypatch.delenv("FENAPACK_AMD_IGNORE_MEMORY")
    monkeypatch.delenv("FENAPACK_AMD_CONCURRENT_BUILDS")
    monkeypatch.delenv("FENAPACK_AMD_MAX_CELLS")
    with pytest.raises(ValueError, match="FENAPACK_AMD_MAX_CELLS"):
        Cavity3D(5, nu=0.01, n0=4)                 # the cell limit comes first


def test_rank_local_row_slices_of_the_python_mirror():
    """field_split._local_rows / petsc.DeviceMat in the rank-local mode: the
    rows, and the positions of their values in the caller's arrays, that go to
    pcd_set_system_local / pcd_set_csr_local / pcd_update_*.  (The engine side
    runs on thread ranks in -m gpu; this is the host logic alone.)"""
    from fenapack_amd import _cabi as c
    from fenapack_amd.field_split import PCDKSP as KSPcls
    from fenapack_amd.petsc import DeviceMat
    rng = np.random.default_rng(2)
    nu, npr = 14, 5
    n = nu + npr
    perm = rng.permutation(n)
    is_u, is_p = np.sort(perm[:nu]), np.sort(perm[nu:])
    A = sp.random(n, n, density=0.4, format="csr", random_state=4)
    A.sort_indices()

    class FakeEngine(object):
        local_handover = True

        def __init__(self):
            self.calls = []

        def row_range(self, n_global, velocity=False):
            return (4, 10) if velocity else (2, n_global)

        def set_csr_local(self, which, rows, shape):
            self.calls.append(("set_csr_local", which, rows.copy(), shape))

        def update_values(self, which, vals):
            self.calls.append(("update_values", which, np.array(vals)))

    class IS_(object):
        def __init__(self, idx):
            self.indices = idx
    ksp = KSPcls.__new__(KSPcls)
    ksp.engine = FakeEngine()
    rows, pos = ksp._local_rows(Mat(A), IS_(is_u), IS_(is_p))
    assert np.array_equal(rows, np.concatenate([is_u[4:10], is_p[2:]]))
    assert np.array_equal(A.data[pos], A[rows].data)
    # a refresh reads the same positions of the re-assembled values
    A2 = A.copy()
    A2.data = A2.data * 3.0
    assert np.array_equal(A2.data[pos], A2[rows].data)
    # operators of the pressure space: rows [2, n_p) and their values
    Mp = sp.random(npr, npr, density=0.6, format="csr", random_state=5)
    Mp.sort_indices()
    eng = FakeEngine()
    M = DeviceMat(eng, c.MAT_MP, Mp)
    name, which, got, shape = eng.calls[0]
    assert name == "set_csr_local" and which == c.MAT_MP and shape == Mp.shape
    assert (got != Mp[2:]).nnz == 0
    M.update(Mp * 2.0)
    assert np.array_equal(eng.calls[1][2], (Mp * 2.0).tocsr()[2:].data)


def test_rank_local_multigrid_push_of_the_python_mirror(monkeypatch):
    """petsc._push_multigrid in the rank-local mode: partitioned levels go to
    pcd_mg_set_level_local as this rank's rows of A, P and (when the level
    below is partitioned too) P^T; levels of at most PCD_REPLICATE_BELOW rows
    go over whole; value refreshes carry the rank's values."""
    from fenapack_amd import _cabi as c
    from fenapack_amd.fem import Cavity
    from fenapack_amd.fem.multigrid import galerkin_chain
    monkeypatch.setenv("PCD_REPLICATE_BELOW", "700")
    pb = Cavity(3, nu=0.01)
    V = pb.space
    A00 = sp.csr_matrix(pb.linearise(*pb.initial_guess())["A00"])
    chain = pb.interpolations().chain("u")
    ops = galerkin_chain(A00, chain)

    class Lib(object):
        hip = False                              # (no composed levels here)

    class FakeEngine(object):
        local_handover = True
        velocity_block = 2
        L = Lib()

        def __init__(self):
            self.calls = []

        def row_range(self, n_global, velocity=False):
            cut = n_global // 2
            cut -= cut % 2 if velocity else 0
            return (cut, n_global)               # rank 1 of 2

        def __getattr__(self, name):
            def record(*a, **k):
                self.calls.append((name,) + a)
            return record
    eng = FakeEngine()
    PETScOptions.clear()
    k = KSP()
    k.setType("richardson")
    k.pc.setType("mg")
    k.setOperators(Mat(A00))
    k.pc.setMGInterpolations(chain)
    k.bind(eng, c.KSP_A00)
    k.setUp()
    # (the KSP cuts the chain at the largest level whose explicit inverse stays
    # small: what it handed over is in pc.mg_data)
    ops, chain = k.pc.mg_data["ops"], k.pc.mg_data["chain"]
    sizes = [o.shape[0] for o in ops]
    assert max(sizes[:-1]) > 700 or len(sizes) == 2
    Lv = len(ops)
    by = {}
    for call in eng.calls:
        by.setdefault(call[0], []).append(call[1:])
    assert by["mg_begin"][0][:2] == (c.KSP_A00, Lv)
    local = {a[1]: a for a in by["mg_set_level_local"]}
    whole = {a[1]: a for a in by["mg_set_level"]}
    for l in range(1, Lv):
        n = sizes[l]
        part = l == Lv - 1 or n > 700
        assert (l in local) == part and (l in whole) == (not part)
        if not part:
            continue
        slot, lev, ng, A_rows, P_rows, R_rows, emin, emax = local[l]
        r0, r1 = eng.row_range(n, velocity=True)
        assert ng == n and 0 < emin < emax
        assert (A_rows is None) == (l == Lv - 1)
        if A_rows is not None:
            assert (A_rows != ops[l][r0:r1]).nnz == 0
        assert (P_rows != sp.csr_matrix(chain[l])[r0:r1]).nnz == 0
        if sizes[l - 1] > 700:
            c0, c1 = eng.row_range(sizes[l - 1], velocity=True)
            assert (R_rows != sp.csr_matrix(chain[l].T)[c0:c1]).nnz == 0
        else:
            assert R_rows is None
    assert 0 in whole                            # the coarse inverse, whole
    # a value refresh: same pattern, the rank's values of partitioned levels
    eng.calls.clear()
    A2 = A00.copy()
    A2.data = A2.data * 2.0
    k.setOperators(Mat(A2))
    k.setUp()
    ops2 = k.pc.mg_data["ops"]
    upd = {a[1]: a for a in eng.calls if a[0] == "mg_update_values"
           for a in [a[1:]]}
    for l in range(1, Lv - 1):
        vals = upd[l][2]
        if sizes[l] > 700:
            r0, r1 = eng.row_range(sizes[l], velocity=True)
            assert np.array_equal(vals, ops2[l][r0:r1].data)
        else:
            assert np.array_equal(vals, ops2[l].data)
    assert upd[Lv - 1][2] is None                # finest level: bounds only
    PETScOptions.clear()
