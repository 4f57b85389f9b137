"""GPU suite: the HIP engine, called through the C ABI, against the oracle and
the golden vectors.  Tolerances (fp64; the only differences are summation
order inside rows / reductions):
  SpMV                              rel <= 1e-13
  fixed-iteration inner solves      rel <= 1e-11
  pcd_apply vs goldens (fixed its)  rel <= 1e-11
  pcd_apply vs goldens (tight CG)   rel <= 1e-9
  GMRES: identical iteration count, solution rel <= 1e-7
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from fenapack_amd import _cabi as c
from helpers import (VARIANTS, golden_files, csr_from, relerr,
                     load_pcd_operators, set_iter_cfg, set_tight_cg,
                     flow_state, configure_engine)

pytestmark = pytest.mark.gpu


def hip_engine(hip_lib, variant):
    return c.Engine(hip_lib, variant, 0)


@pytest.mark.parametrize("path", golden_files(),
                         ids=lambda p: p.split("/")[-1][:-4])
@pytest.mark.parametrize("variant", VARIANTS)
def test_apply_matches_reference_goldens(hip_lib, path, variant):
    d = np.load(path)
    e = hip_engine(hip_lib, variant)
    load_pcd_operators(e, d)
    set_iter_cfg(e)
    e.setup()
    y = e.apply_np(d["x"])
    assert relerr(y, d["y_%s_iter" % variant]) < 1e-11
    assert relerr(e.apply_np(d["x"]), y) == 0.0      # deterministic
    if "cavity" in path and variant.startswith("R"):
        return
    set_tight_cg(e)
    y = e.apply_np(d["x"])
    assert relerr(y, d["y_%s_direct" % variant]) < 1e-9


@pytest.mark.parametrize("kind,level", [("lshape", 3), ("cavity", 2),
                                        ("lshape", 5)])
def test_spmv_all_operators(hip_lib, kind, level):
    st = flow_state(kind, level, dt=0.2)
    V = st["V"]
    e, o = hip_engine(hip_lib, "RBRM1"), oracle.Engine("RBRM1")
    for eng in (e, o):
        configure_engine(eng, st)
        eng.setup()
    rng = np.random.default_rng(0)
    for which, ncols, nrows in ((c.MAT_AP, V.n_p, V.n_p),
                                (c.MAT_MP, V.n_p, V.n_p),
                                (c.MAT_KP, V.n_p, V.n_p),
                                (c.MAT_RP, V.n_p, V.n_p),
                                (c.MAT_A00, V.n_u, V.n_u),
                                (c.MAT_A01, V.n_p, V.n_u),
                                (c.MAT_A, V.ndof, V.ndof)):
        x = rng.standard_normal(ncols)
        assert relerr(e.spmv_np(which, x, nrows),
                      o.spmv_np(which, x, nrows)) < 1e-13, which


def test_empty_and_tiny_operators(hip_lib):
    # ragged input: empty rows, a 1x1 system, no BC dofs
    A = sp.csr_matrix(np.array([[2.0, 0, 0], [0, 0, 0], [1.0, 0, 3.0]]))
    for eng in (hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")):
        eng.set_csr(c.MAT_KP, A)
        y = eng.spmv_np(c.MAT_KP, np.array([1.0, 2.0, 3.0]), 3)
        assert np.array_equal(y, np.array([2.0, 0.0, 10.0]))
    one = sp.csr_matrix(np.array([[4.0]]))
    e = hip_engine(hip_lib, "BRM2")
    for m in (c.MAT_AP, c.MAT_MP, c.MAT_KP):
        e.set_csr(m, one)
    e.set_bc(np.zeros(0, dtype=np.int32), np.zeros(0))
    set_tight_cg(e)
    e.setup()
    # y = -(I + Ap^-1 Kp) Mp^-1 x = -(1 + 1) * 0.25 * x
    assert abs(e.apply_np(np.array([2.0]))[0] + 1.0) < 1e-14


@pytest.mark.parametrize("kind,level", [("lshape", 3), ("cavity", 2)])
def test_inner_solvers(hip_lib, kind, level):
    st = flow_state(kind, level, nu=0.1)
    V = st["V"]
    e, o = hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")
    for eng in (e, o):
        configure_engine(eng, st)
        eng.setup()
    rng = np.random.default_rng(1)
    bp, bu = rng.standard_normal(V.n_p), rng.standard_normal(V.n_u)
    cases = [
        (c.KSP_AP, bp, ("cg", "jacobi", 25, 0.0)),
        (c.KSP_AP, bp, ("cg", "none", 7, 0.0)),
        (c.KSP_MP, bp, ("chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)),
        (c.KSP_MP, bp, ("chebyshev", "jacobi", 1, 0.0, 0.5, 2.0)),
        (c.KSP_MP, bp, ("chebyshev", "jacobi", 0, 0.0, 0.5, 2.0)),
        (c.KSP_MP, bp, ("richardson", "jacobi", 1, 0.0)),
        (c.KSP_MP, bp, ("richardson", "jacobi", 6, 0.0)),
        (c.KSP_MP, bp, ("preonly", "jacobi", 1, 0.0)),
        (c.KSP_A00, bu, ("chebyshev", "jacobi", 9, 0.0, 0.05, 2.2)),
        (c.KSP_A00, bu, ("richardson", "jacobi", 3, 0.0)),
    ]
    for slot, b, cfg in cases:
        e.set_inner(slot, *cfg)
        o.set_inner(slot, *cfg)
        assert relerr(e.inner_solve_np(slot, b),
                      o.inner_solve_np(slot, b)) < 1e-11, cfg
    # tolerance-terminated CG: same iteration count, same answer
    for rtol in (1e-4, 1e-10):
        e.set_inner(c.KSP_AP, "cg", "jacobi", 5000, rtol)
        o.set_inner(c.KSP_AP, "cg", "jacobi", 5000, rtol)
        xe, xo = e.inner_solve_np(c.KSP_AP, bp), o.inner_solve_np(c.KSP_AP, bp)
        assert abs(e.info(c.INFO_ITS_AP) - o.info(c.INFO_ITS_AP)) <= 1
        assert relerr(xe, xo) < 10 * rtol
    # zero right-hand side must give zero, not NaN
    e.set_inner(c.KSP_AP, "cg", "jacobi", 10, 0.0)
    assert np.array_equal(e.inner_solve_np(c.KSP_AP, 0 * bp), 0 * bp)


def test_apply_bc_inserts_values(hip_lib):
    st = flow_state("lshape", 3)
    pb = st["pb"]
    e = hip_engine(hip_lib, "BRM1")
    configure_engine(e, st, with_system=False)
    e.set_bc(pb.bc_p_idx, np.arange(pb.bc_p_idx.size) + 1.0)
    e.setup()
    x = np.random.default_rng(5).standard_normal(pb.space.n_p)
    ref = x.copy()
    ref[pb.bc_p_idx] = np.arange(pb.bc_p_idx.size) + 1.0
    e.apply_bc(x)
    assert np.array_equal(x, ref)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("kind,level", [("lshape", 4), ("cavity", 3)])
def test_pcd_and_fieldsplit_apply_vs_oracle(hip_lib, variant, kind, level):
    flavour = "BRM1" if variant.endswith("1") else "BRM2"
    st = flow_state(kind, level, variant=flavour, dt=0.2)
    V = st["V"]
    e, o = hip_engine(hip_lib, variant), oracle.Engine(variant)
    for eng in (e, o):
        configure_engine(eng, st)
        set_iter_cfg(eng)
        eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
        eng.setup()
    rng = np.random.default_rng(7)
    xp = rng.standard_normal(V.n_p)
    assert relerr(e.apply_np(xp), o.apply_np(xp)) < 1e-11
    x = rng.standard_normal(V.ndof)
    assert relerr(e.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)) < 1e-11
    # linearity of the fixed-iteration (Chebyshev-only) preconditioner
    for eng in (e,):
        eng.set_inner(c.KSP_AP, "chebyshev", "jacobi", 6, 0.0, 0.05, 2.0)
        eng.set_inner(c.KSP_RP, "chebyshev", "jacobi", 6, 0.0, 0.05, 2.0)
    x2 = rng.standard_normal(V.ndof)
    lhs = e.fieldsplit_apply_np(2.0 * x - 3.0 * x2)
    rhs = 2.0 * e.fieldsplit_apply_np(x) - 3.0 * e.fieldsplit_apply_np(x2)
    assert relerr(lhs, rhs) < 1e-10


def test_values_update_keeps_pattern(hip_lib):
    st = flow_state("lshape", 3)
    V, L = st["V"], st["L"]
    e = hip_engine(hip_lib, "BRM1")
    configure_engine(e, st)
    e.setup()
    x = np.random.default_rng(8).standard_normal(V.n_u)
    y0 = e.spmv_np(c.MAT_A00, x, V.n_u)
    e.update_system(2.0 * st["A"].data)
    assert relerr(e.spmv_np(c.MAT_A00, x, V.n_u), 2.0 * y0) < 1e-15
    # a separate preconditioner matrix P feeds A00/A01, A stays the operator
    e.update_system(st["A"].data, pvals=3.0 * st["A"].data)
    assert relerr(e.spmv_np(c.MAT_A00, x, V.n_u), 3.0 * y0) < 1e-15
    xs = np.random.default_rng(9).standard_normal(V.ndof)
    As = sp.bmat([[L["A00"], L["A01"]], [L["A10"], None]]).tocsr()
    assert relerr(e.spmv_np(c.MAT_A, xs, V.ndof), As @ xs) < 1e-13
    K = st["Kp"]
    e.update_values(c.MAT_KP, -K.data)
    xp = np.random.default_rng(10).standard_normal(V.n_p)
    assert relerr(e.spmv_np(c.MAT_KP, xp, V.n_p), -(K @ xp)) < 1e-13


@pytest.mark.parametrize("kind,level,nu", [("lshape", 2, 0.1),
                                           ("cavity", 1, 0.1)])
def test_gmres_iteration_count_and_solution(hip_lib, kind, level, nu):
    st = flow_state(kind, level, nu=nu)
    e, o = hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")
    res = []
    for eng in (e, o):
        configure_engine(eng, st)
        eng.set_inner(c.KSP_AP, "cg", "jacobi", 2000, 1e-13)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
        eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 30, 0.0, 0.03, 2.2)
        eng.setup()
        res.append(eng.gmres_np(st["b"], rtol=1e-8, restart=40, max_it=600))
    (xe, ie, re_), (xo, io, ro) = res
    assert ie == io
    assert relerr(xe, xo) < 1e-7
    assert relerr(st["A"] @ xe, st["b"]) < 1e-6


def test_device_pointer_path_with_torch(hip_lib):
    import torch
    st = flow_state("lshape", 4)
    V = st["V"]
    e = hip_engine(hip_lib, "BRM1")
    configure_engine(e, st)
    set_iter_cfg(e)
    e.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
    e.setup()
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    xh = np.random.default_rng(11).standard_normal(V.ndof)
    x = torch.from_numpy(xh).cuda()
    y = torch.empty_like(x)
    e.fieldsplit_apply(x, y, c.MEM_DEVICE)
    torch.cuda.synchronize()
    assert relerr(y.cpu().numpy(), e.fieldsplit_apply_np(xh)) == 0.0


def test_error_paths(hip_lib):
    e = hip_engine(hip_lib, "RBRM2")
    with pytest.raises(c.EngineError):
        e.setup()
    with pytest.raises(c.EngineError):
        e.apply_np(np.zeros(3))
    with pytest.raises(c.EngineError):
        e.set_inner(c.KSP_AP, 9, 1, 1, 0.0, 0.5, 2.0)
    with pytest.raises(c.EngineError):
        c.Engine(hip_lib, "BRM1", 99)
    e.destroy()
    e.destroy()                          # idempotent


@pytest.mark.parametrize("kind,level", [("cavity", 3), ("lshape", 3)])
def test_multigrid_vcycle_vs_oracle(hip_lib, kind, level):
    from helpers import push_multigrid
    st = flow_state(kind, level)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()
    e, o = hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")
    for eng in (e, o):
        configure_engine(eng, st)
        push_multigrid(eng, c.KSP_AP, pb.Ap, I.chain("p"), cycles=2)
        push_multigrid(eng, c.KSP_A00, L["A00"], I.chain("u"), cycles=1)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
        eng.setup()
    rng = np.random.default_rng(12)
    for slot, n in ((c.KSP_AP, V.n_p), (c.KSP_A00, V.n_u)):
        b = rng.standard_normal(n)
        assert relerr(e.inner_solve_np(slot, b),
                      o.inner_solve_np(slot, b)) < 1e-11
    # asymmetric smoothing and a truncated hierarchy
    for eng in (e, o):
        push_multigrid(eng, c.KSP_A00, L["A00"], I.chain("u", 2), nu=3)
        eng.set_inner(c.KSP_A00, "preonly", "mg", 1, 0.0)
    b = rng.standard_normal(V.n_u)
    assert relerr(e.inner_solve_np(c.KSP_A00, b),
                  o.inner_solve_np(c.KSP_A00, b)) < 1e-11
    x = rng.standard_normal(V.ndof)
    assert relerr(e.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)) < 1e-11
    # refreshed operators between Newton steps: new values, same patterns
    from fenapack_amd.fem.multigrid import galerkin_chain, coarse_inverse
    ops = galerkin_chain(2.0 * L["A00"], I.chain("u", 2))
    for eng in (e, o):
        eng.update_system(2.0 * st["A"].data)
        eng.mg_update_values(c.KSP_A00, 0, coarse_inverse(ops[0]).data)
    assert relerr(e.inner_solve_np(c.KSP_A00, b),
                  o.inner_solve_np(c.KSP_A00, b)) < 1e-11
    assert relerr(2.0 * e.inner_solve_np(c.KSP_A00, b),
                  o.inner_solve_np(c.KSP_A00, 2.0 * b)) < 1e-11


def test_graph_replay_is_bitwise_equal_to_eager_launches(hip_lib):
    from helpers import push_multigrid
    st = flow_state("cavity", 3)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()
    e = hip_engine(hip_lib, "BRM1")
    configure_engine(e, st)
    push_multigrid(e, c.KSP_AP, pb.Ap, I.chain("p"))
    push_multigrid(e, c.KSP_A00, L["A00"], I.chain("u"))
    e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
    e.setup()
    rng = np.random.default_rng(13)
    x1, x2 = rng.standard_normal(V.ndof), rng.standard_normal(V.ndof)
    y1, y2 = e.fieldsplit_apply_np(x1), e.fieldsplit_apply_np(x2)
    e.graph_enable(True)
    assert np.array_equal(e.fieldsplit_apply_np(x1), y1)     # capture
    assert np.array_equal(e.fieldsplit_apply_np(x2), y2)     # replay
    assert np.array_equal(e.fieldsplit_apply_np(x1), y1)
    # a settings change re-captures; a tolerance-driven CG falls back to eager
    e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 3, 0.0, 0.5, 2.0)
    y3 = e.fieldsplit_apply_np(x1)
    e.graph_enable(False)
    assert np.array_equal(e.fieldsplit_apply_np(x1), y3)
    assert not np.array_equal(y3, y1)
    e.graph_enable(True)
    e.set_inner(c.KSP_MP, "cg", "jacobi", 50, 1e-6)
    y4 = e.fieldsplit_apply_np(x1)
    e.graph_enable(False)
    assert np.array_equal(e.fieldsplit_apply_np(x1), y4)
    # GMRES drives the replayed PCApply through staging copies
    e.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
    xa, ia, _ = e.gmres_np(st["b"], rtol=1e-8, restart=60, max_it=200)
    e.graph_enable(True)
    xb, ib, _ = e.gmres_np(st["b"], rtol=1e-8, restart=60, max_it=200)
    assert ia == ib and np.array_equal(xa, xb)


def test_two_component_fast_path_and_its_fallback(hip_lib):
    """A00 = F (x) I_2 (Picard) takes the double2 kernels; the same pattern
    with component-dependent values, and a Newton (coupled) block, must take
    the general kernels - all three against the oracle."""
    from fenapack_amd.fem import BackwardStep
    st = flow_state("lshape", 4)
    V, A00 = st["V"], st["L"]["A00"]
    rng = np.random.default_rng(14)
    x = rng.standard_normal(V.n_u)
    skew = A00.copy()
    rows = np.repeat(np.arange(V.n_u), np.diff(A00.indptr))
    skew.data = np.where(rows % 2 == 1, 2.0 * A00.data, A00.data)
    pbn = BackwardStep(3, nls="newton")
    xun = 0.1 * rng.standard_normal(pbn.space.n_u)
    newton = pbn.linearise(xun, np.zeros(pbn.space.n_p))["A00"]
    for M, xx in ((A00, x), (skew, x),
                  (newton, rng.standard_normal(pbn.space.n_u))):
        e, o = hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")
        for eng in (e, o):
            eng.set_csr(c.MAT_A00, M)
            eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 6, 0.0, 0.1, 2.2)
        assert relerr(e.spmv_np(c.MAT_A00, xx, M.shape[0]),
                      o.spmv_np(c.MAT_A00, xx, M.shape[0])) < 1e-13
        assert relerr(e.inner_solve_np(c.KSP_A00, xx),
                      o.inner_solve_np(c.KSP_A00, xx)) < 1e-11
        # values that break the structure later must be noticed too
        e.update_values(c.MAT_A00, np.where(
            np.repeat(np.arange(M.shape[0]), np.diff(M.indptr)) % 2 == 1,
            3.0 * M.data, M.data))
        o.update_values(c.MAT_A00, np.where(
            np.repeat(np.arange(M.shape[0]), np.diff(M.indptr)) % 2 == 1,
            3.0 * M.data, M.data))
        assert relerr(e.spmv_np(c.MAT_A00, xx, M.shape[0]),
                      o.spmv_np(c.MAT_A00, xx, M.shape[0])) < 1e-13


def test_supg_preconditioner_matrix_path(hip_lib):
    """BASELINE config 3 in miniature: the operator A stays unstabilised while
    the preconditioner matrix P carries the SUPG term in its 00-block
    (demo_navier-stokes-pcd.py:122-125; nonlinear_solvers.py:75-76).  A00/A01
    of the engine must come from P, the GMRES operator from A."""
    from fenapack_amd.fem import BackwardStep
    pb = BackwardStep(3, nu=0.005, stabilize=True)
    V = pb.space
    rng = np.random.default_rng(21)
    xu = 0.3 * rng.standard_normal(V.n_u)
    L = pb.linearise(xu, np.zeros(V.n_p))
    assert abs(L["P00"] - L["A00"]).max() > 0
    A = V.monolithic(L["A00"], L["A01"], L["A10"])
    P = V.monolithic(L["P00"], L["A01"], L["A10"])
    b = V.to_mixed(L["bu"], L["bp"])
    res = []
    for eng in (hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")):
        eng.set_csr(c.MAT_AP, pb.Ap)
        eng.set_csr(c.MAT_MP, pb.Mp)
        eng.set_csr(c.MAT_KP, pb.Kp(xu))
        eng.set_bc(pb.bc_p_idx, pb.bc_p_val)
        eng.set_system(A, V.is_u, V.is_p, P)
        eng.set_inner(c.KSP_AP, "cg", "jacobi", 3000, 1e-12)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
        eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 20, 0.0, 0.05, 2.2)
        eng.setup()
        xs = rng.standard_normal(V.n_u) if not res else res[0][3]
        y00 = eng.spmv_np(c.MAT_A00, xs, V.n_u)
        ya = eng.spmv_np(c.MAT_A, np.concatenate([xs, np.zeros(V.n_p)]),
                         V.ndof)
        # a fixed, short Krylov run: the comparison is about which matrix
        # feeds which slot, not about convergence at this Peclet number
        x, its, _ = eng.gmres_np(b, rtol=1e-30, restart=12, max_it=12)
        res.append((y00, ya, (x, its), xs))
    (y00e, yae, (xe, ie), xs), (y00o, yao, (xo, io), _) = res
    assert relerr(y00e, L["P00"] @ xs) < 1e-13          # A00 slot holds P00
    assert relerr(yae[:V.n_u], L["A00"] @ xs) < 1e-13   # operator holds A00
    assert relerr(y00e, y00o) < 1e-13 and relerr(yae, yao) < 1e-13
    assert ie == io == 12 and relerr(xe, xo) < 1e-8


def test_single_reduction_cg_vs_oracle(hip_lib):
    """-ksp_cg_single_reduction on the HIP engine (one fused SpMV + two dots,
    one update kernel per iteration) against the oracle's restatement."""
    st = flow_state("lshape", 4)
    V = st["V"]
    e, o = hip_engine(hip_lib, "BRM1"), oracle.Engine("BRM1")
    rng = np.random.default_rng(31)
    b = rng.standard_normal(V.n_p)
    for eng in (e, o):
        configure_engine(eng, st)
    for its in (0, 1, 2, 9, 40):
        ys = []
        for eng in (e, o):
            eng.set_inner(c.KSP_AP, "cgsr", "jacobi", its, 0.0)
            eng.setup()
            ys.append(eng.inner_solve_np(c.KSP_AP, b))
            assert int(eng.info(c.INFO_ITS_AP)) == its
        assert its == 0 or relerr(ys[0], ys[1]) < 1e-11
    got = []
    for eng in (e, o):
        eng.set_inner(c.KSP_AP, "cgsr", "jacobi", 5000, 1e-10)
        eng.setup()
        got.append((eng.inner_solve_np(c.KSP_AP, b), int(eng.info(c.INFO_ITS_AP))))
    assert abs(got[0][1] - got[1][1]) <= 1, (got[0][1], got[1][1])
    assert relerr(got[0][0], got[1][0]) < 1e-8
    # inside the PCD apply, against standard CG on the same engine
    xp = rng.standard_normal(V.n_p)
    e.set_inner(c.KSP_MP, "cgsr", "jacobi", 200, 1e-12)
    y1 = e.apply_np(xp)
    e.set_inner(c.KSP_AP, "cg", "jacobi", 5000, 1e-10)
    e.set_inner(c.KSP_MP, "cg", "jacobi", 200, 1e-12)
    assert relerr(y1, e.apply_np(xp)) < 1e-7
