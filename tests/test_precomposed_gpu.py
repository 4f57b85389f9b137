"""GPU suite, part 6: pre-composed inner solves (``fenapack_amd/compose.py``,
``pcd_mg_set_fused``, ``pcd_set_inner_factor``) against the step-by-step
engine path and against the oracle (which only knows the step-by-step form).
Tolerance 1e-11: the composed operators equal the recurrences in exact
arithmetic; only the association of the sums differs."""
import numpy as np
import pytest

import oracle
from fenapack_amd import _cabi as c
from fenapack_amd.compose import chebyshev_factors, vcycle_level
from helpers import (relerr, flow_state, configure_engine, push_multigrid)

pytestmark = pytest.mark.gpu


def fuse_all(e, slot, ops, chain, bounds, nu, A_fine, first=1):
    for l in range(first, len(ops)):
        A = ops[l] if l < len(ops) - 1 else A_fine
        Wd, Wu = vcycle_level(A, chain[l], bounds[l][0], bounds[l][1], nu, nu)
        e.mg_set_fused(slot, l, Wd, Wu)


@pytest.mark.parametrize("kind,level,nu", [("cavity", 3, 2), ("lshape", 3, 2),
                                           ("lshape", 2, 3), ("cube", 1, 2),
                                           ("cavity", 2, 1)])
def test_fused_multigrid_levels(hip_lib, kind, level, nu):
    kw = {"n0": 2} if kind == "cube" else {}
    st = flow_state(kind, level, **kw)
    pb, V, L = st["pb"], st["V"], st["L"]
    I = pb.interpolations()
    e, f, o = (c.Engine(hip_lib, "BRM1", 0), c.Engine(hip_lib, "BRM1", 0),
               oracle.Engine("BRM1"))
    data = {}
    for eng in (e, f, o):
        configure_engine(eng, st)
        data["p"] = push_multigrid(eng, c.KSP_AP, pb.Ap, I.chain("p"), nu=nu)
        data["u"] = push_multigrid(eng, c.KSP_A00, L["A00"], I.chain("u"),
                                   nu=nu)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
    # f: every level of both hierarchies in composed form
    fuse_all(f, c.KSP_AP, data["p"][0], I.chain("p"), data["p"][1], nu, pb.Ap)
    fuse_all(f, c.KSP_A00, data["u"][0], I.chain("u"), data["u"][1], nu,
             L["A00"])
    for eng in (e, f, o):
        eng.setup()
    rng = np.random.default_rng(21)
    for slot, n in ((c.KSP_AP, V.n_p), (c.KSP_A00, V.n_u)):
        b = rng.standard_normal(n)
        ye, yf, yo = (eng.inner_solve_np(slot, b) for eng in (e, f, o))
        assert relerr(ye, yo) < 1e-11
        assert relerr(yf, yo) < 1e-11 and relerr(yf, ye) < 1e-11
        assert not np.array_equal(yf, ye)         # it IS another code path
    x = rng.standard_normal(V.ndof)
    yo = o.fieldsplit_apply_np(x)
    assert relerr(f.fieldsplit_apply_np(x), yo) < 1e-11
    # only the two coarsest levels fused (what the size rule gives at scale)
    Lu = len(data["u"][0])
    for l in range(min(3, Lu), Lu):
        f.mg_set_fused(c.KSP_A00, l)                  # drop
    assert relerr(f.fieldsplit_apply_np(x), yo) < 1e-11
    # Richardson with two cycles: the second starts from a residual
    for eng in (f, o):
        eng.set_inner(c.KSP_A00, "richardson", "mg", 2, 0.0)
    b = rng.standard_normal(V.n_u)
    assert relerr(f.inner_solve_np(c.KSP_A00, b),
                  o.inner_solve_np(c.KSP_A00, b)) < 1e-11
    # graph replay of the composed cycle is bitwise the eager one
    y0 = f.fieldsplit_apply_np(x)
    f.graph_enable(True)
    assert np.array_equal(f.fieldsplit_apply_np(x), y0)
    assert np.array_equal(f.fieldsplit_apply_np(x), y0)
    f.graph_enable(False)
    # new values drop the composed form instead of applying a stale one
    from fenapack_amd.fem.multigrid import galerkin_chain, coarse_inverse
    ops = galerkin_chain(2.0 * L["A00"], I.chain("u"))
    for eng in (f, o):
        eng.set_inner(c.KSP_A00, "richardson", "mg", 1, 0.0)
        eng.update_system(2.0 * st["A"].data)
        eng.mg_update_values(c.KSP_A00, 0, coarse_inverse(ops[0]).data)
        for l in range(1, len(ops) - 1):
            eng.mg_update_values(c.KSP_A00, l, ops[l].data, *data["u"][1][l])
    assert relerr(f.inner_solve_np(c.KSP_A00, b),
                  o.inner_solve_np(c.KSP_A00, b)) < 1e-11


@pytest.mark.parametrize("its,nf", [(5, 2), (5, 1), (3, 2), (2, 2), (6, 3)])
def test_chebyshev_as_explicit_factors(hip_lib, its, nf):
    st = flow_state("lshape", 4)
    pb, V = st["pb"], st["V"]
    e, f, o = (c.Engine(hip_lib, "BRM1", 0), c.Engine(hip_lib, "BRM1", 0),
               oracle.Engine("BRM1"))
    for eng in (e, f, o):
        configure_engine(eng, st)
        eng.set_inner(c.KSP_AP, "cg", "jacobi", 8, 0.0)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", its, 0.0, 0.5, 2.0)
        eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
    F = chebyshev_factors(pb.Mp, 0.5, 2.0, its, max_factors=nf)
    assert len(F) <= nf
    f.set_inner_factors(c.KSP_MP, F)
    f.set_inner(c.KSP_MP, "preonly", "explicit", its, 0.0)
    for eng in (e, f, o):
        eng.setup()
    rng = np.random.default_rng(22)
    b = rng.standard_normal(V.n_p)
    yo = o.inner_solve_np(c.KSP_MP, b)
    assert relerr(f.inner_solve_np(c.KSP_MP, b), yo) < 1e-12
    assert relerr(e.inner_solve_np(c.KSP_MP, b), yo) < 1e-12
    assert int(f.info(c.INFO_ITS_MP)) == its        # reported as its steps
    # BRM1 folds the final sign into the last factor (-A x kernels)
    xp = rng.standard_normal(V.n_p)
    assert relerr(f.apply_np(xp), o.apply_np(xp)) < 1e-11
    x = rng.standard_normal(V.ndof)
    assert relerr(f.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)) < 1e-11
    # BRM2 uses the same solve without the sign
    g, o2 = c.Engine(hip_lib, "BRM2", 0), oracle.Engine("BRM2")
    for eng in (g, o2):
        configure_engine(eng, st)
        eng.set_inner(c.KSP_AP, "cg", "jacobi", 8, 0.0)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", its, 0.0, 0.5, 2.0)
    g.set_inner_factors(c.KSP_MP, F)
    g.set_inner(c.KSP_MP, "preonly", "explicit", its, 0.0)
    for eng in (g, o2):
        eng.setup()
    assert relerr(g.apply_np(xp), o2.apply_np(xp)) < 1e-11
    # stale factors are never applied: the step-by-step recurrence they stand
    # for takes over (bounds from set_inner) until new factors arrive
    f.update_values(c.MAT_MP, 2.0 * pb.Mp.data)
    assert relerr(f.inner_solve_np(c.KSP_MP, b), 0.5 * yo) < 1e-12
    f.set_inner_factors(c.KSP_MP, chebyshev_factors(2.0 * pb.Mp, 0.5, 2.0,
                                                    its, max_factors=nf))
    assert relerr(f.inner_solve_np(c.KSP_MP, b), 0.5 * yo) < 1e-12
    # going back to the step-by-step solver forgets the factors
    f.set_inner(c.KSP_MP, "chebyshev", "jacobi", its, 0.0, 0.5, 2.0)
    assert relerr(f.inner_solve_np(c.KSP_MP, b), 0.5 * yo) < 1e-12
    with pytest.raises(c.EngineError):
        f.set_inner(c.KSP_MP, "chebyshev", "explicit", its, 0.0, 0.5, 2.0)


def test_python_stack_composes_by_itself(hip_lib):
    """multigrid_inner_options through the Python API: M_p arrives as explicit
    factors, small levels arrive fused, and the result is the oracle's."""
    from fenapack_amd import PETScOptions
    from fenapack_amd.driver import make_solver, multigrid_inner_options
    from fenapack_amd.fem import Cavity
    pb = Cavity(4, nu=0.01)      # n_p = 6561: the Ap hierarchy has two levels
    PETScOptions.clear()
    multigrid_inner_options()
    w, nls, nlp = make_solver(pb, max_newton=2)
    nls.parameters["error_on_nonconvergence"] = False
    nls.solve(nlp, w.vector(), on_update=w.touch)
    hist = list(nls.krylov_history)
    ksp = nls.linear_solver().ksp()
    ksp0, ksp1 = ksp.pc.getFieldSplitSubKSP()
    pcd = ksp1.pc.getPythonContext()
    assert pcd.ksp_Mp.precomposed is not None
    assert len(ksp0.pc.mg_fused) >= 2 and len(pcd.ksp_Ap.pc.mg_fused) == 1
    o = oracle.mirror(oracle.Engine("BRM1"), pb, ksp)
    x = np.random.default_rng(5).standard_normal(pb.space.ndof)
    assert relerr(ksp.engine.fieldsplit_apply_np(x),
                  o.fieldsplit_apply_np(x)) < 1e-11
    # the same solve with composition switched off: same GMRES history
    PETScOptions.clear()
    multigrid_inner_options()
    for pre in ("fieldsplit_u_", "fieldsplit_p_PCD_Ap_"):
        PETScOptions.set(pre + "pc_mg_fuse_nnz", 0)
    PETScOptions.set("fieldsplit_p_PCD_Mp_ksp_chebyshev_precompose", 0)
    w2, nls2, nlp2 = make_solver(pb, max_newton=2)
    nls2.parameters["error_on_nonconvergence"] = False
    nls2.solve(nlp2, w2.vector(), on_update=w2.touch)
    PETScOptions.clear()
    k2 = nls2.linear_solver().ksp()
    assert k2.pc.getFieldSplitSubKSP()[0].pc.mg_fused == []
    assert list(nls2.krylov_history) == hist
    assert relerr(w2.vector(), w.vector()) < 1e-9


def test_graph_is_recaptured_when_the_kronecker_path_is_dropped(hip_lib):
    """Advisor finding (round 1): a value update that makes the components of
    an F (x) I operator differ switches kernels; a captured graph must not be
    replayed over it - and equal components later switch the fast path back
    on."""
    st = flow_state("lshape", 3)
    V, A = st["V"], st["A"]
    e, o = c.Engine(hip_lib, "BRM1", 0), oracle.Engine("BRM1")
    for eng in (e, o):
        configure_engine(eng, st)
        eng.set_inner(c.KSP_AP, "cg", "jacobi", 8, 0.0)
        eng.set_inner(c.KSP_MP, "chebyshev", "jacobi", 5, 0.0, 0.5, 2.0)
        eng.set_inner(c.KSP_A00, "chebyshev", "jacobi", 4, 0.0, 0.2, 2.2)
        eng.setup()
    assert int(e.info(c.INFO_A00_COMPONENTS)) == 2
    x = np.random.default_rng(6).standard_normal(V.ndof)
    e.graph_enable(True)
    assert relerr(e.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)) < 1e-11
    # second velocity component's rows scaled: same pattern, F (x) I no more
    rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
    odd_u = np.zeros(A.shape[0], bool)
    odd_u[V.is_u[1::2]] = True
    skew = np.where(odd_u[rows], 1.5 * A.data, A.data)
    for eng in (e, o):
        eng.update_system(skew)
    assert int(e.info(c.INFO_A00_COMPONENTS)) == 0
    assert relerr(e.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)) < 1e-11
    for eng in (e, o):
        eng.update_system(A.data)
    assert int(e.info(c.INFO_A00_COMPONENTS)) == 2        # fast path is back
    assert relerr(e.fieldsplit_apply_np(x), o.fieldsplit_apply_np(x)) < 1e-11
