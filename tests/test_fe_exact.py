"""CPU suite: the host producer's element matrices against closed-form
integration (oracle/fe_exact.py) - no quadrature rule or basis table shared."""
import numpy as np
import pytest

from oracle.fe_exact import element_matrices
from fenapack_amd.fem import Cavity, Cavity3D, BackwardStep


@pytest.mark.parametrize("make", [lambda: Cavity(1), lambda: BackwardStep(1),
                                  lambda: Cavity3D(1, n0=2)])
def test_element_matrices_are_exact(make):
    pb = make()
    V, m = pb.space, pb.space.mesh
    rng = np.random.default_rng(5)
    U = rng.standard_normal((V.nn, V.dim))
    nu = 0.37
    w, _ = V.wind_at_qp(U)
    mass, stiff = V.p2_mass_cells(), V.p2_stiffness_cells()
    conv = V.p2_convection_cells(w)
    supg = V.p2_supg_cells(U, np.ones(m.num_cells))
    wg = np.einsum('cqd,cjd->cqj', w, V.gradlam)
    kp = np.einsum('cq,qi,cqj->cij', V.wq, V.psi, wg) / nu
    mass1 = np.einsum('cq,qi,qj->cij', V.wq, V.psi, V.psi)
    stiff1 = V.area[:, None, None] * np.einsum('cid,cjd->cij', V.gradlam,
                                               V.gradlam)
    div = -np.einsum('cq,qj,cqak->cajk', V.wq, V.psi, V.gphi)
    cells = rng.choice(m.num_cells, size=4, replace=False)
    for c in cells:
        ex = element_matrices(m.vertices[m.cells[c]], V.local_edges,
                              U[V.cell_dofs2[c]], nu)
        assert abs(ex["measure"] - V.area[c]) < 1e-14
        scale = lambda a: max(np.abs(a).max(), 1e-300)
        assert np.abs(mass[c] - ex["mass"]).max() < 1e-13 * scale(ex["mass"])
        assert np.abs(stiff[c] - ex["stiffness"]).max() \
            < 1e-12 * scale(ex["stiffness"])
        # degree 5 integrand, degree 5 rules: exact
        assert np.abs(conv[c] - ex["convection"]).max() \
            < 1e-12 * scale(ex["convection"])
        assert np.abs(kp[c] - ex["kp"]).max() < 1e-12 * scale(ex["kp"])
        assert np.abs(mass1[c] - ex["mass_p1"]).max() \
            < 1e-13 * scale(ex["mass_p1"])
        assert np.abs(stiff1[c] - ex["stiffness_p1"]).max() \
            < 1e-12 * scale(ex["stiffness_p1"])
        assert np.abs(div[c] - ex["div"]).max() < 1e-12 * scale(ex["div"])
        # streamline diffusion with a P2 wind has degree 6: its own rule
        # (exact to degree 7; the degree-5 rules were off by 10-25 % here)
        assert np.abs(supg[c] - ex["supg"]).max() < 1e-12 * scale(ex["supg"])


@pytest.mark.parametrize("d,n", [(2, 3), (2, 4), (3, 3), (3, 4)])
def test_conical_rules_integrate_monomials_exactly(d, n):
    """Stroud conical Gauss-Jacobi rules: exact to degree 2n - 1 against
    int lambda^alpha = d! prod(alpha_i!) / (d + |alpha|)! on the unit simplex
    (measure 1/d! divided out: weights sum to one)."""
    from itertools import product
    from math import factorial
    from fenapack_amd.fem.taylor_hood import _conical_rule
    pts, wts = _conical_rule(d, n)
    assert abs(wts.sum() - 1.0) < 1e-14 and np.all(wts > 0)
    assert np.allclose(pts.sum(axis=1), 1.0) and np.all(pts > 0)
    for alpha in product(range(2 * n), repeat=d + 1):
        if sum(alpha) > 2 * n - 1:
            continue
        exact = factorial(d)
        for a in alpha:
            exact *= factorial(a)
        exact /= factorial(d + sum(alpha))
        got = float(wts @ np.prod(pts ** np.array(alpha), axis=1))
        assert abs(got - exact) < 1e-13, alpha


def test_pattern_of_distinct_entries_matches_the_general_constructor():
    from fenapack_amd.fem.taylor_hood import FixedPattern
    rng = np.random.default_rng(0)
    n, m = 300, 170
    keys = rng.choice(n * m, size=4000, replace=False)
    rows, cols = keys // m, keys % m
    a = FixedPattern(rows, cols, (n, m))
    b = FixedPattern.from_unique(rows, cols, (n, m))
    assert a.nnz == b.nnz == 4000
    assert np.array_equal(a.indptr, b.indptr)
    assert np.array_equal(a.indices, b.indices)
    assert np.array_equal(a.inv, b.inv)
    vals = rng.standard_normal(4000)
    assert np.array_equal(a.assemble(vals).data, vals[b.order])
    with pytest.raises(ValueError):
        FixedPattern.from_unique(np.r_[rows, rows[:1]], np.r_[cols, cols[:1]],
                                 (n, m))
