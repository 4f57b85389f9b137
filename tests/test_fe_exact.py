"""CPU suite: the host producer's element matrices against closed-form
integration (oracle/fe_exact.py) - no quadrature rule or basis table shared."""
import numpy as np
import pytest

from oracle.fe_exact import element_matrices
from fenapack_amd.fem import Cavity, Cavity3D, BackwardStep, Channel3D


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


def test_robin_term_in_space_is_exact():
    """BRM2's boundary term of Kp, ``int_inflow (w.n) p q ds``
    (demo_navier-stokes-pcd.py:131-135), over the inflow FACES of a 3-D duct:
    with a wind in P2 and hat functions p, q the integrand has degree 4 and
    the 6-point facet rule integrates it exactly - checked against integrals
    over the unit square taken by hand."""
    pb = Channel3D(1, nu=0.02, n0=2, variant="BRM2")
    V, m = pb.space, pb.space.mesh
    assert V.dim == 3 and len(pb.robin_edges) == 2 * 4 * 4
    pl = V.robin_plan(pb.robin_edges)
    assert np.allclose(pl["normal"], [-1.0, 0.0, 0.0])           # outward at x = 0
    assert abs(pl["length"].sum() - 1.0) < 1e-14                  # the inflow face
    xyz, pc = V.node_coords, V.p_coords
    one = np.ones(V.n_p)
    # constant wind
    U = np.tile(np.array([2.0, 0.5, -0.3]), (V.nn, 1))
    R = V._boundary_flux_mass(U, pb.robin_edges)
    assert abs(one @ (R @ one) + 2.0) < 1e-13
    # quadratic wind w_x = y^2 + y z + 1 (in P2), tangential part ignored
    U = np.zeros((V.nn, 3))
    U[:, 0] = xyz[:, 1] ** 2 + xyz[:, 1] * xyz[:, 2] + 1.0
    U[:, 1] = 3.0
    R = V._boundary_flux_mass(U, pb.robin_edges)
    assert abs(one @ (R @ one) + 19.0 / 12.0) < 1e-13
    # p = y, q = z:  -int (y^2 + y z + 1) y z  over the unit square
    assert abs(pc[:, 2] @ (R @ pc[:, 1]) + (1 / 8. + 1 / 9. + 1 / 4.)) < 1e-13
    assert abs((R - R.T)).max() < 1e-15
    # rows of dofs off the inflow face are empty
    off = np.abs(pc[:, 0]) > 1e-12
    assert np.abs(R[off]).sum() == 0.0
    # ... and Kp carries it with the factor -1/nu
    K0 = V.assemble_Kp(pb.nu, U)
    K1 = V.assemble_Kp(pb.nu, U, robin_edges=pb.robin_edges)
    assert np.abs((K0 - K1) - R / pb.nu).max() < 1e-12


def test_channel_in_space_boundary_sets():
    """Inflow face ds(1), outflow face ds(2), walls no-slip; BRM1 pins the
    pressure operators on the inlet, BRM2 on the outlet
    (demo_navier-stokes-pcd.py:56-87, 138-141)."""
    for variant, x0 in (("BRM1", 0.0), ("BRM2", 1.0)):
        pb = Channel3D(1, nu=0.02, n0=2, variant=variant)
        V = pb.space
        assert np.allclose(V.p_coords[pb.bc_p_idx, 0], x0)
        assert pb.bc_p_idx.size == 25
        g = pb.bc_u_values(0.0).reshape(-1, 3)
        xyz = V.node_coords[pb._bc_nodes]
        assert np.all(g[:, 1:] == 0.0)
        inflow = g[:, 0] > 0
        assert inflow.any() and np.allclose(xyz[inflow, 0], 0.0)
        # the outflow face's interior carries no Dirichlet value
        out_int = (np.abs(V.node_coords[:, 0] - 1) < 1e-12) \
            & (np.abs(V.node_coords[:, 1] - 0.5) < 0.49) \
            & (np.abs(V.node_coords[:, 2] - 0.5) < 0.49)
        assert not np.isin(np.nonzero(out_int)[0], pb._bc_nodes).any()
