"""CPU suite: the static plans the device operator producer is driven by
(contribution lists, Galerkin gather lists, injection maps) evaluated with
numpy against the host assembler / scipy products."""
import numpy as np
import pytest
import scipy.sparse as sp

from fenapack_amd.device_producer import (_contribution_plan, galerkin_plan)
from fenapack_amd.fem import BackwardStep, Cavity, Cavity3D
from fenapack_amd.fem.multigrid import injection_map


def _gather(ptr, src, cells, const=None):
    """What k_fe_gather computes (component-major element storage)."""
    s = np.add.reduceat(np.concatenate([cells[src], [0.0]]), ptr[:-1])
    s[np.diff(ptr) == 0] = 0.0
    return s if const is None else s + const


@pytest.mark.parametrize("make", [lambda: Cavity(2), lambda: BackwardStep(2),
                                  lambda: Cavity3D(1, n0=2)])
def test_contribution_plan_reproduces_bincount_assembly(make):
    pb = make()
    V = pb.space
    rng = np.random.default_rng(0)
    for name, nloc in (("SS", V.na), ("PP", V.nvl)):
        pat = V._patterns(False)[name]
        nc = V.mesh.num_cells
        vals = rng.standard_normal((nc, nloc, nloc))
        ptr, src = _contribution_plan(pat.inv, nc, nloc * nloc, pat.nnz)
        assert ptr[-1] == nc * nloc * nloc
        # component-major storage: [ab][cell]
        cells = vals.reshape(nc, -1).T.ravel()
        got = _gather(ptr.astype(np.int64), src, cells)
        ref = pat.assemble(vals).data
        assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("make,d", [(lambda: Cavity(3), 2),
                                    (lambda: BackwardStep(2), 2),
                                    (lambda: Cavity3D(1, n0=2), 3)])
def test_galerkin_plan_equals_the_triple_product(make, d):
    pb = make()
    V = pb.space
    rng = np.random.default_rng(1)
    xu = rng.standard_normal(V.n_u)
    A00 = pb.linearise(xu, np.zeros(V.n_p))["A00"]
    F = sp.csr_matrix(A00)[::d, ::d]
    F.sort_indices()
    pat = V._patterns(False)["SS"]
    assert np.array_equal(F.indices, pat.indices)      # F x I_d on SS
    Ps = sp.csr_matrix(pb.interpolations().velocity[-1])[::d, ::d]
    rows = np.repeat(np.arange(V.nn), np.diff(pat.indptr))
    b_ptr, b_src, b_w, c_ptr, c_src, c_w, indptr, indices = \
        galerkin_plan(rows, pat.indices, Ps)
    B = np.add.reduceat(b_w * F.data[b_src], b_ptr[:-1])
    C = np.add.reduceat(c_w * B[c_src], c_ptr[:-1])
    got = sp.csr_matrix((C, indices, indptr), shape=(Ps.shape[1],) * 2)
    ref = (Ps.T @ F @ Ps).tocsr()
    # scipy drops exact cancellations: structural pattern >= scipy's
    assert got.nnz >= ref.nnz
    assert abs(got - ref).max() < 1e-13 * abs(ref).max()
    # the pattern does not depend on the values
    F2 = F.copy()
    F2.data[:] = 1.0
    ref2 = (Ps.T @ F2 @ Ps).tocsr()
    ref2.sort_indices()
    assert np.array_equal(ref2.indices, indices)


@pytest.mark.parametrize("make,d", [(lambda: Cavity(3), 2),
                                    (lambda: Cavity3D(1, n0=2), 3)])
def test_injection_map_picks_coincident_nodes(make, d):
    pb = make()
    chain = pb.interpolations().velocity
    fine = pb.space
    coarse = pb._same_problem_on_level(len(chain) - 2).space
    inj = injection_map(chain[-1], d)
    assert inj.size == coarse.nn and np.unique(inj).size == inj.size
    assert np.allclose(fine.node_coords[inj], coarse.node_coords, atol=1e-12)
    with pytest.raises(ValueError):
        injection_map(sp.csr_matrix(chain[-1]) * 0.5, d)


@pytest.mark.parametrize("make", [lambda: Cavity(2, nls="newton"),
                                  lambda: Cavity3D(1, n0=2, nls="newton")])
def test_block_positions_rebuild_the_newton_block(make):
    """pcd_fe_set_newton's position lists: scattering delta_ij F + N_ij
    through them (what k_fe_scatter_blocks does) reproduces the host's
    coupled velocity block; the structural pattern is (pattern of F) x
    ones(d, d)."""
    from fenapack_amd.device_producer import block_positions, coupled_pattern
    pb = make()
    V, d = pb.space, pb.space.dim
    rng = np.random.default_rng(5)
    U = rng.standard_normal((V.nn, d))
    ref = V.assemble_A00(pb.nu, U, newton=True)              # host, no BCs
    patS, patA = V._patterns(False)["SS"], V._patterns(True)["A00"]
    K = coupled_pattern(patS.indptr, patS.indices, V.nn, d)
    assert K.nnz == patA.nnz == d * d * patS.nnz
    assert np.array_equal(K.indptr, patA.indptr)
    assert np.array_equal(K.indices, patA.indices)
    pos = block_positions(K, patS.indptr, patS.indices, d)
    assert np.array_equal(np.sort(pos.ravel()), np.arange(K.nnz))
    # scalar pieces on the pattern of F
    w, gw = V.wind_at_qp(U)
    S = pb.nu * V.p2_stiffness_cells() + V.p2_convection_cells(w)
    F = np.bincount(patS.inv, weights=S.ravel(), minlength=patS.nnz)
    Ncells = np.einsum('cq,qa,qb,cqkd->kdcab', V.wq, V.phi, V.phi, gw)
    vals = np.zeros(K.nnz)
    for i in range(d):
        for j in range(d):
            Nij = np.bincount(patS.inv, weights=Ncells[i, j].ravel(),
                              minlength=patS.nnz)
            vals[pos[i * d + j]] = (F if i == j else 0.0) + Nij
    assert np.allclose(vals, ref.data, rtol=1e-12, atol=1e-12)
    # an operator that lacks entries of the coupled pattern is refused
    with pytest.raises(ValueError):
        block_positions(V._patterns(False)["A00"].matrix(
            np.ones(V._patterns(False)["A00"].nnz)), patS.indptr,
            patS.indices, d)


def test_kron_layout_of_a_level_matches_scipy():
    """pcd_fe_bind_pattern's closed form: entry k of row a of F, component
    c, sits at d*rowptr[a] + c*len_a + (k - rowptr[a]) of the sorted CSR of
    F x I_d on interleaved dofs."""
    pb = Cavity(2)
    V, d = pb.space, 2
    pat = V._patterns(False)["SS"]
    rng = np.random.default_rng(2)
    F = rng.standard_normal(pat.nnz)
    K = sp.kron(pat.matrix(F), sp.identity(d), format="csr")
    K.sort_indices()
    rows = np.repeat(np.arange(V.nn), np.diff(pat.indptr))
    b0 = pat.indptr[rows].astype(np.int64)
    ln = np.diff(pat.indptr)[rows]
    k = np.arange(pat.nnz)
    got = np.zeros(K.nnz)
    for c in range(d):
        got[d * b0 + c * ln + (k - b0)] = F
    # scipy keeps the structural zeros of kron(F, I)? it does not store them:
    assert K.nnz == d * pat.nnz
    assert np.array_equal(got, K.data)


@pytest.mark.parametrize("dim", [2, 3])
def test_native_galerkin_plan_is_the_numpy_plan_term_for_term(dim, monkeypatch):
    """``libpcd_host``'s product-plan builder (threads over the rows) against
    the numpy builder: the same entries, the same terms in the same order -
    so the device's weighted gathers add in the same order whichever built
    the plan.  Smoothed-aggregation prolongator (several entries per row)."""
    import scipy.sparse as sp
    from fenapack_amd import _host
    from fenapack_amd import device_producer as dp
    from fenapack_amd.amg import smoothed_aggregation_chain
    from fenapack_amd.fem import Cavity, Cavity3D
    if _host.use_numpy():
        pytest.skip("libpcd_host is not in use (FENAPACK_AMD_HOST=numpy)")
    pb = Cavity(3, nu=0.01) if dim == 2 else Cavity3D(0, nu=0.01, n0=8)
    V = pb.space
    A00 = sp.csr_matrix(pb.linearise(np.zeros(V.n_u), np.zeros(V.n_p))["A00"])
    chain = smoothed_aggregation_chain(A00, block=dim, coarse_rows=200,
                                       theta=0.02)
    F = _host.kron_factor(A00, dim)
    Ps = sp.csr_matrix(chain[-1])[::dim, ::dim]
    rows_f = np.repeat(np.arange(F.shape[0]), np.diff(F.indptr))
    nat = dp.galerkin_plan(rows_f, F.indices, Ps)
    monkeypatch.setattr(_host, "use_numpy", lambda: True)
    ref = dp.galerkin_plan(rows_f, F.indices, Ps)
    assert len(nat) == len(ref) == 8
    for a, b in zip(nat, ref):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    # ... and the plan computes the product
    Fv = np.random.default_rng(0).standard_normal(F.nnz)
    Fm = sp.csr_matrix((Fv, F.indices, F.indptr), shape=F.shape)
    b_ptr, b_src, b_w, c_ptr, c_src, c_w, ip, ix = nat
    B = np.add.reduceat(np.append(b_w * Fv[b_src], 0.0), b_ptr[:-1])
    B[np.diff(b_ptr) == 0] = 0.0
    C = np.add.reduceat(np.append(c_w * B[c_src], 0.0), c_ptr[:-1])
    C[np.diff(c_ptr) == 0] = 0.0
    got = sp.csr_matrix((C, ix, ip), shape=(Ps.shape[1],) * 2)
    ref_m = (Ps.T @ Fm @ Ps).tocsr()
    assert abs(got - ref_m).max() < 1e-12 * abs(ref_m).max()
