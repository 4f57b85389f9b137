"""GPU suite, part 5: the SpMV / smoother / CG kernels on adversarial CSR
shapes a caller could hand over - ragged and empty rows, sizes around the
wave (64) and workgroup (256) widths, rows longer than the LDS tile, dense
blocks, rectangular operators, two-component (F x I_2) structure and its
near misses - always against scipy / the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from fenapack_amd import _cabi as c
from helpers import relerr

pytestmark = pytest.mark.gpu


def _random_csr(rng, n, m, avg, long_rows=0, empty_frac=0.1):
    rows, cols = [], []
    for i in range(n):
        if rng.random() < empty_frac:
            continue
        k = min(m, max(1, rng.poisson(avg)))
        if long_rows and i % max(n // long_rows, 1) == 0:
            k = min(m, 6000)
        cc = rng.choice(m, size=k, replace=False)
        rows += [i] * k
        cols += list(cc)
    A = sp.csr_matrix((rng.standard_normal(len(rows)), (rows, cols)),
                      shape=(n, m))
    A.sort_indices()
    return A


@pytest.mark.parametrize("n,m,avg,long_rows", [
    (1, 1, 1, 0), (63, 63, 5, 0), (64, 65, 9, 0), (255, 257, 12, 0),
    (1000, 777, 30, 0), (5000, 5000, 7, 0), (7001, 7001, 20, 3),
    (300, 9000, 400, 0), (4097, 4097, 2, 0)])
def test_spmv_random_shapes(hip_lib, n, m, avg, long_rows):
    rng = np.random.default_rng(n * 7 + m)
    A = _random_csr(rng, n, m, avg, long_rows)
    e = c.Engine(hip_lib, "BRM1", 0)
    # A01 is the only rectangular slot; square ones go through Kp as well
    e.set_csr(c.MAT_A01, A)
    x = rng.standard_normal(m)
    ref = A @ x
    assert relerr(e.spmv_np(c.MAT_A01, x, n), ref) < 1e-13
    if n == m:
        e.set_csr(c.MAT_KP, A)
        assert relerr(e.spmv_np(c.MAT_KP, x, n), ref) < 1e-13


def _tile_form(monkeypatch, form):
    """``0``: stream kernels; ``2``: the LDS-staged vector-tile kernels
    (k_*_tc) on every F (x) I operator, whatever its size, direct form;
    ``2staged``: the form operators beyond the Infinity Cache get - matrix
    entries lane-major, straight to registers (k_*_lm)."""
    monkeypatch.setenv("PCD_VEC_TILE", form[0])
    if form == "2staged":
        monkeypatch.setenv("PCD_NT_BYTES", "0")


@pytest.mark.parametrize("vec_tile", ["0", "2", "2staged"])
@pytest.mark.parametrize("nc", [2, 3])
@pytest.mark.parametrize("nodes,mnodes", [(1, 1), (33, 33), (500, 321),
                                          (4000, 4000)])
def test_multi_component_operators_and_near_misses(hip_lib, monkeypatch, nodes,
                                                   mnodes, nc, vec_tile):
    _tile_form(monkeypatch, vec_tile)
    rng = np.random.default_rng(nodes)
    F = _random_csr(rng, nodes, mnodes, 9, empty_frac=0.0)
    K = sp.kron(F, sp.identity(nc), format="csr")
    K.sort_indices()
    x = rng.standard_normal(nc * mnodes)
    variants = [K]
    # near misses: one value differs / one entry removed / odd size
    K2 = K.copy()
    K2.data[K2.nnz // 2] *= 1.5
    variants.append(K2)
    K3 = K.tolil()
    r = int(K.nonzero()[0][0])
    cidx = int(K.nonzero()[1][0])
    K3[r, cidx] = 0.0
    K3 = K3.tocsr()
    K3.eliminate_zeros()
    K3.sort_indices()
    variants.append(K3)
    for M in variants:
        e = c.Engine(hip_lib, "BRM1", 0)
        e.set_csr(c.MAT_A01, M)
        assert relerr(e.spmv_np(c.MAT_A01, x, M.shape[0]), M @ x) < 1e-13


@pytest.mark.parametrize("nc", [2, 3])
@pytest.mark.parametrize("nodes,m,avg,nt", [(1100, 400, 5, False),
                                            (4000, 1500, 14, False),
                                            (4000, 1500, 14, True),
                                            (1700, 9000, 300, False)])
def test_row_blocked_operators_and_near_misses(hip_lib, monkeypatch, nc, nodes,
                                               m, avg, nt):
    """The discrete gradient's structure: the nc rows of a node share their
    column pattern, the values differ (k_spmv_rk: one column index per
    node-entry).  Near misses - one entry removed, one row longer - must fall
    back to the general kernels; a value update must reach the row-blocked
    copy."""
    if nt:
        monkeypatch.setenv("PCD_NT_BYTES", "0")
    rng = np.random.default_rng(nodes + nc)
    F = _random_csr(rng, nodes, m, avg, empty_frac=0.05)
    K = sp.kron(F, np.ones((nc, 1)), format="csr")
    K.sort_indices()
    K.data[:] = rng.standard_normal(K.nnz)
    x = rng.standard_normal(m)
    add = rng.standard_normal(nc * nodes)
    K2 = K.tolil()
    r, cidx = int(K.nonzero()[0][5]), int(K.nonzero()[1][5])
    K2[r, cidx] = 0.0
    K2 = K2.tocsr()
    K2.eliminate_zeros()
    K2.sort_indices()
    for M in (K, K2):
        e = c.Engine(hip_lib, "BRM1", 0)
        e.set_velocity_block(nc)
        e.set_csr(c.MAT_A01, M)
        assert relerr(e.spmv_np(c.MAT_A01, x, M.shape[0]), M @ x) < 1e-13
        M.data[:] = rng.standard_normal(M.nnz)           # new values, same pattern
        e.update_values(c.MAT_A01, M.data)
        assert relerr(e.spmv_np(c.MAT_A01, x, M.shape[0]), M @ x) < 1e-13


@pytest.mark.parametrize("vec_tile", ["0", "2", "2staged"])
@pytest.mark.parametrize("nc", [2, 3])
@pytest.mark.parametrize("nodes", [1, 70, 2500, 9000])
def test_smoother_on_multi_component_spd(hip_lib, monkeypatch, nodes, nc,
                                         vec_tile):
    """F (x) I_nc with F SPD: the fused multi-component Chebyshev start/step
    (and their fall-back when one value breaks the structure); with the
    stream kernels and with the LDS-staged vector-tile kernels."""
    _tile_form(monkeypatch, vec_tile)
    rng = np.random.default_rng(nodes + nc)
    B = _random_csr(rng, nodes, nodes, 6, empty_frac=0.0)
    F = (B @ B.T + sp.identity(nodes) * (1.0 + abs(B).sum(axis=1).max())).tocsr()
    A = sp.kron(F, sp.identity(nc), format="csr")
    A.sort_indices()
    b = rng.standard_normal(nc * nodes)
    A_broken = A.copy()
    A_broken.data[A.indptr[-1] - 1] *= 1.25
    for M in (A, A_broken):
        e, o = c.Engine(hip_lib, "BRM1", 0), oracle.Engine("BRM1")
        for eng in (e, o):
            eng.set_csr(c.MAT_AP, M)
        for cfg in (("chebyshev", "jacobi", 1, 0.0, 0.3, 1.8),
                    ("chebyshev", "jacobi", 6, 0.0, 0.3, 1.8),
                    ("richardson", "jacobi", 3, 0.0)):
            e.set_inner(c.KSP_AP, *cfg)
            o.set_inner(c.KSP_AP, *cfg)
            assert relerr(e.inner_solve_np(c.KSP_AP, b),
                          o.inner_solve_np(c.KSP_AP, b)) < 1e-11, cfg


@pytest.mark.parametrize("n", [1, 64, 257, 3000])
def test_smoother_and_cg_on_random_spd(hip_lib, n):
    rng = np.random.default_rng(n)
    B = _random_csr(rng, n, n, 6, empty_frac=0.0)
    A = (B @ B.T + sp.identity(n) * (1.0 + abs(B).sum(axis=1).max())).tocsr()
    A.sort_indices()
    b = rng.standard_normal(n)
    e, o = c.Engine(hip_lib, "BRM1", 0), oracle.Engine("BRM1")
    for eng in (e, o):
        eng.set_csr(c.MAT_AP, A)
    for cfg in (("chebyshev", "jacobi", 7, 0.0, 0.3, 1.8),
                ("chebyshev", "none", 4, 0.0, 0.5, 3.0),
                ("richardson", "jacobi", 5, 0.0),
                ("cg", "jacobi", 9, 0.0), ("cg", "none", 200, 1e-9)):
        e.set_inner(c.KSP_AP, *cfg)
        o.set_inner(c.KSP_AP, *cfg)
        tol = 1e-7 if cfg[3] else 1e-10
        assert relerr(e.inner_solve_np(c.KSP_AP, b),
                      o.inner_solve_np(c.KSP_AP, b)) < tol, cfg


def _full_unsorted(rng, M):
    """CSR that stores EVERY entry of the dense ``M`` with the columns of each
    row in a random order (what a scipy product can return)."""
    n, m = M.shape
    cols = np.concatenate([rng.permutation(m) for _ in range(n)])
    rows = np.repeat(np.arange(n), m)
    A = sp.csr_matrix((M[rows, cols], cols, np.arange(n + 1) * m),
                      shape=(n, m))
    assert not A.has_sorted_indices
    return A


@pytest.mark.parametrize("n", [64, 130])
def test_full_pattern_with_unsorted_columns(hip_lib, n):
    """A full-pattern CSR whose column indices are not 0..n-1 in order must
    not take the row-major dense kernel (it ignores ``col``): scalar operator
    and the F (x) I_2 form."""
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    A = _full_unsorted(rng, M)
    e = c.Engine(hip_lib, "BRM1", 0)
    e.set_csr(c.MAT_KP, A)
    x = rng.standard_normal(n)
    assert relerr(e.spmv_np(c.MAT_KP, x, n), M @ x) < 1e-12
    # two components per node, every block stored, columns shuffled per row
    K = sp.kron(sp.csr_matrix(M), sp.identity(2), format="csr")
    Kd = K.toarray()
    rows, cols = [], []
    for i in range(2 * n):
        cc = np.nonzero(Kd[i])[0]
        cols.append(rng.permutation(cc))
        rows.append(len(cc))
    cols = np.concatenate(cols)
    indptr = np.concatenate([[0], np.cumsum(rows)])
    r = np.repeat(np.arange(2 * n), rows)
    K2 = sp.csr_matrix((Kd[r, cols], cols, indptr), shape=K.shape)
    e.set_csr(c.MAT_A01, K2)
    x2 = rng.standard_normal(2 * n)
    assert relerr(e.spmv_np(c.MAT_A01, x2, 2 * n), Kd @ x2) < 1e-12


@pytest.mark.parametrize("nc", [2, 3])
def test_smoother_on_a_fully_dense_multi_component_level(hip_lib, nc):
    """F (x) I_nc with a FULL F of 64+ nodes (a small Galerkin / gamg level):
    its row blocks may fit no LDS tile (rb2 == 0), the SpMV still takes the
    dense multi-component kernel and the smoother must fall back to the
    general kernels instead of launching a tile kernel with rb2 = 0."""
    nodes = 600                      # 600 entries per node row: > one 32-row tile
    rng = np.random.default_rng(nc)
    B = rng.standard_normal((nodes, nodes)) / np.sqrt(nodes)
    F = B @ B.T + 2.0 * np.eye(nodes)
    A = sp.kron(sp.csr_matrix(F), sp.identity(nc), format="csr")
    A.sort_indices()
    b = rng.standard_normal(nc * nodes)
    e, o = c.Engine(hip_lib, "BRM1", 0), oracle.Engine("BRM1")
    for eng in (e, o):
        eng.set_csr(c.MAT_AP, A)
    assert relerr(e.spmv_np(c.MAT_AP, b, nc * nodes), A @ b) < 1e-12
    for cfg in (("chebyshev", "jacobi", 4, 0.0, 0.3, 1.8),
                ("richardson", "jacobi", 3, 0.0)):
        e.set_inner(c.KSP_AP, *cfg)
        o.set_inner(c.KSP_AP, *cfg)
        assert relerr(e.inner_solve_np(c.KSP_AP, b),
                      o.inner_solve_np(c.KSP_AP, b)) < 1e-11, cfg


def _grid_operator(rng, shape, diag_links):
    """SPD-ish operator on a structured grid: nearest neighbours (and the
    diagonal ones of a P1 triangulation where asked), random positive
    weights, diagonally dominant."""
    n = int(np.prod(shape))
    idx = np.arange(n).reshape(shape)
    rows, cols = [], []
    dirs = []
    for ax in range(len(shape)):
        d = [0] * len(shape)
        d[ax] = 1
        dirs.append(tuple(d))
    if diag_links:
        dirs.append(tuple([1] * len(shape)))
    for d in dirs:
        sl_a = tuple(slice(0, s - dd) for s, dd in zip(shape, d))
        sl_b = tuple(slice(dd, s) for s, dd in zip(shape, d))
        rows.append(idx[sl_a].ravel())
        cols.append(idx[sl_b].ravel())
    r, cc = np.concatenate(rows), np.concatenate(cols)
    w = rng.random(r.size) + 0.1
    A = sp.coo_matrix((np.r_[-w, -w], (np.r_[r, cc], np.r_[cc, r])), shape=(n, n)).tocsr()
    A = A + sp.diags(-np.asarray(A.sum(axis=1)).ravel() + rng.random(n) + 0.5)
    A = sp.csr_matrix(A)
    A.sort_indices()
    return A


@pytest.mark.parametrize("shape,diag", [((150, 150), True), ((97, 61), False),
                                        ((14, 14, 14), False)])
def test_chebyshev_steps_in_one_launch(hip_lib, monkeypatch, shape, diag):
    """ChebPatch (k_cheb_patch): the m Chebyshev-Jacobi steps of a small scalar
    operator in ONE launch - graph clusters with their m-edge patches, the
    steps run in LDS - against the step-by-step path (PCD_CHEB_PATCH=0) and
    the oracle, m = 2 .. 8; a value update reaches the patch copies; an
    operator whose patches would not fit stays on the step-by-step path."""
    rng = np.random.default_rng(sum(shape))
    A = _grid_operator(rng, shape, diag)
    n = A.shape[0]
    b = rng.standard_normal(n)
    o = oracle.Engine("BRM1")
    o.set_csr(c.MAT_MP, A)
    engines = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("PCD_CHEB_PATCH", sw)
        e = c.Engine(hip_lib, "BRM1", 0)
        e.set_csr(c.MAT_MP, A)
        engines[sw] = e
    used = 0
    for m in (2, 3, 5, 8):
        cfg = ("chebyshev", "jacobi", m, 0.0, 0.4, 2.1)
        o.set_inner(c.KSP_MP, *cfg)
        ref = o.inner_solve_np(c.KSP_MP, b)
        out = {}
        for sw, e in engines.items():
            monkeypatch.setenv("PCD_CHEB_PATCH", sw)
            e.set_inner(c.KSP_MP, *cfg)
            e.inner_solve_np(c.KSP_MP, b)                       # (set-up on first use)
            l0 = e.info(c.INFO_LAUNCHES)
            out[sw] = e.inner_solve_np(c.KSP_MP, b)
            out[sw + "launches"] = e.info(c.INFO_LAUNCHES) - l0
        assert relerr(out["0"], ref) < 1e-11, m
        assert relerr(out["1"], ref) < 1e-11, m
        assert relerr(out["1"], out["0"]) < 1e-13, m
        if out["1launches"] < out["0launches"]:
            used += 1
            assert out["0launches"] - out["1launches"] == m - 1, (m, out)
    # the plane fits for every m here; in space the patches of many edges
    # outgrow the workgroup and the step-by-step path stays
    assert used >= (4 if len(shape) == 2 else 1), used
    # new values, same pattern
    A2 = A.copy()
    A2.data[:] = A.data * (1.0 + 0.1 * rng.random(A.nnz))
    A2 = sp.csr_matrix((A2 + A2.T) * 0.5)
    A2.sort_indices()
    assert np.array_equal(A2.indices, A.indices)
    cfg = ("chebyshev", "jacobi", 5, 0.0, 0.4, 2.1)
    o.set_csr(c.MAT_MP, A2)
    o.set_inner(c.KSP_MP, *cfg)
    ref = o.inner_solve_np(c.KSP_MP, b)
    for sw, e in engines.items():
        monkeypatch.setenv("PCD_CHEB_PATCH", sw)
        e.set_inner(c.KSP_MP, *cfg)
        e.update_values(c.MAT_MP, A2.data)
        assert relerr(e.inner_solve_np(c.KSP_MP, b), ref) < 1e-11, sw


def test_chebyshev_patch_follows_a_pattern_change(hip_lib, monkeypatch):
    """A second pcd_set_csr on the slot with ANOTHER pattern (fewer rows, fewer
    entries) drops the one-launch patch of the first (upload_csr releases it;
    round-5 advisor finding): the solve on the new operator equals the
    step-by-step path and the oracle, and is again one launch."""
    rng = np.random.default_rng(7)
    A1 = _grid_operator(rng, (120, 120), True)
    A2 = _grid_operator(rng, (61, 47), False)
    assert A2.nnz < A1.nnz and A2.shape[0] < A1.shape[0]
    cfg = ("chebyshev", "jacobi", 5, 0.0, 0.4, 2.1)
    engines = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("PCD_CHEB_PATCH", sw)
        e = c.Engine(hip_lib, "BRM1", 0)
        e.set_csr(c.MAT_MP, A1)
        e.set_inner(c.KSP_MP, *cfg)
        e.inner_solve_np(c.KSP_MP, rng.standard_normal(A1.shape[0]))   # builds the patch of A1
        engines[sw] = e
    b = rng.standard_normal(A2.shape[0])
    o = oracle.Engine("BRM1")
    o.set_csr(c.MAT_MP, A2)
    o.set_inner(c.KSP_MP, *cfg)
    ref = o.inner_solve_np(c.KSP_MP, b)
    out = {}
    for sw, e in engines.items():
        monkeypatch.setenv("PCD_CHEB_PATCH", sw)
        e.set_csr(c.MAT_MP, A2)
        e.set_inner(c.KSP_MP, *cfg)
        e.inner_solve_np(c.KSP_MP, b)
        l0 = e.info(c.INFO_LAUNCHES)
        out[sw] = e.inner_solve_np(c.KSP_MP, b)
        out[sw + "launches"] = e.info(c.INFO_LAUNCHES) - l0
    assert relerr(out["0"], ref) < 1e-11
    assert relerr(out["1"], ref) < 1e-11
    assert relerr(out["1"], out["0"]) < 1e-13
    assert out["0launches"] - out["1launches"] == 4, out
