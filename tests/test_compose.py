"""CPU suite: the host composition of inner solves (fenapack_amd/compose.py)
against plain numpy restatements of the engine's recurrences (mg_smooth /
solve_cheb / mg_vcycle in csrc/pcd_apply.hip).  The GPU suite
(test_precomposed_gpu.py) then checks the engine's use of these operators
against its own step-by-step path and the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from fenapack_amd.compose import (chebyshev_factors, smoother_maps,
                                  vcycle_level)
from fenapack_amd.petsc import estimate_emax
from helpers import flow_state


def smooth(A, emin, emax, nu, b, x0):
    """mg_smooth: nu Chebyshev-Jacobi steps (x0 None = zero guess)."""
    dinv = 1.0 / A.diagonal()
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    mu, op = 1.0 / alpha, 2.0 / alpha
    ckm1, ck = 1.0, mu
    if x0 is None:
        pk, pm = scale * dinv * b, None
    else:
        pm, pk = x0, x0 + scale * dinv * (b - A @ x0)
    for _ in range(nu - 1):
        ckp1 = 2.0 * mu * ck - ckm1
        om = op * ck / ckp1
        pn = om * pk + om * scale * dinv * (b - A @ pk)
        if pm is not None:
            pn = pn + (1.0 - om) * pm
        pm, pk, ckm1, ck = pk, pn, ck, ckp1
    return pk


def cheb(A, lo, hi, its, b, f=1.0):
    """solve_cheb: zero guess, its three-term updates, final scale f."""
    dinv = 1.0 / A.diagonal()
    scale = 2.0 / (hi + lo)
    alpha = 1.0 - scale * lo
    mu, op = 1.0 / alpha, 2.0 / alpha
    ckm1, ck = 1.0, mu
    pk, pm = scale * dinv * b, None
    for it in range(its):
        ckp1 = 2.0 * mu * ck - ckm1
        om = op * ck / ckp1
        pn = om * pk + om * scale * dinv * (b - A @ pk)
        if it > 0:
            pn = pn + (1.0 - om) * pm
        pm, pk, ckm1, ck = pk, pn, ck, ckp1
    return f * pk


@pytest.fixture(scope="module")
def state():
    st = flow_state("cavity", 2)
    A = sp.csr_matrix(st["L"]["A00"])
    emax = 1.1 * estimate_emax(A, iters=12)
    return st, A, 0.1 * emax, emax


@pytest.mark.parametrize("nu", [1, 2, 3, 4])
def test_smoother_maps_are_the_recurrences(state, nu):
    st, A, emin, emax = state
    rng = np.random.default_rng(nu)
    b, x0 = rng.standard_normal(A.shape[0]), rng.standard_normal(A.shape[0])
    G, H = smoother_maps(A, emin, emax, nu, True)
    assert G is None
    ref = smooth(A, emin, emax, nu, b, None)
    assert np.abs(H @ b - ref).max() <= 1e-12 * np.abs(ref).max()
    G, H = smoother_maps(A, emin, emax, nu, False)
    ref = smooth(A, emin, emax, nu, b, x0)
    assert np.abs(G @ x0 + H @ b - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("nu_pre,nu_post", [(2, 2), (1, 2), (3, 1)])
def test_vcycle_level_products(state, nu_pre, nu_post):
    st, A, emin, emax = state
    P = st["pb"].interpolations().chain("u")[-1]
    n, nc = P.shape
    Wd, Wu = vcycle_level(A, P, emin, emax, nu_pre, nu_post)
    assert Wd.shape == (nc, n) and Wu.shape == (n, 2 * n + 2 * nc)
    rng = np.random.default_rng(7)
    b, e = rng.standard_normal(n), rng.standard_normal(nc)
    x1 = smooth(A, emin, emax, nu_pre, b, None)
    rc = P.T @ (b - A @ x1)
    assert np.abs(Wd @ b - rc).max() <= 1e-12 * np.abs(rc).max()
    ref = smooth(A, emin, emax, nu_post, b, x1 + P @ e)
    got = Wu @ np.concatenate([x1, rng.standard_normal(nc), e, b])
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    assert abs(Wu[:, n:n + nc]).sum() == 0.0        # the r_c piece is unused
    with pytest.raises(ValueError):
        vcycle_level(A, P, emin, emax, 0, 2)


@pytest.mark.parametrize("its,nf", [(1, 2), (2, 2), (3, 2), (5, 2), (5, 1),
                                    (6, 3), (8, 2)])
def test_chebyshev_factors_are_the_solve(state, its, nf):
    st = state[0]
    Mp = sp.csr_matrix(st["pb"].Mp)
    b = np.random.default_rng(its).standard_normal(Mp.shape[0])
    F = chebyshev_factors(Mp, 0.5, 2.0, its, -1.0, max_factors=nf)
    assert 1 <= len(F) <= nf
    y = b
    for W in F:
        y = W @ y
    ref = cheb(Mp, 0.5, 2.0, its, b, -1.0)
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
