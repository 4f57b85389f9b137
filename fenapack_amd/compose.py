"""Pre-composed operators for the latency-bound parts of a PCApply.

On an MI355X a dependent kernel on an operator of <= ~10^5 rows costs about
5 us whatever it computes (launch boundary + three dependent memory round
trips), and a PCApply at the benchmark size spends half of its
time in ~35 such launches.  Every inner solver on this path that runs a FIXED
number of steps is a fixed linear operator, so adjacent steps can be
multiplied out once on the host and applied as ONE sparse product:

* Chebyshev(k) + Jacobi from a zero guess (the ``M_p`` solve of
  ``demo_navier-stokes-pcd.py:161-165``) is ``x = q(D^-1 A) D^-1 b`` with a
  polynomial ``q`` of degree ``k``: factored into <= 2 sparse factors;
* one level of a multigrid V(nu1, nu2) cycle is two affine maps around the
  coarse solve:
    down:  x1 = H1 b (the ordinary smoothing kernels),
           r_c = W_d b                W_d = R (I - A H1)
    up:    x = W_u [x1; r_c; e_c; b]  W_u = [G2 | 0 | G2 P | H2]
  where ``p_k = G x_in + H b`` is what ``nu`` smoothing steps do.

The matrices are built by running the engine's own recurrences (same
coefficients, same order: ``mg_smooth`` / ``solve_cheb`` in
``csrc/pcd_apply.hip``) on sparse matrices instead of vectors, so the
composed operator equals the step-by-step one in exact arithmetic; in floating
point they differ by reassociation only (tests compare both engine paths and
the oracle at 1e-11).  More bytes per launch, far fewer launches: worth it
exactly where a launch is latency-bound, which is why the finest velocity
levels keep their step-by-step kernels.
"""

import numpy as np
import scipy.sparse as sp


def _diag_inv(A):
    d = A.diagonal().copy()
    d[d == 0.0] = 1.0
    return sp.diags(1.0 / d, format="csr")


def _cheb_coeffs(emin, emax):
    """(scale, mu, omegaprod) of [ext PETSc] KSPCHEBYSHEV as the engine
    computes them."""
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    return scale, 1.0 / alpha, 2.0 / alpha


def smoother_maps(A, emin, emax, nu, zero_guess):
    """(G, H) with ``x_out = G x_in + H b`` for ``nu`` Chebyshev-Jacobi steps
    of ``mg_smooth`` (zero guess: ``G`` is None)."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    Dinv = _diag_inv(A)
    DA = (Dinv @ A).tocsr()
    scale, mu, omegaprod = _cheb_coeffs(emin, emax)
    c_km1, c_k = 1.0, mu
    if nu == 0:
        return (None if zero_guess else I), sp.csr_matrix((n, n))
    # (G, H) of p_k and of p_{k-1}.  Both starts are followed by nu - 1
    # three-term updates whose coefficients begin at (c_{k-1}, c_k) = (1, mu)
    if zero_guess:
        Gm = Gk = Hm = None
        Hk = (scale * Dinv).tocsr()             # p0 = scale D^-1 b
        have_pm = False                         # p_{-1} = 0
    else:
        Gm, Hm = I, sp.csr_matrix((n, n))       # the guess itself
        Gk = (I - scale * DA).tocsr()           # p1 = p0 + scale D^-1 (b - A p0)
        Hk = (scale * Dinv).tocsr()
        have_pm = True
    for _ in range(nu - 1):
        c_kp1 = 2.0 * mu * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        c0 = (1.0 - omega) if have_pm else 0.0
        Hn = omega * Hk + omega * scale * (Dinv - DA @ Hk)
        if have_pm:
            Hn = Hn + c0 * Hm
        Gn = None
        if Gk is not None:
            Gn = omega * Gk - omega * scale * (DA @ Gk) + c0 * Gm
            Gn = Gn.tocsr()
        Gm, Hm, Gk, Hk = Gk, Hk, Gn, Hn.tocsr()
        c_km1, c_k = c_k, c_kp1
        have_pm = True
    return Gk, Hk


def _clean(M):
    M = sp.csr_matrix(M)
    M.sum_duplicates()
    M.sort_indices()
    return M


def vcycle_level(A, P, emin, emax, nu_pre, nu_post):
    """(W_d, W_u) of one multigrid level: operator ``A`` (n x n), prolongation
    ``P`` (n x nc), restriction ``P^T``.

    ``W_d = R (I - A H1)`` is nc x n: the coarse right-hand side straight from
    ``b`` (the pre-smoothed ``x1 = H1 b`` itself comes from the ordinary
    smoothing kernels);  ``W_u`` is n x (n + 2 nc + n) over the stacked input
    ``[x1 (n); r_c (nc, unused); e_c (nc); b (n)]``."""
    A, P = sp.csr_matrix(A), sp.csr_matrix(P)
    n, nc = P.shape
    if nu_pre < 1 or nu_post < 1:
        raise ValueError("fused levels need nu_pre, nu_post >= 1")
    I = sp.identity(n, format="csr")
    _, H1 = smoother_maps(A, emin, emax, nu_pre, True)
    R = P.T.tocsr()
    Wd = (R @ (I - A @ H1)).tocsr()
    G2, H2 = smoother_maps(A, emin, emax, nu_post, False)
    Wu = sp.hstack([G2, sp.csr_matrix((n, nc)), G2 @ P, H2], format="csr")
    return _clean(Wd), _clean(Wu)


def chebyshev_factors(A, emin, emax, its, out_scale=1.0, max_factors=2):
    """Chebyshev(its) + Jacobi from a zero guess (``solve_cheb``) as a product
    of sparse factors: ``x = W_{m-1} ... W_0 b``.

    The solution operator is ``q(B) D^-1`` with ``B = D^-1 A`` and ``deg q =
    its``; ``q`` is expanded in monomials by running the three-term recurrence
    on polynomial coefficients and split at its (complex-conjugate pairs of)
    roots into ``max_factors`` real factors of about equal degree, so that no
    factor is denser than ``A^(ceil(its / max_factors))`` (+ 1 for odd
    splits)."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    Dinv = _diag_inv(A)
    B = (Dinv @ A).tocsr()
    scale, mu, omegaprod = _cheb_coeffs(emin, emax)
    # p_k = q_k(B) D^-1 b: coefficient arrays in ascending powers of B
    qm = np.zeros(1)
    qk = np.array([scale])                         # p0 = scale D^-1 b
    c_km1, c_k = 1.0, mu
    for it in range(its):
        c_kp1 = 2.0 * mu * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        c0 = 0.0 if it == 0 else 1.0 - omega
        # p_{k+1} = c0 p_{k-1} + omega p_k + omega scale (D^-1 b - B p_k)
        qn = np.zeros(qk.size + 1)
        qn[:qk.size] += omega * qk
        qn[1:] -= omega * scale * qk
        qn[0] += omega * scale
        qn[:qm.size] += c0 * qm
        qm, qk = qk, qn
        c_km1, c_k = c_k, c_kp1
    q = qk * out_scale
    deg = q.size - 1
    I = sp.identity(n, format="csr")
    if deg == 0:
        return [_clean(q[0] * Dinv)]
    roots = np.roots(q[::-1])
    # group the roots: conjugate pairs stay together
    roots = sorted(roots, key=lambda z: (abs(z.imag) < 1e-14 * abs(z), z.real,
                                         abs(z.imag)))
    groups, used = [], [False] * len(roots)
    for i, z in enumerate(roots):
        if used[i]:
            continue
        used[i] = True
        if abs(z.imag) > 1e-12 * max(abs(z), 1e-300):
            j = min((k for k in range(len(roots)) if not used[k]),
                    key=lambda k: abs(roots[k] - np.conj(z)))
            used[j] = True
            groups.append(np.poly([z, roots[j]]).real[::-1])   # ascending
        else:
            groups.append(np.array([-z.real, 1.0]))
    nf = max(1, min(max_factors, len(groups)))
    # distribute the groups over nf factors with about equal degree
    groups.sort(key=lambda g: -g.size)
    buckets = [[] for _ in range(nf)]
    degs = [0] * nf
    for g in groups:
        k = int(np.argmin(degs))
        buckets[k].append(g)
        degs[k] += g.size - 1
    lead = q[-1]

    def poly_matrix(coeffs):                       # Horner in B
        M = coeffs[-1] * I
        for cf in coeffs[-2::-1]:
            M = (B @ M + cf * I).tocsr()
        return M

    factors = []
    for k, bucket in enumerate(buckets):
        c = np.array([1.0])
        for g in bucket:
            c = np.convolve(c, g)
        factors.append(poly_matrix(c))
    # x = lead * prod(factors) D^-1 b; the scalar and D^-1 ride on the ends
    factors[0] = factors[0] @ Dinv
    factors[-1] = lead * factors[-1]
    return [_clean(F) for F in factors]
