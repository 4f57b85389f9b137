"""numpy/scipy restatement of the inner Krylov methods and the outer solver.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Independent of
``pcd_oracle.c`` on purpose: the golden generator plugs these functions in as
the ``KSP.solve`` fakes under the reference's own ``apply`` bodies, and
``tests/test_oracle.py`` checks the C oracle against the resulting vectors.

[ext PETSc] The algorithms are PETSc's (absent from /root/reference, version
unpinned by it): KSPCG + PCJACOBI with the natural norm, KSPCHEBYSHEV with
user eigenvalue bounds, KSPRICHARDSON (scale 1), KSPGMRES (classical
Gram-Schmidt, right preconditioning), PCFIELDSPLIT Schur/upper.
"""

import numpy as np
import scipy.sparse.linalg as spla


def jacobi(A, pc):
    if pc == "jacobi":
        d = A.diagonal().copy()
        d[d == 0.0] = 1.0
        return 1.0 / d
    return np.ones(A.shape[0])


def cg(A, b, max_it, rtol=0.0, pc="jacobi"):
    dinv = jacobi(A, pc)
    x = np.zeros_like(b)
    r = b.copy()
    z = dinv * r
    rz = rz0 = float(r @ z)
    p = None
    it = 0
    while it < max_it:
        if rz == 0.0:
            break
        if rtol > 0.0 and np.sqrt(abs(rz)) <= rtol * np.sqrt(abs(rz0)):
            break
        p = z.copy() if it == 0 else z + (rz / rz_old) * p
        q = A @ p
        alpha = rz / float(p @ q)
        x += alpha * p
        r -= alpha * q
        z = dinv * r
        rz_old, rz = rz, float(r @ z)
        it += 1
    return x, it


def chebyshev(A, b, max_it, emin, emax, pc="jacobi"):
    dinv = jacobi(A, pc)
    scale = 2.0 / (emax + emin)
    alpha = 1.0 - scale * emin
    mu, omegaprod = 1.0 / alpha, 2.0 / alpha
    c_km1, c_k = 1.0, mu
    pkm1 = np.zeros_like(b)
    pk = scale * (dinv * b) + pkm1
    for _ in range(max_it):
        c_kp1 = 2.0 * mu * c_k - c_km1
        omega = omegaprod * c_k / c_kp1
        r = b - A @ pk
        pkp1 = (1.0 - omega) * pkm1 + omega * pk + omega * scale * (dinv * r)
        pkm1, pk = pk, pkp1
        c_km1, c_k = c_k, c_kp1
    return pk, max_it


def richardson(A, b, max_it, pc="jacobi"):
    dinv = jacobi(A, pc)
    x = np.zeros_like(b)
    for it in range(max_it):
        r = b.copy() if it == 0 else b - A @ x
        x += dinv * r
    return x, max_it


def make_inner(A, cfg):
    """``cfg``: ("direct",) | ("cg", max_it, rtol) | ("chebyshev", max_it,
    emin, emax) | ("richardson", max_it) | ("preonly",) -> callable b -> x."""
    kind = cfg[0]
    if kind == "direct":
        lu = spla.splu(A.tocsc())
        return lambda b: lu.solve(b)
    if kind == "cg":
        return lambda b: cg(A, b, cfg[1], cfg[2])[0]
    if kind == "chebyshev":
        return lambda b: chebyshev(A, b, cfg[1], cfg[2], cfg[3])[0]
    if kind == "richardson":
        return lambda b: richardson(A, b, cfg[1])[0]
    if kind == "preonly":
        return lambda b: jacobi(A, "jacobi") * b
    raise ValueError(kind)


def pcd_apply(variant, x, Ap, Mp, Kp, bc_idx, bc_val, solve_Ap, solve_Mp,
              solve_Rp=None):
    """Hand restatement of preconditioners.py:124-135,158-169,239-252,285-298
    (used to cross-check the goldens produced by the reference's own code)."""
    if variant in ("BRM1", "RBRM1"):
        z = x.copy()
        z[bc_idx] = bc_val
        y = solve_Ap(z)
        z = Kp @ y
        z = z + 1.0 * x
        y = solve_Mp(z)
        if variant == "RBRM1":
            y = y + 1.0 * solve_Rp(x)
        return -1.0 * y
    y = solve_Mp(x)
    z0 = y.copy()
    z1 = Kp @ z0
    z1[bc_idx] = bc_val
    z0 = solve_Ap(z1)
    y = y + 1.0 * z0
    if variant == "RBRM2":
        y = y + 1.0 * solve_Rp(x)
    return -1.0 * y


def fieldsplit_upper(xu, xp, pcd, A01, solve_A00):
    """[ext PETSc] PCApply_FieldSplit_Schur, UPPER factorisation."""
    yp = pcd(xp)
    yu = solve_A00(xu - A01 @ yp)
    return yu, yp


def gmres_right(A, b, M, rtol=1e-6, atol=0.0, restart=150, max_it=10000):
    """[ext PETSc] right-preconditioned restarted GMRES, classical GS."""
    n = b.size
    x = np.zeros(n)
    bnorm = np.linalg.norm(b)
    tol = max(rtol * bnorm, atol)
    r = b.copy()
    it, res = 0, bnorm
    hist = [res]
    while it < max_it and res > tol:
        beta = np.linalg.norm(r)
        res = beta
        if beta <= tol:
            break
        m = restart
        V = np.zeros((m + 1, n))
        H = np.zeros((m + 1, m))
        cs, sn, g = np.zeros(m), np.zeros(m), np.zeros(m + 1)
        V[0] = r / beta
        g[0] = beta
        k = 0
        while k < m and it < max_it:
            w = A @ M(V[k])
            h = V[:k + 1] @ w
            w = w - V[:k + 1].T @ h
            hn = np.linalg.norm(w)
            H[:k + 1, k] = h
            H[k + 1, k] = hn
            if hn != 0.0:
                V[k + 1] = w / hn
            for j in range(k):
                t = cs[j] * H[j, k] + sn[j] * H[j + 1, k]
                H[j + 1, k] = -sn[j] * H[j, k] + cs[j] * H[j + 1, k]
                H[j, k] = t
            d = np.hypot(H[k, k], H[k + 1, k])
            cs[k], sn[k] = H[k, k] / d, H[k + 1, k] / d
            H[k, k], H[k + 1, k] = d, 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            res = abs(g[k + 1])
            it += 1
            hist.append(res)
            k += 1
            if res <= tol or hn == 0.0:
                break
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k])
        x = x + M(V[:k].T @ y)
        if res <= tol or it >= max_it:
            break
        r = b - A @ x
    return x, it, res, hist


# --------------------------------------------------------------- multigrid
class Multigrid(object):
    """[ext PETSc] PCMG-style geometric V-cycle, multiplicative, Chebyshev +
    Jacobi smoothing with given eigenvalue bounds per level, explicit coarse
    inverse on level 0.  ``ops``: operators coarsest..finest; ``chain[l]``:
    prolongation level l-1 -> l (``chain[0] is None``); ``bounds[l]`` =
    (emin, emax) of D^-1 A on level l >= 1."""

    def __init__(self, ops, chain, bounds, nu_pre=2, nu_post=2,
                 coarse_inverse=None):
        self.ops, self.chain, self.bounds = ops, chain, bounds
        self.nu_pre, self.nu_post = nu_pre, nu_post
        self.dinv = [None if A is None else jacobi(A, "jacobi") for A in ops]
        self.C = (np.linalg.inv(ops[0].toarray()) if coarse_inverse is None
                  else coarse_inverse)

    def smooth(self, l, b, x, nu):
        if nu == 0:
            return np.zeros_like(b) if x is None else x
        A, dinv = self.ops[l], self.dinv[l]
        emin, emax = self.bounds[l]
        scale = 2.0 / (emax + emin)
        alpha = 1.0 - scale * emin
        mu, omegaprod = 1.0 / alpha, 2.0 / alpha
        c_km1, c_k = 1.0, mu
        if x is None:
            pkm1 = np.zeros_like(b)
            pk = scale * (dinv * b)
        else:
            pkm1 = x
            pk = x + scale * (dinv * (b - A @ x))
        for _ in range(nu - 1):
            c_kp1 = 2.0 * mu * c_k - c_km1
            omega = omegaprod * c_k / c_kp1
            pkp1 = ((1.0 - omega) * pkm1 + omega * pk
                    + omega * scale * (dinv * (b - A @ pk)))
            pkm1, pk = pk, pkp1
            c_km1, c_k = c_k, c_kp1
        return pk

    def vcycle(self, l, b):
        if l == 0:
            return self.C @ b
        x = self.smooth(l, b, None, self.nu_pre)
        r = b - self.ops[l] @ x if self.nu_pre else b
        e = self.vcycle(l - 1, self.chain[l].T @ r)
        x = x + self.chain[l] @ e
        return self.smooth(l, b, x, self.nu_post)

    def apply(self, b):
        return self.vcycle(len(self.ops) - 1, b)

    def richardson(self, b, its):
        A = self.ops[-1]
        x = np.zeros_like(b)
        for it in range(its):
            r = b if it == 0 else b - A @ x
            x = x + self.apply(r)
        return x
