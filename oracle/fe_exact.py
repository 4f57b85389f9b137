"""TEST INFRASTRUCTURE ONLY - exact element matrices of the fixed P2/P1 forms.

An independent restatement of the element integrals the producers evaluate
(host: ``fenapack_amd/fem/taylor_hood.py`` with quadrature; device:
``fenapack_amd/csrc/pcd_fe.hpp``): every integrand is expanded into monomials
of the barycentric coordinates and integrated in closed form,

    int_K lambda^alpha dx = |K| d! prod(alpha_i!) / (d + |alpha|)!,

so no quadrature rule, basis table or tensor contraction is shared with the
code under test.  Forms (``demo/navier-stokes-pcd/demo_navier-stokes-pcd.py:
104-137``): mass ``(u, v)``, stiffness ``(grad u, grad v)``, convection
``((w.grad) u, v)`` with a P2 wind, pressure convection ``(w.grad p, q)``,
streamline diffusion ``(w.grad u, w.grad v)`` (``:122-125``).

The reference delegates these integrals to DOLFIN/FFC, which is absent here:
parity of the producers is therefore pinned by this closed form, not by the
reference ("parity unpinned" at the DOLFIN boundary).  Pure-Python loops over
monomials: use on a handful of cells only.
"""

from math import factorial

import numpy as np


# ---- polynomials in the barycentric coordinates: {exponent tuple: coeff} ----
def _mono(nv, i=None, power=1):
    e = [0] * nv
    if i is not None:
        e[i] = power
    return {tuple(e): 1.0}


def _add(p, q, a=1.0, b=1.0):
    out = {}
    for k, v in p.items():
        out[k] = out.get(k, 0.0) + a * v
    for k, v in q.items():
        out[k] = out.get(k, 0.0) + b * v
    return out


def _mul(p, q):
    out = {}
    for k1, v1 in p.items():
        for k2, v2 in q.items():
            k = tuple(a + b for a, b in zip(k1, k2))
            out[k] = out.get(k, 0.0) + v1 * v2
    return out


def _scale(p, a):
    return {k: a * v for k, v in p.items()}


def _diff(p, i):
    """d p / d lambda_i, the lambdas taken as independent variables."""
    out = {}
    for k, v in p.items():
        if k[i]:
            e = list(k)
            e[i] -= 1
            out[tuple(e)] = out.get(tuple(e), 0.0) + v * k[i]
    return out


def _integrate(p, measure):
    d = len(next(iter(p))) - 1 if p else 0
    s = 0.0
    for k, v in p.items():
        num = factorial(d)
        for a in k:
            num *= factorial(a)
        s += v * num / factorial(d + sum(k))
    return measure * s


def p2_basis(nv, edges):
    """Vertex functions lambda_i (2 lambda_i - 1), edge functions
    4 lambda_i lambda_j in the order of ``edges``."""
    phi = []
    for i in range(nv):
        li = _mono(nv, i)
        phi.append(_add(_scale(_mul(li, li), 2.0), li, 1.0, -1.0))
    for i, j in edges:
        phi.append(_scale(_mul(_mono(nv, i), _mono(nv, j)), 4.0))
    return phi


def _grad_dot(p, vec_of_lambda):
    """(vec . grad) p for a vector field given through its components along
    grad lambda_k: sum_k d p/d lambda_k * c_k, c_k polynomials."""
    out = {}
    for k, ck in enumerate(vec_of_lambda):
        out = _add(out, _mul(_diff(p, k), ck))
    return out


def element_matrices(vertices, edges, U=None, nu=1.0):
    """Exact element matrices of one simplex.

    ``vertices``: (d+1, d) coordinates; ``edges``: local edge list;
    ``U``: (na, d) nodal values of the P2 wind (optional).
    Returns a dict with ``mass``, ``stiffness`` (na x na); with a wind also
    ``convection`` (na x na), ``supg`` (na x na, *without* the factor
    delta), ``kp`` ((d+1) x (d+1), includes 1/nu)."""
    X = np.asarray(vertices, dtype=float)
    nv, d = X.shape
    T = (X[1:] - X[0]).T                       # x = x0 + T lam_{1..d}
    measure = abs(np.linalg.det(T)) / factorial(d)
    Tinv = np.linalg.inv(T)
    g = np.empty((nv, d))                      # grad lambda_k
    g[1:] = Tinv
    g[0] = -Tinv.sum(axis=0)
    phi = p2_basis(nv, edges)
    na = len(phi)
    out = {"measure": measure, "gradlam": g}
    M = np.empty((na, na))
    K = np.empty((na, na))
    dphi = [[_diff(p, k) for k in range(nv)] for p in phi]
    for a in range(na):
        for b in range(na):
            M[a, b] = _integrate(_mul(phi[a], phi[b]), measure)
            s = 0.0
            for k in range(nv):
                for l in range(nv):
                    s += float(g[k] @ g[l]) * _integrate(
                        _mul(dphi[a][k], dphi[b][l]), measure)
            K[a, b] = s
    out["mass"], out["stiffness"] = M, K
    # pressure space (P1 = the barycentric coordinates) and the coupling
    # -(p, div v): div[a, j, c] = - int lambda_j d phi_a / d x_c
    lam = [_mono(nv, i) for i in range(nv)]
    out["mass_p1"] = np.array([[_integrate(_mul(lam[i], lam[j]), measure)
                                for j in range(nv)] for i in range(nv)])
    out["stiffness_p1"] = measure * (g @ g.T)
    D = np.empty((na, nv, d))
    for a in range(na):
        for j in range(nv):
            v = np.zeros(d)
            for k in range(nv):
                v += g[k] * _integrate(_mul(lam[j], dphi[a][k]), measure)
            D[a, j] = -v
    out["div"] = D
    if U is None:
        return out
    U = np.asarray(U, dtype=float)
    # w . grad lambda_k as a polynomial: sum_a phi_a (U_a . g_k)
    wl = []
    for k in range(nv):
        p = {}
        for a in range(na):
            p = _add(p, _scale(phi[a], float(U[a] @ g[k])))
        wl.append(p)
    wgrad = [_grad_dot(p, wl) for p in phi]    # (w . grad) phi_b
    C = np.empty((na, na))
    S = np.empty((na, na))
    for a in range(na):
        for b in range(na):
            C[a, b] = _integrate(_mul(phi[a], wgrad[b]), measure)
            S[a, b] = _integrate(_mul(wgrad[a], wgrad[b]), measure)
    out["convection"], out["supg"] = C, S
    Kp = np.empty((nv, nv))
    for i in range(nv):
        for j in range(nv):
            Kp[i, j] = _integrate(_mul(_mono(nv, i), wl[j]), measure) / nu
    out["kp"] = Kp
    return out
