"""Algebraic multigrid hierarchies (smoothed aggregation) for ``pc_type gamg``.

The reference's "iterative" configuration preconditions ``A00``, ``Ap`` (and
``Rp``) with hypre BoomerAMG cycles (``demo/navier-stokes-pcd/
demo_navier-stokes-pcd.py:153-160``): an ALGEBRAIC method, it needs nothing
but the matrix.  The engine's cycle (``pc_type mg``) takes any prolongation
chain; until now the only producer of chains was the nested-mesh hierarchy of
``fenapack_amd.fem``.  This module builds the chain from the matrix alone, so
matrices from any mesh or assembler get the multigrid path - the role hypre
plays for the reference.  The method is PETSc's other algebraic multigrid,
``-pc_type gamg`` [ext PETSc]: smoothed aggregation (Vanek, Mandel, Brezina
1996) with MIS-2 aggregation; it is NOT BoomerAMG's classical coarsening, and
iteration counts differ from both (measured in DESIGN.md).

Set-up runs on the host (scipy), once per sparsity pattern: aggregates and
prolongators are kept while only values change (``-pc_gamg_reuse_
interpolation``); the cycle itself - smoothing, restriction, coarse solve -
is the engine's, on the device, with Galerkin coarse operators.

For a velocity block ``F (x) I_d`` on interleaved dofs the aggregation works
on the scalar stencil ``F`` and the prolongator is ``P_F (x) I_d``: every
coarse level keeps the Kronecker structure the multi-component kernels use.
"""

import numpy as np
import scipy.sparse as sp

from . import _host


def _strength(A, theta):
    """Symmetric strength graph: |a_ij| >= theta sqrt(|a_ii a_jj|), no
    diagonal."""
    A = sp.csr_matrix(A)
    d = np.abs(A.diagonal())
    d[d == 0.0] = 1.0
    C = sp.coo_matrix(A)
    keep = (C.row != C.col) & (np.abs(C.data)
                               >= theta * np.sqrt(d[C.row] * d[C.col]))
    S = sp.csr_matrix((np.abs(C.data[keep]), (C.row[keep], C.col[keep])),
                      shape=A.shape)
    S = S.maximum(S.T).tocsr()
    S.sort_indices()
    return S


def _row_max(G, w):
    """max_j w[j] over the pattern of each row of ``G`` (-inf on empty rows)."""
    out = np.full(G.shape[0], -np.inf)
    nz = np.diff(G.indptr) > 0
    if G.nnz:
        red = np.maximum.reduceat(w[G.indices], G.indptr[:-1][nz])
        out[nz] = red
    return out


def _mis(G, seed=0):
    """Maximal independent set of the graph ``G`` (Luby, random priorities)."""
    n = G.shape[0]
    rng = np.random.default_rng(seed)
    # degree-biased priorities: low-degree vertices first gives larger sets
    w = rng.random(n) + 1.0 / (1.0 + np.diff(G.indptr))
    state = np.zeros(n, dtype=np.int8)            # 0 undecided, 1 in, -1 out
    while True:
        und = state == 0
        if not und.any():
            break
        wm = np.where(und, w, -np.inf)
        nb = _row_max(G, wm)
        win = und & (wm > nb)
        if not win.any():                         # ties: break by index
            idx = np.nonzero(und)[0]
            win = np.zeros(n, bool)
            win[idx[np.argmax(w[idx])]] = True
        state[win] = 1
        hit = _row_max(G, np.where(win, 1.0, -np.inf)) > 0
        state[hit & (state == 0)] = -1
    return state == 1


def aggregate(S, seed=0, distance=2):
    """MIS-2 aggregation of the strength graph ``S``: roots no closer than
    three edges, distance-1 vertices join their root, distance-2 vertices the
    aggregate of their strongest aggregated neighbour.  Isolated vertices -
    Dirichlet rows: nothing couples to them, one Jacobi step solves them - get
    NO coarse representative (``agg = -1``, a zero row of the prolongator);
    kept as singleton aggregates they would survive on every level (the 3-D
    cavity has 6 N^2 of them: the hierarchy of cube N = 36 stalled at 58 k
    rows, profiles/r03_u_gamg_cube_n36_stalled.txt).
    Returns ``agg`` (n,) with values in ``[-1, nagg)``."""
    n = S.shape[0]
    pat = sp.csr_matrix((np.ones(S.nnz), S.indices, S.indptr), shape=S.shape)
    if distance >= 2:
        G2 = (pat @ pat + pat).tocsr()
        G2.setdiag(0)
        G2.eliminate_zeros()
        G2.sort_indices()
    else:
        G2 = pat          # MIS-1: roots two edges apart, aggregates = stars
    isolated = np.diff(S.indptr) == 0
    roots = _mis(G2, seed) & ~isolated
    agg = np.full(n, -1, dtype=np.int64)
    ridx = np.nonzero(roots)[0]
    agg[ridx] = np.arange(ridx.size)
    # distance 1: strongest root neighbour
    for _ in range(2):
        C = sp.coo_matrix(S)
        ok = (agg[C.row] < 0) & (agg[C.col] >= 0)
        if not ok.any():
            break
        r, c, v = C.row[ok], C.col[ok], C.data[ok]
        order = np.lexsort((v, r))                # last per row = strongest
        r, c = r[order], c[order]
        last = np.r_[r[1:] != r[:-1], True]
        new = agg.copy()
        new[r[last]] = agg[c[last]]
        agg = new
    # connected vertices the two sweeps did not reach (farther than two edges
    # from every root cannot happen for a maximal set; kept for safety)
    left = np.nonzero((agg < 0) & ~isolated)[0]
    if left.size:
        agg[left] = ridx.size + np.arange(left.size)
    return agg, int(agg.max()) + 1 if agg.size and agg.max() >= 0 else 0


def _rho(A, iters=15, seed=0):
    """Spectral radius estimate of D^-1 A (power iteration, envelope)."""
    d = A.diagonal().copy()
    d[d == 0.0] = 1.0
    v = np.random.default_rng(seed).standard_normal(A.shape[0])
    best = 0.0
    for k in range(iters):
        v /= np.linalg.norm(v)
        v = (A @ v) / d
        if k >= 2:
            best = max(best, np.linalg.norm(v))
    return best


def _tentative(agg, nagg):
    """Piecewise-constant prolongator of the aggregates, columns normalised;
    vertices without an aggregate (``agg < 0``) get a zero row."""
    has = np.nonzero(agg >= 0)[0]
    size = np.bincount(agg[has], minlength=nagg).astype(float)
    return sp.csr_matrix((1.0 / np.sqrt(size[agg[has]]), (has, agg[has])),
                         shape=(agg.size, nagg))


def sa_prolongator(A, theta=0.02, omega=4.0 / 3.0, seed=0, distance=2):
    """One smoothed-aggregation prolongator for the scalar operator ``A``:
    ``P = (I - omega/rho D^-1 A) T`` with the piecewise-constant tentative
    prolongator ``T`` of the aggregates (columns normalised)."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    agg, nagg = aggregate(_strength(A, theta), seed, distance)
    T = _tentative(agg, nagg)
    d = A.diagonal().copy()
    d[d == 0.0] = 1.0
    rho = _rho(A)
    P = T - (omega / rho) * (sp.diags(1.0 / d) @ (A @ T))
    P = sp.csr_matrix(P)
    P.eliminate_zeros()
    P.sort_indices()
    return P


def scalar_stencil(A, block):
    """``F`` if ``A = F (x) I_block`` on interleaved dofs (pattern AND values),
    else ``None``."""
    if block <= 1:
        return sp.csr_matrix(A)
    A = sp.csr_matrix(A)
    if A.shape[0] % block:
        return None
    F = getattr(A, "kron_scalar", None)
    if F is not None and getattr(A, "kron_block", 0) == block:
        return F
    return _host.kron_factor(A, block)


def _galerkin(A, P):
    """``P^T A P`` with sorted columns; large products on the threaded native
    SpGEMM (the same sums in the same order as scipy's)."""
    if A.nnz > 400000 and not _host.use_numpy():
        return _host.spgemm(_host.spgemm(_host.transpose(P), A), P)
    C = (P.T @ A @ P).tocsr()
    C.sort_indices()
    return C


def block_graph_operator(A, block):
    """Scalar operator on the NODES of a general block matrix (Newton velocity
    block): entry (I, J) = Frobenius norm of the block, diagonal kept positive
    - only used to choose aggregates."""
    A = sp.coo_matrix(A)
    n = A.shape[0] // block
    G = sp.csr_matrix((A.data ** 2, (A.row // block, A.col // block)),
                      shape=(n, n))
    G.data = np.sqrt(G.data)
    return G


def smoothed_aggregation_chain(A, block=1, coarse_rows=2000, max_levels=12,
                               theta=0.02, min_ratio=1.5, distance=2):
    """Prolongation chain ``[None, P_1, ..., P_L]`` for the finest operator
    ``A`` (``P_l`` maps level ``l-1`` to ``l``; the format
    ``PC.setMGInterpolations`` takes).  Coarsening stops at ``coarse_rows``
    rows (explicit inverse there) or when it stalls."""
    A = sp.csr_matrix(A)
    Ps = []
    cur = A
    curF = scalar_stencil(cur, block) if block > 1 else None
    while cur.shape[0] > coarse_rows and len(Ps) < max_levels - 1:
        if block == 1 or curF is not None:
            # scalar operator, or F (x) I_d: everything on the scalar factor,
            # expanded (with its factor attached) for the hand-over
            F = cur if block == 1 else curF
            Pf = sa_prolongator(F, theta, seed=len(Ps), distance=distance)
            if Pf.shape[1] * min_ratio > Pf.shape[0]:
                break                              # coarsening stalled
            Fc = _galerkin(F, Pf)
            if block == 1:
                P, cur = Pf, Fc
            else:
                P = _host.kron_expand(Pf, block)
                curF, cur = Fc, _host.kron_expand(Fc, block)
            Ps.append(P)
            continue
        # coupled block: aggregate nodes on the block-norm graph, smooth
        # the tentative prolongator with the true operator
        G = block_graph_operator(cur, block)
        agg, nagg = aggregate(_strength(G, theta), len(Ps), distance)
        T = sp.kron(_tentative(agg, nagg), sp.identity(block),
                    format="csr")
        d = cur.diagonal().copy()
        d[d == 0.0] = 1.0
        P = T - (4.0 / 3.0 / _rho(cur)) * (sp.diags(1.0 / d) @ (cur @ T))
        P = sp.csr_matrix(P)
        P.sort_indices()
        if P.shape[1] * min_ratio > P.shape[0]:
            break                                  # coarsening stalled
        Ps.append(P)
        cur = _galerkin(cur, P)
    return [None] + Ps[::-1]
