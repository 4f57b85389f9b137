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

import os

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
    isolated = np.diff(S.indptr) == 0
    G2 = None
    if distance >= 2 and not _host.use_numpy() and n < 2 ** 31 \
            and os.environ.get("FENAPACK_AMD_MIS2_EXPLICIT", "0") != "1":
        # the distance-2 graph is never formed: its row lengths (the degree
        # term of Luby's priorities) and the rounds themselves walk two hops
        # of `pat` (libpcd_host pcdh_mis2*: cube N = 73's 5.4e8-entry square -
        # half a minute and tens of GB - is gone; the same set as _mis on the
        # explicit graph, tests/test_host_native.py)
        rng = np.random.default_rng(seed)
        w = rng.random(n) + 1.0 / (1.0 + _host.mis2_degrees(pat))
        roots = _host.mis2(pat, w) & ~isolated
    elif distance >= 2:
        # vertices within two edges: the off-diagonal pattern of (pat + I)^2
        # (= pat^2 + pat off the diagonal); one product - the threaded native
        # SpGEMM from a few 10^5 entries - and the diagonal dropped on the
        # arrays (scipy's setdiag goes through COO: 2 s of 6 at cube N = 28)
        pI = (pat + sp.identity(n, format="csr")).tocsr()
        if pI.nnz > 400000 and not _host.use_numpy():
            pI.sort_indices()
            G2 = _host.spgemm(pI, pI)
        else:
            G2 = (pI @ pI).tocsr()
            G2.sort_indices()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(G2.indptr))
        keep = rows != G2.indices
        indptr = np.zeros(n + 1, dtype=G2.indptr.dtype)
        np.cumsum(np.bincount(rows[keep], minlength=n), out=indptr[1:])
        G2 = sp.csr_matrix((np.ones(int(keep.sum())), G2.indices[keep], indptr),
                           shape=(n, n))
        G2.has_sorted_indices = True
    else:
        G2 = pat          # MIS-1: roots two edges apart, aggregates = stars
    if G2 is not None:
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


def diagonal_block_mean(A, block):
    """Scalar operator on the nodes of a COUPLED block matrix: the mean of its
    ``block`` diagonal blocks ``A[i::block, i::block]``.  For the Newton
    velocity block ``F (x) I_d + N(w)`` that is ``F + (1/d) sum_i N_ii`` - the
    Picard stencil plus ``(div w / d)`` times a mass-like term."""
    A = sp.csr_matrix(A)
    F = None
    for i in range(block):
        B = sp.csr_matrix(A[i::block, i::block])
        F = B if F is None else F + B
    F = sp.csr_matrix(F * (1.0 / block))
    F.sort_indices()
    return F


def smoothed_aggregation_chain(A, block=1, coarse_rows=2000, max_levels=12,
                               theta=0.02, min_ratio=1.5, distance=2,
                               coupled="scalar"):
    """Prolongation chain ``[None, P_1, ..., P_L]`` for the finest operator
    ``A`` (``P_l`` maps level ``l-1`` to ``l``; the format
    ``PC.setMGInterpolations`` takes).  Coarsening stops at ``coarse_rows``
    rows (explicit inverse there) or when it stalls.

    ``coupled`` - what a COUPLED block operator (the Newton velocity block)
    is coarsened with: ``"scalar"`` (default) aggregates and smooths on the
    mean of its diagonal blocks - the scalar stencil, as for the Picard block -
    and prolongates every component alike, ``P = P_s (x) I_d``: the form the
    device producer refreshes (``P^T (F (x) I + N) P`` block by block with the
    one scalar ``P_s``); ``"block"`` aggregates on the block-norm graph and
    smooths with the coupled operator itself (a general ``P``; host refresh
    only)."""
    A = sp.csr_matrix(A)
    Ps = []
    cur = A
    curF = scalar_stencil(cur, block) if block > 1 else None
    if coupled not in ("scalar", "block"):
        raise ValueError("smoothed_aggregation_chain: coupled = %r" % (coupled,))
    while cur.shape[0] > coarse_rows and len(Ps) < max_levels - 1:
        if block > 1 and curF is None and coupled == "scalar":
            Fd = diagonal_block_mean(cur, block)
            Pf = sa_prolongator(Fd, theta, seed=len(Ps), distance=distance)
            if Pf.shape[1] * min_ratio > Pf.shape[0]:
                break                              # coarsening stalled
            P = _host.kron_expand(Pf, block)
            Ps.append(P)
            cur = _galerkin(cur, P)
            continue
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


# --------------------------------------------------------------- partitioned
class PartitionedSA(object):
    """Smoothed-aggregation hierarchy built RANK BY RANK from an operator of
    which every rank holds its rows only (``fem/partition.py``: global shape,
    other rows empty) - [ext PETSc] PCGAMG on a distributed matrix.

    * aggregation runs on the rank's diagonal block (couplings to other ranks'
      rows are ignored when choosing aggregates), so aggregates never cross a
      rank boundary and every rank OWNS the coarse dofs of its aggregates: the
      coarse levels are cut where the aggregates fall
      (``pcd_mg_set_level_cuts``), not evenly;
    * the tentative prolongator is smoothed with the TRUE rows of ``A``
      (``P = (I - omega/rho D^-1 A) T``; the tentative rows of the halo nodes
      are exchanged first), so ``P`` reaches into the neighbours' aggregates
      like the one-rank prolongator does; ``smooth="block"`` (or
      ``PCD_GAMG_SMOOTH=block``) smooths with the diagonal block only - ``P``
      block diagonal by rank, one exchange less, more iterations at the cuts;
    * the Galerkin operator ``P^T A P``: a rank needs the prolongator rows of
      its halo columns (exchanged once per coarsening) and forms its rows'
      TERMS of every coarse row; terms of other ranks' coarse rows travel to
      their owners (``HostComm.sum_rows``).  The prolongator handed over
      holds my rows AND the halo rows: the restriction rows of my coarse dofs
      (``P^T`` by rows) need the neighbours' entries in my columns, and with
      a structurally symmetric ``A`` those sit in my halo rows;
    * levels of at most ``replicate_rows`` dofs are gathered whole and the
      rest of the chain is the replicated ``smoothed_aggregation_chain``.

    The hierarchy therefore depends on the number of ranks (as PCGAMG's and
    BoomerAMG's do); iteration counts are reported per rank count.
    ``chain()`` / ``operators(F)`` give scalar prolongators / operators,
    coarsest first, global-shaped with this rank's rows populated on the
    partitioned levels; ``cuts[l]`` the row cuts of level ``l`` (scalar)."""

    def __init__(self, F, own, host, block=1, theta=0.02, coarse_rows=2000,
                 replicate_rows=60000, omega=4.0 / 3.0, distance=2,
                 min_ratio=1.5, max_levels=12, smooth=None):
        self.host, self.block = host, block
        # "global": the tentative prolongator is smoothed with the true rows
        # of A (halo rows of T exchanged); "block": with the rank's diagonal
        # block only (P block diagonal by rank - cheaper set-up, more
        # iterations at the cuts)
        self.smooth = smooth or os.environ.get("PCD_GAMG_SMOOTH", "global")
        if self.smooth not in ("global", "block"):
            raise ValueError("PartitionedSA: smooth = %r" % (self.smooth,))
        self.part = []                  # finest first: dicts of one coarsening
        self.tail = [None]              # replicated chain below (coarsest first)
        cur, cur_own = sp.csr_matrix(F), (int(own[0]), int(own[1]))
        n = cur.shape[0]
        self.n_fine, self.own_fine = n, cur_own
        whole = None
        # (the engine's rule is by size alone: a level of more than
        # PCD_REPLICATE_BELOW rows is partitioned - so coarsening goes on,
        # rank by rank, until a level is below that limit, whatever
        # `coarse_rows` says)
        # (the finest level is partitioned whatever its size: at least one
        # coarsening runs rank by rank)
        while (not self.part or n * block > replicate_rows) \
                and len(self.part) < max_levels - 1:
            lev = self._coarsen(cur, cur_own, theta, omega, distance,
                                seed=len(self.part))
            if lev["nc"] * min_ratio > n:
                break                                   # coarsening stalled
            self.part.append(lev)
            Ac = self._galerkin(cur, lev)
            n = lev["nc"]
            if n * block <= replicate_rows:
                whole = self._gather(Ac, lev)
                break
            cur, cur_own = Ac, lev["own_c"]
        if whole is None:
            if self.part and n * block > replicate_rows:
                raise ValueError(
                    "partitioned gamg: coarsening stalled at %d rows, above "
                    "the replication limit" % (n * block))
            if not self.part:
                raise ValueError("partitioned gamg: nothing to coarsen")
        self._tail_A = whole
        self.tail = smoothed_aggregation_chain(
            whole, block=1, coarse_rows=coarse_rows, theta=theta,
            distance=distance, max_levels=max_levels - len(self.part))

    # one coarsening of the rows [r0, r1) of `A` (global shape)
    def _coarsen(self, A, own, theta, omega, distance, seed):
        host = self.host
        r0, r1 = own
        n = A.shape[0]
        Arows = sp.csr_matrix(A[r0:r1])
        Add = sp.csr_matrix(Arows[:, r0:r1])
        agg, nagg = aggregate(_strength(Add, theta), seed, distance)
        counts = host.allgather(int(nagg))
        cuts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        off, nc = int(cuts[host.rank]), int(cuts[-1])
        T = _tentative(agg, nagg)
        Tg = sp.csr_matrix((T.data, T.indices.astype(np.int64) + off,
                            T.indptr), shape=(r1 - r0, nc))
        d = Add.diagonal().copy()
        d[d == 0.0] = 1.0
        rho = self._rho(A, own, d)
        # halo: the columns of my rows owned by other ranks
        cols = np.unique(Arows.indices)
        halo = cols[(cols < r0) | (cols >= r1)]
        bounds = host.allgather((r0, r1))
        want = host.allgather(halo)
        if self.smooth == "global":
            # the true rows of A: tentative rows of the halo nodes first
            AT = Arows @ self._with_halo_rows(Tg, own, n, want)
        else:
            AT = Add @ Tg
        Pg = sp.csr_matrix(Tg - (omega / rho) * (sp.diags(1.0 / d) @ AT))
        Pg.eliminate_zeros()
        Pg.sort_indices()                    # (r1 - r0) x nc, global columns
        # P_ext: n x nc, my rows and the halo rows populated
        Pext = self._with_halo_rows(Pg, own, n, want)
        if self.smooth == "global":
            # the restriction rows of my coarse dofs are taken from P_ext (my
            # rows + my halo rows): complete iff every row with an entry in my
            # columns is one of them - true when the pattern of A is
            # structurally symmetric; checked, because a silent miss would be
            # a restriction that is not P^T
            c0, c1 = off, off + nagg
            mine = Pext.tocsc()[:, c0:c1].nnz
            if host.sum(float(mine)) != host.sum(float(Pg.nnz)):
                raise ValueError(
                    "partitioned gamg: the operator's pattern is not "
                    "structurally symmetric - prolongator entries in a rank's "
                    "coarse columns lie outside its halo rows; use "
                    "PCD_GAMG_SMOOTH=block")
        # my rows only, global shape
        ip2 = np.zeros(n + 1, dtype=np.int64)
        ip2[r0 + 1:r1 + 1] = np.diff(Pg.indptr)
        np.cumsum(ip2, out=ip2)
        Pown = sp.csr_matrix((Pg.data, Pg.indices.astype(np.int64), ip2),
                             shape=(n, nc))
        return {"own": own, "own_c": (off, off + nagg), "nc": nc, "n": n,
                "cuts": cuts, "Pg": Pg, "Pext": Pext, "Pown": Pown,
                "bounds": bounds}

    def _with_halo_rows(self, Mloc, own, n, want):
        """``n x nc`` matrix holding my rows ``Mloc`` (global columns) and the
        rows every ``want[q]`` asks of their owners (``want``: the halo lists
        of all ranks, gathered once per coarsening)."""
        host = self.host
        r0, r1 = own
        reply = {}
        for q, need in enumerate(want):
            if q == host.rank or need.size == 0:
                continue
            mine = need[(need >= r0) & (need < r1)]
            if mine.size:
                sub = Mloc[mine - r0]
                reply[q] = (mine, sub.indptr, sub.indices, sub.data)
        got = host.allgather(reply)
        rows_i, rows_p, rows_c, rows_v = [], [], [], []
        for q, rep in enumerate(got):
            if q == host.rank or host.rank not in rep:
                continue
            idx, ip, ci, cv = rep[host.rank]
            rows_i.append(idx)
            rows_p.append(np.diff(ip))
            rows_c.append(ci)
            rows_v.append(cv)
        own_len = np.diff(Mloc.indptr)
        idx = np.concatenate([np.arange(r0, r1)] + rows_i)
        ln = np.concatenate([own_len] + rows_p)
        ci = np.concatenate([Mloc.indices] + rows_c)
        cv = np.concatenate([Mloc.data] + rows_v)
        # (rows arrive rank by rank in ascending order and my block sits
        # between them: a stable sort of the row ids orders the segments)
        order = np.argsort(idx, kind="stable")
        lens = ln[order]
        src0 = (np.cumsum(ln) - ln)[order]
        dst0 = np.cumsum(lens) - lens
        take = np.repeat(src0 - dst0, lens) + np.arange(int(lens.sum()))
        indptr = np.zeros(n + 1, dtype=np.int64)
        indptr[idx[order] + 1] = ln[order]
        np.cumsum(indptr, out=indptr)
        return sp.csr_matrix((cv[take], ci[take].astype(np.int64), indptr),
                             shape=(n, Mloc.shape[1]))

    def _rho(self, A, own, d_own, iters=15, seed=0):
        """Spectral radius estimate of D^-1 A for the whole operator: the
        power iteration of ``_rho`` with the product completed over the
        ranks."""
        n = A.shape[0]
        r0, r1 = own
        Arows = A[r0:r1]
        v = np.random.default_rng(seed).standard_normal(n)
        best = 0.0
        for k in range(iters):
            v /= np.linalg.norm(v)
            w = np.zeros(n)
            w[r0:r1] = (Arows @ v) / d_own
            v = self.host.sum(w)
            if k >= 2:
                best = max(best, float(np.linalg.norm(v)))
        return best

    def _galerkin(self, A, lev):
        """My rows of ``P^T A P`` (the coarse dofs of my aggregates), global
        shape.  ``P^T`` sums over fine rows: my rows' terms of the coarse rows
        of OTHER ranks (smoothing across the cut put entries there) travel to
        their owners (``HostComm.sum_rows``)."""
        r0, r1 = lev["own"]
        if A.nnz > 400000 and not _host.use_numpy():
            # the threaded native SpGEMM on the STRUCTURAL pattern: the
            # refresh of every nonlinear step is these two products, and a
            # pattern that does not depend on the values (scipy drops entries
            # that cancel to an exact zero) is a level the engine keeps
            # instead of taking it again
            AP = _host.spgemm(A, lev["Pext"], r0, r1)    # nloc x nc
            if "PgT" not in lev:
                lev["PgT"] = _host.transpose(lev["Pg"])
            C = _host.spgemm(lev["PgT"], AP)             # nc x nc, my terms
        else:
            AP = sp.csr_matrix(A[r0:r1] @ lev["Pext"])
            C = sp.csr_matrix(lev["Pg"].T @ AP)
        if self.smooth == "global":
            C = self.host.sum_rows(C, lev["cuts"])
        C.sort_indices()
        C.has_sorted_indices = True
        return C

    def _gather(self, Ac, lev):
        c0, c1 = lev["own_c"]
        rows = sp.csr_matrix(Ac[c0:c1])
        parts = self.host.allgather((rows.indptr, rows.indices, rows.data))
        nc = lev["nc"]
        mats = [sp.csr_matrix((dv, ix, ip), shape=(ip.size - 1, nc))
                for ip, ix, dv in parts]
        W = sp.vstack(mats, format="csr")
        W.sort_indices()
        return W

    # -- what the solver stack reads ------------------------------------------
    @property
    def nlevels(self):
        return len(self.part) + len(self.tail)

    def chain(self):
        """``[None, P_1, ..., P_L]`` (scalar), coarsest first."""
        key = "Pext" if self.smooth == "global" else "Pown"
        return list(self.tail) + [lev[key] for lev in self.part[::-1]]

    def level_cuts(self):
        """``cuts[l]`` (scalar dofs) of every level, ``None`` where the level
        is replicated or has the field's even cuts (the finest)."""
        L = self.nlevels
        cuts = [None] * L
        for k, lev in enumerate(self.part):
            l = L - 2 - k                       # level the coarsening produced
            if k < len(self.part) - 1:          # (the last one was gathered)
                cuts[l] = lev["cuts"]
        return cuts

    def partitioned_levels(self):
        L = self.nlevels
        flags = [False] * L
        flags[L - 1] = True
        for k in range(len(self.part) - 1):
            flags[L - 2 - k] = True
        return flags

    def operators(self, F):
        """Galerkin operators for the (re-assembled) finest operator ``F``
        (my rows, global shape), coarsest first; partitioned levels as my
        rows, replicated ones whole."""
        from .fem.multigrid import galerkin_chain
        ops_part = [sp.csr_matrix(F)]
        cur = ops_part[0]
        whole = None
        for k, lev in enumerate(self.part):
            Ac = self._galerkin(cur, lev)
            if k == len(self.part) - 1:
                whole = self._gather(Ac, lev)
            else:
                ops_part.append(Ac)
                cur = Ac
        tail_ops = galerkin_chain(whole, self.tail)     # coarsest first
        return tail_ops + ops_part[::-1]
