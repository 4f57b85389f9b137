"""Nested mesh hierarchies and P2 / P1 prolongation operators.

Input producer for the engine's geometric-multigrid inner solves (the
device-native stand-in for the hypre BoomerAMG cycles of the reference's
"iterative" configuration, ``demo/navier-stokes-pcd/demo_navier-stokes-pcd.py:
153-160``).  Meshes are nested by uniform refinement, so a fine node lies in
exactly one parent cell and the prolongation row is the parent's basis
evaluated at that node (<= 6 entries for P2, <= 3 for P1).  Coarse operators
are Galerkin products ``P^T A P`` (host side, scipy)."""

import numpy as np
import scipy.sparse as sp

from .. import _host
from .taylor_hood import TaylorHood, _p2_basis, small_det_inv


#: the numpy producer holds every element matrix of a level at once; beyond this
#: many cells on the finest level (largest sizes exercised: 819 200 triangles,
#: 196 608 tetrahedra) it may exhaust host memory before anything reaches the
#: GPU.  A mistyped level must fail here, not take the machine down; override
#: with FENAPACK_AMD_MAX_CELLS on a host with the memory for it.
MAX_FINEST_CELLS = {2: 1_000_000, 3: 800_000}


#: host memory the producer + solver set-up holds per cell of the finest level
#: (measured: problem and one linearisation 12-16 KB per tetrahedron, the whole
#: set-up with patterns, hierarchy and products about three times that)
HOST_BYTES_PER_CELL = {2: 12_000, 3: 48_000}


from .._guard import host_memory_available, concurrent_builds  # noqa: E402,F401


def _check_size(cells, dim, what):
    """A mistyped level, or R rank threads each building a 10 M-DOF problem,
    must fail HERE, not take the machine down (it did, once: eight thread
    ranks at cube N = 73 on a box whose container owned a fraction of the
    memory ``free`` reported).  ``FENAPACK_AMD_CONCURRENT_BUILDS`` = how many
    problems of this size the process builds at once (the thread-rank tools
    set it); ``FENAPACK_AMD_IGNORE_MEMORY=1`` skips the estimate."""
    import os
    limit = int(os.environ.get("FENAPACK_AMD_MAX_CELLS", MAX_FINEST_CELLS[dim]))
    if cells > limit:
        raise ValueError("%s: %.3g cells on the finest level exceed the host "
                         "assembler's limit of %d (FENAPACK_AMD_MAX_CELLS)"
                         % (what, cells, limit))
    if os.environ.get("FENAPACK_AMD_IGNORE_MEMORY") == "1":
        return
    builds = concurrent_builds()
    need = float(cells) * HOST_BYTES_PER_CELL[dim] * builds
    have = host_memory_available()
    if have is not None and need > 0.5 * have:
        raise MemoryError(
            "%s: about %.0f GB of host memory for %d concurrent build(s) of "
            "%.3g cells, %.0f GB available to this process (control group / "
            "MemAvailable); refusing above half of it "
            "(FENAPACK_AMD_IGNORE_MEMORY=1 overrides)"
            % (what, need / 1e9, builds, cells, have / 1e9))


class MeshHierarchy(object):
    """``meshes[0]`` (coarsest) ... ``meshes[-1]``; ``parents[l][f]`` = cell of
    level ``l-1`` containing fine cell ``f`` of level ``l``."""

    def __init__(self, base, levels):
        _check_size(float(base.num_cells) * float(2 ** base.dim) ** levels,
                    base.dim, "MeshHierarchy(levels=%d)" % levels)
        self.meshes, self.parents = [base], [None]
        for _ in range(levels):
            m = self.meshes[-1]
            fine = m.refine()          # children stacked in 4 blocks of nc
            self.meshes.append(fine)
            self.parents.append(np.arange(fine.num_cells) % m.num_cells)
        self._spaces = {}

    @property
    def finest(self):
        return self.meshes[-1]

    def space(self, level, finest_space=None):
        if level == len(self.meshes) - 1 and finest_space is not None:
            return finest_space
        if level not in self._spaces:
            self._spaces[level] = TaylorHood(self.meshes[level])
        return self._spaces[level]


class CubeHierarchy(MeshHierarchy):
    """Kuhn-triangulated unit cubes of ``n0 * 2**l`` cells per side; the mesh
    of level l is the uniform refinement of level l-1, parents are located
    geometrically (``kuhn_parents``)."""

    def __init__(self, n0, levels):
        from .mesh import unit_cube_mesh, kuhn_parents
        _check_size(6.0 * (float(n0) * 2.0 ** levels) ** 3, 3,
                    "CubeHierarchy(n0=%d, levels=%d)" % (n0, levels))
        self.meshes, self.parents = [unit_cube_mesh(n0)], [None]
        for l in range(1, levels + 1):
            fine = unit_cube_mesh(n0 * 2 ** l)
            self.parents.append(kuhn_parents(n0 * 2 ** (l - 1), fine))
            self.meshes.append(fine)
        self._spaces = {}


def _unique_entries(rows, cols, vals, shape):
    """CSR of the distinct (row, col) entries; the value of the FIRST
    occurrence is kept (every occurrence of a prolongation weight carries the
    same number), entries below 1e-14 are dropped."""
    from .. import _host
    if _host.use_numpy():
        key = rows.ravel().astype(np.int64) * shape[1] + cols.ravel()
        u, idx = np.unique(key, return_index=True)
        v = vals.ravel()[idx]
        keep = np.abs(v) > 1e-14
        P = sp.csr_matrix((v[keep], (u[keep] // shape[1], u[keep] % shape[1])),
                          shape=shape)
        P.sort_indices()
        return P
    g = _host.group_pairs(rows, cols, shape[0])
    ptr, order = g.members()
    v = vals.ravel()[order[ptr[:-1]]]
    keep = np.abs(v) > 1e-14
    urow = np.repeat(np.arange(shape[0], dtype=np.int64), np.diff(g.indptr))
    indptr = np.zeros(shape[0] + 1, dtype=np.int32)
    np.cumsum(np.bincount(urow[keep], minlength=shape[0]), out=indptr[1:])
    P = sp.csr_matrix((v[keep], g.ucols[keep].astype(np.int32), indptr),
                      shape=shape)
    P.has_sorted_indices = True
    g.release()
    return P


def prolongations(Vc, Vf, parent):
    """(scalar P2, P1) prolongation matrices coarse -> fine; any dimension."""
    mc, mf = Vc.mesh, Vf.mesh
    d = Vf.dim
    pc = mc.vertices[mc.cells[parent]]                 # parent simplices
    pf = mf.vertices[mf.cells]
    mids = np.stack([0.5 * (pf[:, i] + pf[:, j]) for i, j in Vf.local_edges],
                    axis=1)
    pts = np.concatenate([pf, mids], axis=1)           # P2 nodes of the cell
    T = np.stack([pc[:, k + 1] - pc[:, 0] for k in range(d)], axis=2)
    Tinv = small_det_inv(T)[1]                         # (nf, d, d)
    l1d = np.matmul(pts - pc[:, None, 0, :], Tinv.transpose(0, 2, 1))
    lam = np.concatenate([1.0 - l1d.sum(axis=2, keepdims=True), l1d], axis=2)
    na, nvl = Vf.na, Vf.nvl
    phi, _ = _p2_basis(lam.reshape(-1, nvl), Vc.local_edges, grad=False)
    phi = phi.reshape(lam.shape[0], na, na)
    rows = np.repeat(Vf.cell_dofs2[:, :, None], na, axis=2)
    cols = np.repeat(Vc.cell_dofs2[parent][:, None, :], na, axis=1)
    P2 = _unique_entries(rows, cols, phi, (Vf.nn, Vc.nn))
    rows1 = np.repeat(Vf.cell_dofs1[:, :, None], nvl, axis=2)
    cols1 = np.repeat(Vc.cell_dofs1[parent][:, None, :], nvl, axis=1)
    P1 = _unique_entries(rows1, cols1, lam[:, :nvl, :], (Vf.n_p, Vc.n_p))
    return P2, P1


def interleave(P, d=2):
    """Scalar P2 prolongation -> velocity prolongation (dofs d*node+comp);
    the result remembers its scalar factor (``.kron_scalar``)."""
    return _host.kron_expand(P, d)


interleave2 = interleave


class Interpolations(object):
    """Prolongation chains of a problem: ``velocity[l]`` / ``pressure[l]`` map
    level ``l-1`` to level ``l`` (index 0 is ``None``)."""

    def __init__(self, hierarchy, finest_space):
        self.velocity, self.pressure = [None], [None]
        L = len(hierarchy.meshes) - 1
        for l in range(1, L + 1):
            Vc = hierarchy.space(l - 1)
            Vf = hierarchy.space(l, finest_space)
            P2, P1 = prolongations(Vc, Vf, hierarchy.parents[l])
            self.velocity.append(interleave(P2, finest_space.dim))
            self.pressure.append(P1)

    def chain(self, field, nlevels=None):
        """Prolongations for ``nlevels`` levels ending at the finest one."""
        full = self.velocity if field == "u" else self.pressure
        if nlevels is None or nlevels >= len(full):
            return list(full)
        return [None] + full[len(full) - nlevels + 1:]


def galerkin_chain(A, chain, reduce=None, reduce_level=None):
    """Coarse operators ``A_{l-1} = P_l^T A_l P_l``; returns the list of
    operators, coarsest first (``ops[-1] is A``).  Large products run on the
    threaded native SpGEMM (same sums in the same order as scipy's).

    Partitioned producer: ``A`` holds one rank's rows only, so every product
    is that rank's TERMS of the coarse operator (``P^T`` sums over fine rows).
    ``reduce(C)``: the coarse levels are whole on every rank - summed once,
    below the finest level.  ``reduce_level(C, l) -> (C', rowsparse)``: level
    ``l`` may be partitioned too - the caller hands every coarse row's terms
    to its owner (``HostComm.sum_rows``) and says whether ``C'`` is again one
    rank's rows; products go on being reduced until a level comes back
    whole."""
    ops = [None] * len(chain)
    ops[-1] = sp.csr_matrix(A)
    rowsparse = reduce is not None or reduce_level is not None
    for l in range(len(chain) - 1, 0, -1):
        P = chain[l]
        if ops[l].nnz > 400000 and not _host.use_numpy():
            C = _host.spgemm(_host.spgemm(_host.transpose(P), ops[l]), P)
        else:
            C = (P.T @ ops[l] @ P).tocsr()
            C.sort_indices()
        if rowsparse:
            # (what comes out of a plain sum is whole, and so is everything
            # below it)
            if reduce_level is not None:
                C, rowsparse = reduce_level(C, l - 1)
            else:
                C, rowsparse = reduce(C), False
            C = sp.csr_matrix(C)
            C.sort_indices()
        ops[l - 1] = C
    return ops


def injection_map(P, block=1):
    """``inject[j]`` = fine node that coincides with coarse node ``j`` (P2
    spaces on nested meshes: every coarse node is a fine node), read off the
    prolongation ``P`` (fine x coarse, ``block`` interleaved components): the
    entry of column ``j`` that equals one."""
    Pc = sp.csc_matrix(P)
    cols = np.repeat(np.arange(Pc.shape[1]), np.diff(Pc.indptr))
    one = np.abs(Pc.data - 1.0) < 1e-9
    rows, cols = Pc.indices[one], cols[one]
    sel = (rows % block == 0) & (cols % block == 0)
    inject = np.full(Pc.shape[1] // block, -1, dtype=np.int64)
    inject[cols[sel] // block] = rows[sel] // block
    if np.any(inject < 0) or np.count_nonzero(sel) != inject.size:
        raise ValueError("prolongation does not contain an injection "
                         "(spaces not nested?)")
    return inject


def dense_csr(C):
    """Dense matrix as CSR with the FULL pattern (explicit zeros kept), so a
    refreshed coarse inverse always fits the pattern handed over first."""
    C = np.ascontiguousarray(C, dtype=np.float64)
    n, m = C.shape
    return sp.csr_matrix((C.ravel(), np.tile(np.arange(m, dtype=np.int32), n),
                          np.arange(n + 1, dtype=np.int32) * m), shape=(n, m))


def _kron_factor(A0, block):
    """``F`` if ``A0 == F (x) I_block`` on interleaved dofs, else ``None``."""
    if block < 2 or A0.shape[0] % block:
        return None
    if getattr(A0, "kron_block", 0) == block:
        return A0.kron_scalar
    return _host.kron_factor(A0, block)


def coarse_inverse(A0, block=1):
    """Explicit inverse of the coarsest operator (LU); a pseudo-inverse when
    the operator is singular (enclosed-flow ``R_p``: constants in the
    kernel).  ``block`` > 1: if the operator is ``F (x) I_block`` only the
    scalar factor is inverted (``inv(F (x) I) = inv(F) (x) I``)."""
    F = _kron_factor(A0, block)
    if F is not None:
        # inv(F) (x) I with the entries that couple no components NOT stored:
        # n entries per row instead of n * block - the engine recognises the
        # structure and reads inv(F) once for all components (k_dense_c<NC>;
        # cube N = 48, 6591 coarse dofs: 347 MB -> 39 MB per cycle)
        return _host.kron_expand(coarse_inverse(F, 1), block)
    D = A0.toarray()
    try:
        C = np.linalg.inv(D)
        probe = np.ones(D.shape[0]) / np.sqrt(D.shape[0])
        ok = np.all(np.isfinite(C)) and \
            np.linalg.norm(D @ (C @ probe) - probe) < 1e-8
    except np.linalg.LinAlgError:
        ok = False
    if not ok:
        C = np.linalg.pinv(D, rcond=1e-12)
    return dense_csr(C)
