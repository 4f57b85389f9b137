"""P2/P1 Taylor-Hood discretisation on simplices (triangles, tetrahedra),
vectorised with numpy.

This is the *input producer* of the PCD engine: it stands in for the
DOLFIN/FFC assembly the reference delegates to (``fenapack/assembling.py:
151-180``) and assembles exactly the forms of the reference demos
(``demo/navier-stokes-pcd/demo_navier-stokes-pcd.py:104-137``,
``demo/unsteady-navier-stokes-pcd/demo_unsteady-navier-stokes-pcd.py:
109-147``).  All matrices are fp64 CSR with int32 indices and a sparsity
pattern that stays fixed between nonlinear iterations (only values change),
which is what lets the engine keep its device-side layout and merely refresh
values (``fenapack/field_split_backend.py:79-83, 285-291``).

Numbering
---------
Nodes (vertices, then edge midpoints) are sorted lexicographically along the
longest axis of the domain, so contiguous row blocks are geometric slabs
(at most two neighbours per slab: the row partition of SURVEY 8e).  The mixed
space ``W = P2^d x P1`` is numbered node-major: ``[u_0 .. u_{d-1}, p]`` on
vertices, ``[u_0 .. u_{d-1}]`` on edge midpoints - interleaved like DOLFIN's
mixed dofmaps, so the fieldsplit index sets ``is_u`` / ``is_p`` are genuinely
non-contiguous (``fenapack/field_split.py:71-73``).
"""

import threading

import numpy as np
import scipy.sparse as sp
from scipy.special import roots_jacobi

_LAZY_LOCK = threading.RLock()

# Dunavant 7-point rule, exact to degree 5 (enough for w(P2).grad u(P1) v(P2))
_A1, _B1 = 0.059715871789770, 0.470142064105115
_A2, _B2 = 0.797426985353087, 0.101286507323456
_QP2 = np.array([[1 / 3., 1 / 3., 1 / 3.],
                 [_A1, _B1, _B1], [_B1, _A1, _B1], [_B1, _B1, _A1],
                 [_A2, _B2, _B2], [_B2, _A2, _B2], [_B2, _B2, _A2]])
_QW2 = np.array([0.225] + [0.132394152788506] * 3 + [0.125939180544827] * 3)

# 3-point Gauss rule on [0, 1] (degree 5) for boundary integrals
_G3X = 0.5 + 0.5 * np.array([-np.sqrt(0.6), 0.0, np.sqrt(0.6)])
_G3W = 0.5 * np.array([5 / 9., 8 / 9., 5 / 9.])
# triangle, degree 4 (6 points; weights sum to 1): the facet integrand
# (P2 wind) x (P1) x (P1) of the 3-D Robin term has degree 4
_T6A = (0.445948490915965, 0.091576213509771)
_T6W = (0.223381589678011, 0.109951743655322)
_T6L = np.array([[1 - 2 * a if i == k else a for i in range(3)]
                 for a in _T6A for k in range(3)])               # (6, 3) barycentric
_T6Wq = np.repeat(np.array(_T6W), 3)                            # (6,)


def _tet_rule(n=3):
    """Conical (Stroud) product of Gauss-Jacobi rules on the unit tetrahedron,
    exact to degree 2n - 1 (n = 3: degree 5, 27 points, positive weights).
    Returns barycentric points (nq, 4) and weights summing to 1."""
    xa, wa = roots_jacobi(n, 2, 0)
    xb, wb = roots_jacobi(n, 1, 0)
    xc, wc = roots_jacobi(n, 0, 0)
    xa, xb, xc = 0.5 * (xa + 1), 0.5 * (xb + 1), 0.5 * (xc + 1)
    wa, wb, wc = wa / 8, wb / 4, wc / 2
    pts, wts = [], []
    for i in range(n):
        for j in range(n):
            for k in range(n):
                x = xa[i]
                y = xb[j] * (1 - xa[i])
                z = xc[k] * (1 - xa[i]) * (1 - xb[j])
                pts.append([1 - x - y - z, x, y, z])
                wts.append(wa[i] * wb[j] * wc[k])
    wts = np.array(wts)
    return np.array(pts), wts / wts.sum()


def _conical_rule(d, n):
    """Conical (Stroud) product of Gauss-Jacobi rules on the unit d-simplex,
    exact to degree 2n - 1, positive weights; barycentric points (nq, d+1),
    weights summing to 1."""
    axes = []
    for k in range(d):                      # Jacobi weight (1 - x)^(d-1-k)
        x, w = roots_jacobi(n, d - 1 - k, 0)
        axes.append((0.5 * (x + 1), w))
    pts, wts = [], []
    for idx in np.ndindex(*([n] * d)):
        rest, coords, wt = 1.0, [], 1.0
        for k in range(d):
            x, w = axes[k]
            coords.append(x[idx[k]] * rest)
            rest *= 1.0 - x[idx[k]]
            wt *= w[idx[k]]
        pts.append([1.0 - sum(coords)] + coords)
        wts.append(wt)
    wts = np.array(wts)
    return np.array(pts), wts / wts.sum()


def small_det_inv(T):
    """Determinants and inverses of a batch of 2x2 / 3x3 matrices by cofactors
    (``numpy.linalg.inv`` runs one LAPACK call per matrix: 25 ms for the 2 10^5
    cells of cavity level 6 against 2 ms here)."""
    d = T.shape[-1]
    if d == 2:
        a, b, c, e = T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1]
        det = a * e - b * c
        inv = np.empty_like(T)
        inv[:, 0, 0], inv[:, 0, 1] = e / det, -b / det
        inv[:, 1, 0], inv[:, 1, 1] = -c / det, a / det
        return det, inv
    if d == 3:
        m = [[T[:, i, j] for j in range(3)] for i in range(3)]
        cof = np.empty_like(T)                      # cof[:, j, i]: adjugate
        for i in range(3):
            i1, i2 = (i + 1) % 3, (i + 2) % 3
            for j in range(3):
                j1, j2 = (j + 1) % 3, (j + 2) % 3
                cof[:, j, i] = m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1]
        det = m[0][0] * cof[:, 0, 0] + m[0][1] * cof[:, 1, 0] \
            + m[0][2] * cof[:, 2, 0]
        cof /= det[:, None, None]
        return det, cof
    return np.linalg.det(T), np.linalg.inv(T)


def _p2_basis(lam, edges=((1, 2), (2, 0), (0, 1)), grad=True):
    """P2 basis at barycentric points ``lam`` (nq, d+1).

    Local dofs: the d+1 vertices, then one per local edge (i, j) in the order
    of ``edges``.  Returns (phi (nq, na), dphi/dlam (nq, na, d+1)); with
    ``grad=False`` the second item is ``None`` (millions of points: the
    prolongation weights need the values only).
    """
    nq, nvl = lam.shape
    na = nvl + len(edges)
    phi = np.empty((nq, na))
    dphi = np.zeros((nq, na, nvl)) if grad else None
    for i in range(nvl):
        phi[:, i] = lam[:, i] * (2 * lam[:, i] - 1)
        if grad:
            dphi[:, i, i] = 4 * lam[:, i] - 1
    for k, (i, j) in enumerate(edges):
        phi[:, nvl + k] = 4 * lam[:, i] * lam[:, j]
        if grad:
            dphi[:, nvl + k, i] = 4 * lam[:, j]
            dphi[:, nvl + k, j] = 4 * lam[:, i]
    return phi, dphi


class FixedPattern(object):
    """CSR sparsity pattern assembled once; values refreshed by ``bincount``.

    The grouping of the element entries is native (``libpcd_host.so``,
    ``pcdh_group_pairs`` / ``pcdh_pattern_cells``); the numpy route
    (``numpy.unique``) is its checker (``FENAPACK_AMD_NUMPY_PRODUCER=1``)."""

    def __init__(self, rows, cols, shape):
        from .. import _host
        rows = np.asarray(rows, dtype=np.int64).ravel()
        cols = np.asarray(cols, dtype=np.int64).ravel()
        self.shape = shape
        if _host.use_numpy():
            key = rows * shape[1] + cols
            ukey, inv = np.unique(key, return_inverse=True)
            self.inv = inv.ravel()
            self.nnz = ukey.size
            urow = ukey // shape[1]
            self.indices = (ukey % shape[1]).astype(np.int32)
            self.indptr = np.zeros(shape[0] + 1, dtype=np.int32)
            np.cumsum(np.bincount(urow, minlength=shape[0]),
                      out=self.indptr[1:])
            return
        g = _host.group_pairs(rows, cols, shape[0])
        self._from_group(g)

    def _from_group(self, g):
        self.nnz = g.nnz
        self.indices = g.ucols.astype(np.int32)
        self.indptr = g.indptr.astype(np.int32)
        self.inv = g.inv
        self._group = g            # members() are fetched on demand

    @classmethod
    def from_cells(cls, rdofs, cdofs, shape):
        """Pattern of a form from cell dof tables (ncell, nr) x (ncell, nc);
        ``inv`` is laid out (cell, a, b)."""
        from .. import _host
        if _host.use_numpy():
            nr, nc = rdofs.shape[1], cdofs.shape[1]
            return cls(np.repeat(rdofs[:, :, None], nc, axis=2),
                       np.repeat(cdofs[:, None, :], nr, axis=1), shape)
        self = cls.__new__(cls)
        self.shape = shape
        self._from_group(_host.pattern_cells(rdofs, cdofs, shape[0]))
        return self

    @classmethod
    def from_csr(cls, indptr, indices, inv, shape):
        self = cls.__new__(cls)
        self.shape = shape
        self.indptr = np.asarray(indptr, dtype=np.int32)
        self.indices = np.asarray(indices, dtype=np.int32)
        self.nnz = self.indices.size
        self.inv = inv
        return self

    @property
    def inv(self):
        """Element entry -> CSR slot.  Patterns built as a permutation
        (``order``: slot -> input position) derive it on first use."""
        if getattr(self, "_inv", None) is None and hasattr(self, "order"):
            self._inv = np.empty(self.order.size, dtype=np.int64)
            self._inv[self.order] = np.arange(self.order.size)
        return self._inv

    @inv.setter
    def inv(self, value):
        self._inv = value

    @property
    def rows(self):
        if not hasattr(self, "_rows"):
            self._rows = np.repeat(np.arange(self.shape[0], dtype=np.int32),
                                   np.diff(self.indptr))
        return self._rows

    @property
    def keys(self):
        if not hasattr(self, "_keys"):
            self._keys = self.rows.astype(np.int64) * self.shape[1] \
                + self.indices
        return self._keys

    def members(self):
        """(ptr, order): element entries of CSR slot k are
        ``order[ptr[k]:ptr[k+1]]`` in ascending position (= the stable argsort
        of ``inv``)."""
        with _LAZY_LOCK:        # (spaces may be shared by rank threads)
            if getattr(self, "_members", None) is None and \
                    getattr(self, "_group", None) is not None:
                self._members = self._group.members()
                self._group.release()
                self._group = None
        if getattr(self, "_members", None) is None:
            inv = np.asarray(self.inv).ravel()
            order = np.argsort(inv, kind="stable")
            ptr = np.zeros(self.nnz + 1, dtype=np.int64)
            np.cumsum(np.bincount(inv, minlength=self.nnz), out=ptr[1:])
            self._members = (ptr, order)
        return self._members

    @classmethod
    def from_unique(cls, rows, cols, shape):
        """Pattern of entries known to be pairwise distinct; ``order`` maps
        CSR slot -> input position (assembly is a pure permutation)."""
        from .. import _host
        rows = np.asarray(rows).ravel()
        cols = np.asarray(cols).ravel()
        n = rows.size
        self = cls.__new__(cls)
        self.shape = shape
        self.nnz = n
        if _host.use_numpy():
            tag = sp.coo_matrix((np.arange(1, n + 1, dtype=np.float64),
                                 (rows, cols)), shape=shape).tocsr()
            tag.sort_indices()
            if tag.nnz != n:
                raise ValueError("from_unique: duplicate entries")
            order = tag.data.astype(np.int64) - 1      # CSR slot -> input
            self.indices = tag.indices.astype(np.int32)
            self.indptr = tag.indptr.astype(np.int32)
        else:
            g = _host.group_pairs(rows, cols, shape[0])
            if g.nnz != n:
                raise ValueError("from_unique: duplicate entries")
            order = g.members()[1]
            self.indices = g.ucols.astype(np.int32)
            self.indptr = g.indptr.astype(np.int32)
            g.release()
        self.order = order
        self.inv = None                      # derived from `order` on demand
        return self

    def sum_entries(self, vals):
        """Element values (laid out like ``inv``) summed per CSR slot; large
        patterns: the threaded native gather over the member lists - the same
        additions in the same order as ``numpy.bincount``."""
        from .. import _host
        vals = np.asarray(vals).ravel()
        if vals.size >= 200000 and not _host.use_numpy():
            ptr, order = self.members()
            return _host.gather_sum(ptr, order, vals)
        return np.bincount(self.inv, weights=vals, minlength=self.nnz)

    def assemble(self, vals):
        return self.matrix(self.sum_entries(vals))

    def matrix(self, data):
        """CSR on this pattern; explicit zeros are kept on purpose."""
        return sp.csr_matrix((data, self.indices, self.indptr),
                             shape=self.shape)

    def locate(self, rows, cols):
        """Positions (in CSR data order) of existing entries (rows, cols)."""
        key = (np.asarray(rows, dtype=np.int64) * self.shape[1]
               + np.asarray(cols, dtype=np.int64))
        pos = np.searchsorted(self.keys, key)
        assert np.all(self.keys[pos] == key), "entry outside the pattern"
        return pos


def _kron_rows(S, d, per_row):
    """Row pointer of the pattern whose row ``d*i + k`` holds ``per_row``
    entries for every entry of row ``i`` of the scalar pattern ``S``; also
    returns (row length, row start) of ``S`` per scalar entry's row."""
    ip = S.indptr.astype(np.int64)
    ln = np.diff(ip)
    indptr = np.zeros(d * ln.size + 1, dtype=np.int64)
    np.cumsum(np.repeat(ln * per_row, d), out=indptr[1:])
    return indptr, ln, ip


class BlockPattern(FixedPattern):
    """Pattern of a block operator whose components repeat a SCALAR pattern
    (velocity blocks: ``F (x) I_d``, ``F (x) ones(d, d)``; gradient /
    divergence blocks).  ``pos[c]`` gives, per scalar entry, its slot in the
    block CSR for component ``c``; assembly is one scalar ``bincount`` per
    component plus a scatter - no element-level map of the block (that one,
    ``inv``, is derived on demand for the few callers that ask)."""

    def __init__(self, scalar, pos, indptr, indices, shape, cell_axes):
        self.shape = shape
        self.scalar = scalar
        self.pos = pos                          # list of int64 arrays (nnz_s,)
        self.indptr = np.asarray(indptr, dtype=np.int32)
        self.indices = np.asarray(indices, dtype=np.int32)
        self.nnz = self.indices.size
        # how the scalar element map is laid out inside a cell of `vals`:
        # axes of the scalar (cell, row, col) entries in the vals array
        self._cell_axes = cell_axes

    @property
    def inv(self):
        if getattr(self, "_inv", None) is None:
            si = self.scalar.inv
            if self._cell_axes == "T":           # scalar map is (c, j, a)
                nc = self.scalar_cells
                si = si.reshape(nc[0], nc[1], nc[2]).transpose(0, 2, 1)
            self._inv = np.stack([p[si] for p in self.pos], axis=-1).ravel()
        return self._inv

    def assemble_components(self, comps):
        """``comps[c]``: element values of component ``c`` laid out like the
        scalar pattern's element entries."""
        data = np.zeros(self.nnz)
        done = {}                       # components that ARE the same array
        for p, v in zip(self.pos, comps):
            if v is None:
                continue
            if id(v) not in done:
                done[id(v)] = self.scalar.sum_entries(v)
            data[p] = done[id(v)]
        return self.matrix(data)

    def assemble(self, vals):
        """``vals`` (cells, ..., ncomp): the last axis runs over ``pos``."""
        vals = np.asarray(vals)
        ncomp = len(self.pos)
        v = vals.reshape(-1, ncomp)
        if self._cell_axes == "T":
            nc = self.scalar_cells
            v = vals.reshape(nc[0], nc[2], nc[1], ncomp).transpose(0, 2, 1, 3)
            v = v.reshape(-1, ncomp)
        return self.assemble_components([v[:, c] for c in range(ncomp)])


class TaylorHood(object):
    """P2 velocity / P1 pressure spaces on a :class:`Mesh` (d = 2) or a
    :class:`TetMesh` (d = 3)."""

    def __init__(self, mesh, axes=None):
        """``axes``: sort order of the coordinate axes for the numbering
        (default: longest extent first).  A sub-space of a partitioned build
        passes the WHOLE domain's order, so that its numbering is the global
        one restricted (``fem/partition.py``)."""
        self.mesh = mesh
        self.dim = d = mesh.dim
        self.local_edges = tuple(mesh.local_edges)
        self.nvl = d + 1                              # vertices per cell
        self.na = self.nvl + len(self.local_edges)    # P2 dofs per cell
        nv, ne = mesh.num_vertices, mesh.num_edges
        self.nv, self.ne = nv, ne
        self.nn = nn = nv + ne                       # scalar P2 dofs
        coords = np.concatenate([mesh.vertices, mesh.edge_midpoints()])
        ext = coords.max(axis=0) - coords.min(axis=0)
        if axes is None:
            axes = list(np.argsort(-ext, kind="stable"))  # longest axis first
        self.axes = axes = [int(a) for a in axes]
        # round so that nodes meant to share a coordinate compare equal
        # whatever sequence of midpoint averages produced them
        key = np.round(coords * 2.0 ** 30)
        order = np.lexsort(tuple(key[:, a] for a in reversed(axes)))
        rank = np.empty(nn, dtype=np.int64)
        rank[order] = np.arange(nn)
        self.node_coords = coords[order]             # by scalar P2 dof
        is_vertex_sorted = order < nv
        # P1 numbering: vertices in the same geometric order
        pnum = np.empty(nv, dtype=np.int64)
        pnum[order[is_vertex_sorted]] = np.arange(nv)
        self.p_coords = mesh.vertices[order[is_vertex_sorted]]
        self.p2_of_p1 = np.nonzero(is_vertex_sorted)[0]   # P2 dof of P1 dof
        # cell dof tables
        self.cell_dofs2 = np.concatenate(
            [rank[mesh.cells], rank[nv + mesh.cell_edges]], axis=1)
        self.cell_dofs1 = pnum[mesh.cells]
        self._rank, self._pnum = rank, pnum
        # sizes and fieldsplit index sets (mixed, node-major numbering)
        self.n_u, self.n_p = d * nn, nv
        self.ndof = self.n_u + self.n_p
        width = np.where(is_vertex_sorted, d + 1, d)
        start = np.concatenate([[0], np.cumsum(width)[:-1]])
        self.is_u = (start[:, None] + np.arange(d)).ravel()
        self.is_p = start[is_vertex_sorted] + d
        self._geometry()

    # ------------------------------------------------------------------ geo
    def _geometry(self):
        m, d = self.mesh, self.dim
        p = m.vertices[m.cells]                       # (nc, d+1, d)
        # affine map x = p0 + T lam_{1..d}; rows of T^-1 are grad lam_{1..d}
        T = np.stack([p[:, k + 1] - p[:, 0] for k in range(d)], axis=2)
        det, Tinv = small_det_inv(T)                  # (nc,), (nc, d, d)
        g = np.empty((m.num_cells, d + 1, d))
        g[:, 1:, :] = Tinv
        g[:, 0, :] = -Tinv.sum(axis=1)
        self.gradlam = g
        self.area = det / (2.0 if d == 2 else 6.0)    # cell measure
        if d == 2:
            qp, qw = _QP2, _QW2
        else:
            qp, qw = _tet_rule(3)
        self.phi, self._dphi = _p2_basis(qp, self.local_edges)  # (nq, na)
        self._qw = qw
        # the streamline-diffusion term with a P2 wind has degree 6 (FFC
        # would pick a degree-6 scheme for it): its own rule, exact to 7
        self.qp_s, self.qw_s = _conical_rule(d, 4)
        self.phi_s, self.dphi_s = _p2_basis(self.qp_s, self.local_edges)
        self.psi = qp                                           # (nq, d+1)
        edges = [np.linalg.norm(p[:, i] - p[:, j], axis=1)
                 for i in range(d + 1) for j in range(i)]
        self.cell_hmax = np.maximum.reduce(edges)
        if d == 2:
            # DOLFIN Cell::h() = 2 * circumradius for simplices
            self.cell_h = edges[0] * edges[1] * edges[2] / (2.0 * self.area)
        else:
            self.cell_h = self.cell_hmax

    @property
    def gphi(self):
        """Physical basis gradients at the quadrature points, (nc, nq, na, d).
        Large (20 GB at 3 M tetrahedra): the operators below are assembled
        from reference tensors instead; this array is built on demand for the
        terms that still integrate point by point (SUPG, Newton, tests)."""
        if not hasattr(self, "_gphi"):
            self._gphi = np.einsum('qak,ckd->cqad', self._dphi, self.gradlam)
        return self._gphi

    @property
    def wq(self):
        """Quadrature weights times cell measure, (nc, nq)."""
        return self._qw[None, :] * self.area[:, None]

    def _ref(self):
        """Reference-cell tensors of the fixed forms.  On an affine simplex
        every element matrix of this module is a contraction of a few
        geometric numbers per cell (``grad lam_k . grad lam_l``,
        ``U_m . grad lam_k``) with one of these constant tensors - one GEMM
        per operator instead of a loop over quadrature points.  They are
        integrated with the same rules as the point-wise forms (exact for
        these polynomial degrees), so both routes agree to round-off."""
        if hasattr(self, "_ref_cache"):
            return self._ref_cache
        qw, phi, dphi, psi = self._qw, self.phi, self._dphi, self.psi
        R = {
            # M[a,b] = int phi_a phi_b
            "M": np.einsum('q,qa,qb->ab', qw, phi, phi),
            # K[k,l,a,b] = int d_k phi_a d_l phi_b
            "K": np.einsum('q,qak,qbl->klab', qw, dphi, dphi),
            # C[m,k,a,b] = int phi_m phi_a d_k phi_b
            "C": np.einsum('q,qm,qa,qbk->mkab', qw, phi, phi, dphi),
            # B[l,a,j] = int psi_j d_l phi_a
            "B": np.einsum('q,qj,qal->laj', qw, psi, dphi),
            # P[i,j] = int psi_i psi_j ; S[m,i] = int phi_m psi_i
            "P": np.einsum('q,qi,qj->ij', qw, psi, psi),
            "S": np.einsum('q,qm,qi->mi', qw, phi, psi),
        }
        self._ref_cache = R
        return R

    # ------------------------------------------------------------- patterns
    def _patterns(self, coupled):
        """Fixed sparsity patterns in fieldsplit-local numbering.

        Only the SCALAR patterns are grouped (P2 x P2, P2 x P1, P1 x P2,
        P1 x P1: ``pcdh_pattern_cells`` on the cell dof tables); the velocity
        patterns are their Kronecker expansions - row ``d*i + k`` of the
        block copies row ``i`` of the scalar pattern - so the element -> entry
        maps follow by arithmetic instead of a sort of ``d`` (or ``d^2``)
        times as many keys."""
        key = "_pat_%d" % int(coupled)
        if hasattr(self, key):
            return getattr(self, key)
        from .. import _host
        if _host.use_numpy():
            pat = self._patterns_numpy(coupled)
            setattr(self, key, pat)
            return pat
        d, na, nvl = self.dim, self.na, self.nvl
        d2, d1 = self.cell_dofs2, self.cell_dofs1
        other = getattr(self, "_pat_%d" % int(not coupled), None)
        pat = {}
        if other is not None:
            for k in ("A01", "A10", "PP", "SS"):
                pat[k] = other[k]
        else:
            pat["SS"] = FixedPattern.from_cells(d2, d2, (self.nn, self.nn))
            pat["PP"] = FixedPattern.from_cells(d1, d1, (self.n_p, self.n_p))
        SS = pat["SS"]
        comp = np.arange(d, dtype=np.int64)

        def rows_of(S):
            return S.rows.astype(np.int64), S.indptr.astype(np.int64)

        rS, ipS = rows_of(SS)
        lnS = np.diff(ipS)
        offS = np.arange(SS.nnz, dtype=np.int64) - ipS[rS]   # position in its row
        colS = SS.indices.astype(np.int64)
        if coupled:
            # row d*i+k holds (j, e) sorted by d*j + e: d*len_i entries
            indptr = np.zeros(d * lnS.size + 1, dtype=np.int64)
            np.cumsum(np.repeat(d * lnS, d), out=indptr[1:])
            pos = [indptr[d * rS + k] + d * offS + e
                   for k in range(d) for e in range(d)]
            indices = np.empty(indptr[-1], dtype=np.int32)
            for k in range(d):
                for e in range(d):
                    indices[pos[k * d + e]] = d * colS + e
        else:
            indptr = np.zeros(d * lnS.size + 1, dtype=np.int64)
            np.cumsum(np.repeat(lnS, d), out=indptr[1:])
            pos = [indptr[d * rS + k] + offS for k in range(d)]
            indices = np.empty(indptr[-1], dtype=np.int32)
            for k in range(d):
                indices[pos[k]] = d * colS + k
        pat["A00"] = BlockPattern(SS, pos, indptr, indices,
                                  (self.n_u, self.n_u), "N")
        if other is None:
            # A01: rows (a, k) -> d*node + k, cols j; vals laid out (c, a, j, k)
            G = FixedPattern.from_cells(d2, d1, (self.nn, self.n_p))
            rG, ipG = rows_of(G)
            lnG = np.diff(ipG)
            offG = np.arange(G.nnz, dtype=np.int64) - ipG[rG]
            indptr = np.zeros(d * lnG.size + 1, dtype=np.int64)
            np.cumsum(np.repeat(lnG, d), out=indptr[1:])
            pos = [indptr[d * rG + k] + offG for k in range(d)]
            indices = np.empty(indptr[-1], dtype=np.int32)
            for k in range(d):
                indices[pos[k]] = G.indices
            pat["A01"] = BlockPattern(G, pos, indptr, indices,
                                      (self.n_u, self.n_p), "N")
            # A10: rows j, cols d*node + k; vals laid out (c, a, j, k) as well,
            # the scalar element map (c, j, a)
            Gt = FixedPattern.from_cells(d1, d2, (self.n_p, self.nn))
            slot = np.arange(Gt.nnz, dtype=np.int64)
            pos = [d * slot + k for k in range(d)]
            indices = (d * Gt.indices.astype(np.int64)[:, None] + comp).ravel()
            bp = BlockPattern(Gt, pos, d * Gt.indptr.astype(np.int64), indices,
                              (self.n_p, self.n_u), "T")
            bp.scalar_cells = (d1.shape[0], nvl, na)
            pat["A10"] = bp
        setattr(self, key, pat)
        return pat

    def _patterns_numpy(self, coupled):
        """The numpy route (checker of the native one): every pattern from
        the explicit (row, col) pairs of its element entries."""
        d, na, nvl = self.dim, self.na, self.nvl
        d2, d1 = self.cell_dofs2, self.cell_dofs1
        r2 = np.repeat(d2[:, :, None], na, axis=2)    # (nc,na,na) row = a
        c2 = np.repeat(d2[:, None, :], na, axis=1)    # col = b
        comp = np.arange(d)
        pat = {}
        if coupled:
            # rows (a,c) x cols (b,e), all component pairs: (nc,na,na,d,d)
            rows = np.broadcast_to(d * r2[..., None, None]
                                   + comp[:, None], r2.shape + (d, d))
            cols = np.broadcast_to(d * c2[..., None, None]
                                   + comp[None, :], c2.shape + (d, d))
        else:
            rows = d * r2[..., None] + comp
            cols = d * c2[..., None] + comp
        pat["A00"] = FixedPattern(rows, cols, (self.n_u, self.n_u))
        # A01 rows (a,c) x cols j
        r = d * np.repeat(d2[:, :, None], nvl, axis=2)[..., None] + comp
        c = np.repeat(np.repeat(d1[:, None, :], na, axis=1)[..., None], d,
                      axis=3)
        pat["A01"] = FixedPattern(r, c, (self.n_u, self.n_p))
        pat["A10"] = FixedPattern(c, r, (self.n_p, self.n_u))
        r1 = np.repeat(d1[:, :, None], nvl, axis=2)
        c1 = np.repeat(d1[:, None, :], nvl, axis=1)
        pat["PP"] = FixedPattern(r1, c1, (self.n_p, self.n_p))
        pat["SS"] = FixedPattern(r2, c2, (self.nn, self.nn))
        return pat

    # ----------------------------------------------------- scalar P2 pieces
    def wind_at_qp(self, U):
        """``U`` (nn, d) nodal P2 velocity -> (nc, nq, d) and its gradient
        (nc, nq, d[comp], d[deriv])."""
        Uc = U[self.cell_dofs2]                                 # (nc,na,d)
        w = np.einsum('qa,cak->cqk', self.phi, Uc)
        gw = np.einsum('cqad,cak->cqkd', self.gphi, Uc)
        return w, gw

    def p2_stiffness_cells(self):
        """(grad phi_b, grad phi_a) per cell; geometry only, so it is kept
        (read-only) between nonlinear steps while it is below 512 MB."""
        K = getattr(self, "_stiff_cells", None)
        if K is not None:
            return K
        g, na, nvl = self.gradlam, self.na, self.nvl
        gg = np.matmul(g, g.transpose(0, 2, 1)) * self.area[:, None, None]
        K = gg.reshape(-1, nvl * nvl) @ self._ref()["K"].reshape(nvl * nvl, -1)
        K = K.reshape(-1, na, na)
        if K.nbytes <= 512 << 20:
            K.setflags(write=False)
            self._stiff_cells = K
        return K

    def p2_mass_cells(self):
        return self.area[:, None, None] * self._ref()["M"][None]

    def p2_convection_cells(self, w):
        """((w.grad) phi_b, phi_a) from the wind at the quadrature points
        (point-wise route; the assembler uses ``p2_convection_nodal``)."""
        wg = np.einsum('cqd,cqbd->cqb', w, self.gphi)           # w.grad phi_b
        return np.einsum('cq,qa,cqb->cab', self.wq, self.phi, wg)

    def p2_convection_nodal(self, U):
        """The same element matrices from the nodal P2 wind ``U`` (nn, d):
        ``C_c = sum_{m,k} |T| (U_m . grad lam_k) Chat[m,k]``."""
        from .. import _host
        na, nvl = self.na, self.nvl
        if self.cell_dofs2.shape[0] > 20000 and not _host.use_numpy():
            # one threaded pass (bitwise the chain below)
            ug = _host.wind_gradlam(self.cell_dofs2, U, self.gradlam, self.area)
            C = ug.reshape(-1, na * nvl) @ self._ref()["C"].reshape(na * nvl, -1)
            return C.reshape(-1, na, na)
        Uc, g = U[self.cell_dofs2], self.gradlam      # (nc,na,d), (nc,nvl,d)
        # (elementwise: a batched matmul of (na, d) x (d, nvl) blocks runs one
        # tiny GEMM per cell)
        ug = Uc[:, :, None, 0] * g[:, None, :, 0]
        for k in range(1, self.dim):
            ug += Uc[:, :, None, k] * g[:, None, :, k]
        ug *= self.area[:, None, None]
        C = ug.reshape(-1, na * nvl) @ self._ref()["C"].reshape(na * nvl, -1)
        return C.reshape(-1, na, na)

    def _a01_cells(self):
        """vals[c, a, j, comp] = -int psi_j d_comp phi_a."""
        B = self._ref()["B"]                                    # (l, a, j)
        na, nvl = self.na, self.nvl
        # v[c, d, (a, j)] = sum_l grad lam_l[c, d] B[l, (a, j)]
        v = np.matmul(self.gradlam.transpose(0, 2, 1),
                      B.reshape(nvl, na * nvl)[None])
        v *= -self.area[:, None, None]
        return np.ascontiguousarray(
            v.reshape(-1, self.dim, na, nvl).transpose(0, 2, 3, 1))

    def p2_supg_cells(self, U, delta):
        """delta * (w.grad u, w.grad v): streamline diffusion added to the
        preconditioner's 00-block (demo_navier-stokes-pcd.py:122-125);
        ``U`` (nn, d) nodal wind, integrated with the degree-7 rule."""
        Uc = U[self.cell_dofs2]                                 # (nc,na,d)
        nc = Uc.shape[0]
        out = np.empty((nc, self.na, self.na))
        sq = np.sqrt(self.qw_s)
        # chunks bound the (cells, points, basis) temporaries
        step = max(1, 4_000_000 // (self.qw_s.size * self.na))
        for c0 in range(0, nc, step):
            c1 = min(nc, c0 + step)
            # U_m . grad lam_k per cell, then w . grad lam_k at the points
            ug = np.matmul(Uc[c0:c1], self.gradlam[c0:c1].transpose(0, 2, 1))
            wl = np.matmul(self.phi_s[None], ug)                # (c, q, k)
            # w . grad phi_b = sum_k dphi_s[q, b, k] wl[c, q, k]
            # (batched over the points: (q, c, k) @ (q, k, b) -> (q, c, b))
            wg = np.matmul(wl.transpose(1, 0, 2),
                           self.dphi_s.transpose(0, 2, 1))
            wg *= sq[:, None, None]
            wg = np.ascontiguousarray(wg.transpose(1, 0, 2))    # (c, q, b)
            out[c0:c1] = np.matmul(wg.transpose(0, 2, 1), wg)
        out *= (delta * self.area)[:, None, None]
        return out

    # ------------------------------------------------------------- velocity
    def assemble_A00(self, nu, U=None, idt=0.0, newton=False, delta=None):
        """Velocity block: nu*(grad u, grad v) + ((w.grad)u, v) [+ idt*(u,v)]
        [+ Newton term ((u.grad)w, v)] [+ SUPG]."""
        d = self.dim
        pat = self._patterns(newton)["A00"]
        # the iterate-independent part once per (nu, idt) - every pass over the
        # (cells, na, na) element matrices is a pass over 0.5-2 GB at the 3-D
        # sizes, on one core - and the convection term added to it in place
        # (same sums in the same order: bitwise the three-temporary form)
        key = (float(nu), float(idt))
        S0 = getattr(self, "_s0_cells", (None, None))
        if S0[0] != key:
            S = nu * self.p2_stiffness_cells()
            if idt:
                S = S + idt * self.p2_mass_cells()
            S0 = (key, S)
            if S.nbytes <= 4 << 30:           # (cube N = 73: 1.9 GB beside a 50 GB build)
                S.setflags(write=False)
                self._s0_cells = S0
        S = S0[1]
        if U is not None:
            C = self.p2_convection_nodal(U)
            np.add(S, C, out=C)
            S = C
            if delta is not None:
                S += self.p2_supg_cells(U, delta)
        if not newton:
            if isinstance(pat, BlockPattern):
                return pat.assemble_components([S] * d)
            vals = np.repeat(S[..., None], d, axis=3)           # (nc,na,na,d)
            return pat.assemble(vals)
        vals = np.zeros(S.shape + (d, d))
        for k in range(d):
            vals[..., k, k] = S
        if U is not None:
            # N[(a,c),(b,e)] = int phi_a phi_b d_e w_c
            # = |T| sum_{m,l} U[m,c] (grad lam_l)_e  int d_l phi_m phi_a phi_b
            R = self._ref()
            if "N" not in R:
                R["N"] = np.einsum('q,qml,qa,qb->mlab', self._qw, self._dphi,
                                   self.phi, self.phi)
            T = R["N"].reshape(self.na * self.nvl, -1)
            Uc = U[self.cell_dofs2]                             # (nc,na,d)
            for k in range(d):
                for e in range(d):
                    ug = Uc[:, :, k, None] * self.gradlam[:, None, :, e]
                    ug *= self.area[:, None, None]
                    vals[..., k, e] += (ug.reshape(-1, self.na * self.nvl)
                                        @ T).reshape(S.shape)
        return pat.assemble(vals)

    def assemble_Mu(self, scale=1.0):
        """Velocity mass matrix scale*(u, v) on the decoupled pattern."""
        pat = self._patterns(False)["A00"]
        M = scale * self.p2_mass_cells()
        if isinstance(pat, BlockPattern):
            return pat.assemble_components([M] * self.dim)
        return pat.assemble(np.repeat(M[..., None], self.dim, axis=3))

    def assemble_A01(self):
        """Discrete gradient block from ``-p div v`` (rows velocity)."""
        pat = self._patterns(False)["A01"]
        # vals[c, a, j, comp] = -int psi_j d_comp phi_a
        return pat.assemble(self._a01_cells())

    def assemble_A10(self):
        """Divergence block from ``-q div u`` (rows pressure) = A01^T."""
        pat = self._patterns(False)["A10"]
        return pat.assemble(self._a01_cells())

    # ------------------------------------------------------------- pressure
    def assemble_Mp(self, scale):
        pat = self._patterns(False)["PP"]
        return pat.assemble((scale * self.area)[:, None, None]
                            * self._ref()["P"][None])

    def assemble_Ap(self):
        pat = self._patterns(False)["PP"]
        g = self.gradlam
        return pat.assemble(self.area[:, None, None]
                            * np.matmul(g, g.transpose(0, 2, 1)))

    def assemble_Kp(self, nu, U, idt=0.0, robin_edges=None):
        """(1/nu) * (w.grad p, q) [+ (idt/nu) (p, q)]
        [- (1/nu) int_{robin_edges} (w.n) p q ds]  (BRM2 boundary term,
        demo_navier-stokes-pcd.py:131-135)."""
        pat = self._patterns(False)["PP"]
        # vals[c,i,j] = |T| sum_m (U_m . grad lam_j) int psi_i phi_m / nu
        ug = np.matmul(U[self.cell_dofs2], self.gradlam.transpose(0, 2, 1))
        vals = np.matmul(self._ref()["S"].T[None], ug) \
            * (self.area / nu)[:, None, None]
        if idt:
            vals = vals + (idt / nu) * self.area[:, None, None] \
                * self._ref()["P"][None]
        K = pat.assemble(vals)
        if robin_edges is not None and len(robin_edges):
            R = self._boundary_flux_mass(U, robin_edges)
            K = pat.matrix(K.data - R.data / nu)
        return K

    def robin_plan(self, edges):
        """Geometry of the boundary integral over ``edges``: P2 nodes (2D:
        start, end, midpoint of each boundary edge), outward unit normals,
        lengths, pressure dofs of the end points; 3D: boundary faces, below."""
        key = ("_robin", tuple(np.asarray(edges).tolist()))
        if getattr(self, "_robin_cache", (None,))[0] == key:
            return self._robin_cache[1]
        m = self.mesh
        if self.dim == 3:
            # ``edges`` index the mesh's boundary FACES: P2 nodes (three
            # vertices, then the midpoints of the edges 01, 02, 12), outward
            # unit normals, areas ("length"), pressure dofs of the vertices.
            # The form is the reference's, dimension-free:
            # demo_navier-stokes-pcd.py:131-135
            faces = np.asarray(edges)
            fv = m.boundary_faces[faces]                            # (nb,3)
            a, b, c = (m.vertices[fv[:, k]] for k in range(3))
            cr = np.cross(b - a, c - a)
            twice = np.linalg.norm(cr, axis=1)
            n = cr / twice[:, None]
            inward = m.vertices[m.boundary_face_opposite[faces]] - a
            n = n * np.where((inward * n).sum(axis=1) > 0, -1.0, 1.0)[:, None]
            r = self._rank
            e01 = m.edge_index(fv[:, 0], fv[:, 1])
            e02 = m.edge_index(fv[:, 0], fv[:, 2])
            e12 = m.edge_index(fv[:, 1], fv[:, 2])
            nodes = np.stack([r[fv[:, 0]], r[fv[:, 1]], r[fv[:, 2]],
                              r[self.nv + e01], r[self.nv + e02],
                              r[self.nv + e12]], axis=1)
            plan = {"nodes": nodes, "normal": n, "length": 0.5 * twice,
                    "pdofs": self._pnum[fv]}
            self._robin_cache = (key, plan)
            return plan
        edges = np.asarray(edges)
        ev = m.edges[edges]                                     # (nb,2)
        a, b = m.vertices[ev[:, 0]], m.vertices[ev[:, 1]]
        t = b - a
        length = np.linalg.norm(t, axis=1)
        n = np.stack([t[:, 1], -t[:, 0]], axis=1) / length[:, None]
        # orient outward: the single adjacent cell's opposite vertex is inside
        cell_of = {}
        flat = m.cell_edges.ravel()
        pos = np.nonzero(np.isin(flat, edges))[0]
        for q in pos:
            cell_of[flat[q]] = (q // 3, q % 3)
        opp = np.array([m.cells[cell_of[e][0], cell_of[e][1]] for e in edges])
        inward = m.vertices[opp] - a
        sgn = np.where((inward * n).sum(axis=1) > 0, -1.0, 1.0)
        n = n * sgn[:, None]
        r = self._rank
        nodes = np.stack([r[ev[:, 0]], r[ev[:, 1]], r[self.nv + edges]], axis=1)
        plan = {"nodes": nodes, "normal": n, "length": length,
                "pdofs": self._pnum[ev]}
        self._robin_cache = (key, plan)
        return plan

    def _boundary_flux_mass(self, U, edges):
        """int_edges (w.n) p q ds as an n_p x n_p matrix on the PP pattern
        (2D: boundary facets are edges)."""
        pl = self.robin_plan(edges)
        pat = self._patterns(False)["PP"]
        n, length, nodes = pl["normal"], pl["length"], pl["nodes"]
        if self.dim == 3:
            # facets are triangles: P2 wind from its six facet nodes, P1 test
            # and trial functions = the barycentric coordinates
            L = _T6L                                             # (q, 3)
            phi = np.concatenate([L * (2 * L - 1),
                                  4 * np.stack([L[:, 0] * L[:, 1],
                                                L[:, 0] * L[:, 2],
                                                L[:, 1] * L[:, 2]], axis=1)],
                                 axis=1)                         # (q, 6)
            un = (U[nodes] * n[:, None, :]).sum(axis=2)          # (nb, 6)
            wn = un @ phi.T                                      # (nb, q)
            loc = np.einsum('q,eq,qi,qj->eij', _T6Wq, wn, L, L) \
                * length[:, None, None]
            pd = pl["pdofs"]                                     # (nb,3)
            rows = np.repeat(pd[:, :, None], 3, axis=2).ravel()
            cols = np.repeat(pd[:, None, :], 3, axis=1).ravel()
            data = np.bincount(pat.locate(rows, cols), weights=loc.ravel(),
                               minlength=pat.nnz)
            return pat.matrix(data)
        # P2 wind along the edge: endpoints + midpoint dofs
        Ua, Ub, Um = U[nodes[:, 0]], U[nodes[:, 1]], U[nodes[:, 2]]
        s = _G3X
        la, lb = 1 - s, s                                       # (3,)
        wn = ((la * (2 * la - 1))[None, :] * (Ua * n).sum(1)[:, None]
              + (lb * (2 * lb - 1))[None, :] * (Ub * n).sum(1)[:, None]
              + (4 * la * lb)[None, :] * (Um * n).sum(1)[:, None])   # (nb,3)
        psi = np.stack([la, lb], axis=1)                        # (3q, 2)
        loc = np.einsum('q,eq,qi,qj->eij', _G3W, wn, psi, psi) \
            * length[:, None, None]
        pd = pl["pdofs"]                                        # (nb,2)
        rows = np.repeat(pd[:, :, None], 2, axis=2).ravel()
        cols = np.repeat(pd[:, None, :], 2, axis=1).ravel()
        data = np.bincount(pat.locate(rows, cols), weights=loc.ravel(),
                           minlength=pat.nnz)
        return pat.matrix(data)

    # ------------------------------------------------------- stabilisation
    def supg_delta(self, U, nu, rho=1.0):
        """Streamline-diffusion parameter per cell, formula of
        ``fenapack/stabilization.py:66-67`` evaluated at the cell midpoint:
        ``Pe = 0.5*|w|*h*rho/nu; delta = Pe>1 ? 0.5*h*(1-1/Pe)/|w| : 0``."""
        lam = np.full((1, self.nvl), 1.0 / self.nvl)
        phi, _ = _p2_basis(lam, self.local_edges)
        wmid = np.tensordot(U[self.cell_dofs2], phi[0], axes=([1], [0]))
        wnorm = np.linalg.norm(wmid, axis=1)
        h = self.cell_h
        with np.errstate(divide='ignore', invalid='ignore'):
            pe = 0.5 * wnorm * h * rho / nu
            delta = np.where(pe > 1.0, 0.5 * h * (1.0 - 1.0 / pe) / wnorm, 0.0)
        return np.nan_to_num(delta)

    # ----------------------------------------------------------- utilities
    def to_mixed(self, xu, xp):
        x = np.empty(self.ndof)
        x[self.is_u] = xu
        x[self.is_p] = xp
        return x

    def monolithic(self, A00, A01, A10):
        """Scatter the 2x2 blocks into the node-major mixed numbering on a
        fixed pattern (explicit zero diagonal kept on the pressure rows, as
        DOLFIN's ``keep_diagonal`` would)."""
        from .. import _host
        key = "_mono_%d_%d_%d" % (A00.nnz, A01.nnz, A10.nnz)
        iu, ip = self.is_u, self.is_p
        if not hasattr(self, key):
            if _host.use_numpy():
                def rc(M, ri, ci):
                    rows = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
                    return ri[rows], ci[M.indices]
                r0, c0 = rc(A00, iu, iu)
                r1, c1 = rc(A01, iu, ip)
                r2, c2 = rc(A10, ip, iu)
                rows = np.concatenate([r0, r1, r2, ip])
                cols = np.concatenate([c0, c1, c2, ip])
                # every (row, col) of the blocks is unique: assembly is a pure
                # permutation (pat.order), gathered instead of summed
                pat = FixedPattern.from_unique(rows, cols,
                                               (self.ndof, self.ndof))
            else:
                eye = np.arange(self.n_p + 1, dtype=np.int32)
                indptr, indices, order = _host.union_blocks(self.ndof, [
                    (iu, iu, A00.indptr, A00.indices),
                    (iu, ip, A01.indptr, A01.indices),
                    (ip, iu, A10.indptr, A10.indices),
                    (ip, ip, eye, eye[:-1])])
                pat = FixedPattern.from_csr(indptr, indices, None,
                                            (self.ndof, self.ndof))
                pat.order = order            # (inv follows on demand)
            setattr(self, key, pat)
        pat = getattr(self, key)
        # (an assembly that is a pure permutation: the blocks' values gathered
        # into the matrix's order on threads, no concatenated copy)
        return pat.matrix(_host.take_segments(
            pat.order, [A00.data, A01.data, A10.data, np.zeros(self.n_p)]))
