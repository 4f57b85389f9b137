"""Partitioned operator producer: every rank assembles ITS rows only.

The reference's inputs arrive partitioned - each rank owns ``dofmap.dofs()``
only (``fenapack/_field_split_utils.py:39-50``), boundary maps are built over
owned indices only (``fenapack/SubfieldBC.h:136-155``) and DOLFIN assembles
each rank's cells.  This module gives the engine's producer the same shape:

* what every rank holds of the WHOLE problem is light: the mesh topology, the
  dof numbering and the boundary classification (a ``skeleton`` problem:
  ~0.2 KB per cell against ~25 KB per cell of a full build);
* the heavy part - sparsity patterns, element matrices, assembled operators,
  the finest prolongation - exists only for the cells that touch this rank's
  rows (its slab plus one layer of halo cells): a ``SubSpace`` is a
  ``TaylorHood`` space on that sub-mesh whose numbering is the global one
  restricted (same sort keys, so local -> global is monotone), and the local
  ``FlowProblem`` on it reuses every form of ``fem/problems.py`` unchanged;
* the rows a rank owns are the engine's (``pcd_row_range``: even cuts, on node
  boundaries for velocities - ``cut`` below restates ``pcd_dist.hpp``
  ``Space::cut``), so what is assembled goes straight into
  ``pcd_set_system_local`` / ``pcd_set_csr_local`` /
  ``pcd_mg_set_level_local``; no global CSR of a partitioned level exists
  anywhere.

Operators cross the Python stack as GLOBAL-SHAPED, ROW-SPARSE CSR matrices:
the shape is the global one, only the owned rows are populated (memory: local
entries + one row pointer).  The existing stack (forms, ``PCDAssembler``,
``PCDInterface``, the ``petsc`` shim) slices rows ``[r0, r1)`` of what it is
given, so it runs unchanged; the three places that need other ranks' rows
(residual norms, smoother eigenvalue estimates, the tiny Galerkin coarse
operator of ``A_p``) reduce through a ``HostComm``.

Because local cells keep their global order and element matrices are computed
cell by cell, the owned rows are BITWISE the rows of the global build
(``tests/test_partition.py``).
"""

import os
import threading

import numpy as np
import scipy.sparse as sp

from .. import _host
from .taylor_hood import TaylorHood, _p2_basis, small_det_inv
from .multigrid import _unique_entries, interleave


def cut(n, R, block=1):
    """Row cuts of ``n`` rows over ``R`` ranks - ``Space::cut`` of
    ``csrc/pcd_dist.hpp`` (what ``pcd_row_range`` answers)."""
    b = [n * r // R for r in range(R + 1)]
    if block > 1:
        b = [v - v % block for v in b]
    b[R] = n
    return b


def replicate_below():
    """Levels of at most this many rows are replicated on every rank
    (``PCD_REPLICATE_BELOW``, the engine's rule in ``pcd_mg_set_level``)."""
    return int(os.environ.get("PCD_REPLICATE_BELOW", "60000"))


# ------------------------------------------------------------------ host comm
class HostComm(object):
    """The few host-side reductions of a partitioned set-up (one rank: all
    identities).  Not a data path: vectors and halos of the solve travel on
    the device (RCCL / peer copies inside the engine)."""

    rank, size = 0, 1

    def allgather(self, obj):
        return [obj]

    def sum(self, a):
        """Sum over ranks (same bits on every rank: rank order)."""
        parts = self.allgather(a)
        out = parts[0] if np.isscalar(parts[0]) else parts[0].copy()
        for q in parts[1:]:
            out = out + q
        return out

    def max(self, x):
        return max(self.allgather(x))

    def sum_rows(self, M, cuts):
        """Rows ``cuts[rank]:cuts[rank + 1]`` of the sum over ranks of the
        global-shaped sparse matrices ``M`` (every rank's contribution to a
        product like ``B^T B`` summed over its own rows of ``B``): each rank
        sends the rows it holds for the others - the few beside its cut -
        and keeps a global-shaped matrix with its own rows filled.  Summed
        in rank order, so the bits do not depend on arrival."""
        M = sp.csr_matrix(M)
        if len(cuts) != self.size + 1 or cuts[-1] != M.shape[0]:
            raise ValueError("sum_rows: %d cuts for %d ranks, %d rows"
                             % (len(cuts), self.size, M.shape[0]))

        def rows(q):
            a, b = int(cuts[q]), int(cuts[q + 1])
            lo, hi = int(M.indptr[a]), int(M.indptr[b])
            if hi == lo:
                return None
            return (M.indptr[a:b + 1] - lo, M.indices[lo:hi], M.data[lo:hi])

        got = self.allgather([None if q == self.rank else rows(q)
                              for q in range(self.size)])
        a, b = int(cuts[self.rank]), int(cuts[self.rank + 1])
        out = None
        for q in range(self.size):
            piece = rows(q) if q == self.rank else got[q][self.rank]
            if piece is None:
                continue
            ip, idx, val = piece
            P = sp.csr_matrix((val, idx, ip), shape=(b - a, M.shape[1]))
            out = P if out is None else out + P
        if out is None:
            out = sp.csr_matrix((b - a, M.shape[1]))
        out = sp.csr_matrix(out)
        out.sort_indices()
        indptr = np.zeros(M.shape[0] + 1, dtype=np.int64)
        indptr[a + 1:b + 1] = out.indptr[1:]
        indptr[b + 1:] = out.indptr[-1]
        return sp.csr_matrix((out.data, out.indices, indptr), shape=M.shape)


class ThreadHostComm(HostComm):
    """R ranks as threads of one process (the thread-rank tests and tools)."""

    class _Shared(object):
        def __init__(self, R):
            self.barrier = threading.Barrier(R)
            self.slots = [None] * R

    def __init__(self, rank, size, shared):
        self.rank, self.size, self._sh = rank, size, shared

    @classmethod
    def group(cls, R):
        sh = cls._Shared(R)
        return [cls(r, R, sh) for r in range(R)]

    def allgather(self, obj):
        sh = self._sh
        sh.slots[self.rank] = obj
        sh.barrier.wait()
        out = list(sh.slots)
        sh.barrier.wait()
        return out


class TorchHostComm(HostComm):
    """One process per GPU: ``torch.distributed`` (a gloo group beside the
    RCCL one - these are host objects)."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self._group = None
        if dist.get_backend() != "gloo":
            self._group = dist.new_group(backend="gloo")

    def allgather(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self._group)
        return out

    def sum(self, a):
        if not isinstance(a, np.ndarray):       # scalars, sparse matrices
            return HostComm.sum(self, a)
        import torch
        t = torch.from_numpy(np.ascontiguousarray(a).copy())
        self._dist.all_reduce(t, group=self._group)
        return t.numpy()


def host_comm(comm=None):
    """The ``HostComm`` that goes with an engine communicator
    (``parallel.Comm``): its ``host`` attribute when it has one, the
    ``torch.distributed`` world when that is initialised, else serial."""
    hc = getattr(comm, "host", None)
    if hc is not None:
        return hc
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() \
                and dist.get_world_size() > 1:
            return TorchHostComm()
    except ImportError:
        pass
    return HostComm()


# ------------------------------------------------------------------ sub-space
def _edge_keys(mesh):
    if not hasattr(mesh, "_edge_keys"):
        mesh._edge_keys = mesh.edges[:, 0] * mesh.num_vertices + mesh.edges[:, 1]
    return mesh._edge_keys


class SubSpace(object):
    """P2 / P1 space on the sub-mesh of ``cells`` (ascending global order) of
    the global space ``Vg``, numbered like ``Vg`` restricted.

    ``nodes_g`` / ``p_g``: global scalar P2 dof / P1 dof of every local one
    (strictly increasing); ``u_g`` / ``mixed_g``: the same for the velocity
    and the mixed numbering."""

    def __init__(self, Vg, cells):
        m = Vg.mesh
        self.Vg = Vg
        self.cells = cells = np.sort(np.asarray(cells, dtype=np.int64))
        gc = m.cells[cells]
        self.vg = vg = np.unique(gc)                  # global vertices, ascending
        lc = np.searchsorted(vg, gc)
        sub = type(m)(m.vertices[vg], lc)
        assert np.array_equal(sub.cells, lc), "sub-mesh re-oriented its cells"
        self.V = V = TaylorHood(sub, axes=Vg.axes)
        nvl, d = sub.num_vertices, V.dim
        ent = np.empty(V.nn, dtype=np.int64)          # mesh entity -> global dof
        ent[:nvl] = Vg._rank[vg]
        le = sub.edges
        gkey = vg[le[:, 0]] * m.num_vertices + vg[le[:, 1]]
        mkey = _edge_keys(m)
        ge = np.searchsorted(mkey, gkey)
        assert np.array_equal(mkey[ge], gkey), "sub-mesh edge not in the mesh"
        self.edges_g = ge                             # local edge -> global edge
        ent[nvl:] = Vg._rank[m.num_vertices + ge]
        self.nodes_g = np.empty(V.nn, dtype=np.int64)
        self.nodes_g[V._rank] = ent
        self.p_g = np.empty(nvl, dtype=np.int64)
        self.p_g[V._pnum] = Vg._pnum[vg]
        assert np.all(np.diff(self.nodes_g) > 0) and np.all(np.diff(self.p_g) > 0), \
            "local numbering is not the global one restricted"
        self.u_g = (d * self.nodes_g[:, None] + np.arange(d)).ravel()
        self.mixed_g = np.empty(V.ndof, dtype=np.int64)
        self.mixed_g[V.is_u] = Vg.is_u[self.u_g]
        self.mixed_g[V.is_p] = Vg.is_p[self.p_g]

    # global ids -> local ids (absent ones dropped)
    @staticmethod
    def _find(table, ids, with_mask=False):
        ids = np.asarray(ids, dtype=np.int64)
        pos = np.searchsorted(table, ids)
        pos[pos == table.size] = 0
        hit = table[pos] == ids if table.size else np.zeros(ids.size, bool)
        return (hit, pos[hit]) if with_mask else pos[hit]

    def local_nodes(self, gnodes):
        return self._find(self.nodes_g, gnodes)

    def local_pdofs(self, gp, with_mask=False):
        return self._find(self.p_g, gp, with_mask)

    def local_edges(self, gedges):
        return self._find(self.edges_g, gedges)

    def local_facets(self, gfacets):
        """Boundary facets of the global mesh (2-D: edge numbers; 3-D: rows of
        ``boundary_faces``) that belong to cells of this sub-mesh, as facets of
        the sub-mesh."""
        m, sub = self.Vg.mesh, self.V.mesh
        if m.dim != 3:
            return self.local_edges(gfacets)
        tri = m.boundary_faces[np.asarray(gfacets, dtype=np.int64)]
        pos = np.searchsorted(self.vg, tri)
        pos[pos == self.vg.size] = 0
        lt = pos[(self.vg[pos] == tri).all(axis=1)]      # (monotone map: still sorted triples)
        nv = sub.num_vertices
        key = (lt[:, 0] * nv + lt[:, 1]) * nv + lt[:, 2]
        bf = sub.boundary_faces
        bkey = (bf[:, 0] * nv + bf[:, 1]) * nv + bf[:, 2]
        order = np.argsort(bkey)
        idx = np.searchsorted(bkey[order], key)
        idx[idx == bkey.size] = 0
        ok = bkey[order][idx] == key if bkey.size else np.zeros(key.size, bool)
        return order[idx[ok]]


class _RowLift(object):
    """Owned rows of local matrices -> global-shaped, row-sparse CSR.  The
    index structure of a pattern is translated once (keyed by the pattern's
    size); later calls only slice the values."""

    def __init__(self):
        self._cache = {}

    def __call__(self, M, row_g, col_g, own, shape, tag):
        M = sp.csr_matrix(M)
        key = (tag, M.nnz, M.shape)
        ent = self._cache.get(key)
        if ent is None:
            a, b = np.searchsorted(row_g, own)
            if b - a != own[1] - own[0] or \
                    (b > a and (row_g[a] != own[0] or row_g[b - 1] != own[1] - 1)):
                raise ValueError("lift(%s): an owned row is missing from the "
                                 "local space" % tag)
            ip = M.indptr.astype(np.int64)
            lo, hi = int(ip[a]), int(ip[b])
            indptr = np.zeros(shape[0] + 1, dtype=np.int64)
            indptr[own[0] + 1:own[1] + 1] = ip[a + 1:b + 1] - lo
            indptr[own[1] + 1:] = hi - lo
            it = np.int32 if max(shape) < 2 ** 31 - 1 and hi - lo < 2 ** 31 - 1 \
                else np.int64
            ent = (lo, hi, indptr.astype(it),
                   col_g[M.indices[lo:hi]].astype(it))
            self._cache[key] = ent
        lo, hi, indptr, indices = ent
        out = sp.csr_matrix((M.data[lo:hi], indices, indptr), shape=shape)
        out.has_sorted_indices = True
        return out


# ------------------------------------------------------------------ the problem
class _Level(object):
    """One multigrid level of a partitioned problem: the global skeleton and,
    when the level is partitioned, this rank's sub-space and local problem."""

    def __init__(self, G, R, rank, partitioned):
        self.G, self.partitioned = G, partitioned
        V = G.space
        d = V.dim
        ub, pb = cut(V.n_u, R, d), cut(V.n_p, R, 1)
        if partitioned:
            self.own_u = (ub[rank], ub[rank + 1])
            self.own_p = (pb[rank], pb[rank + 1])
        else:
            self.own_u, self.own_p = (0, V.n_u), (0, V.n_p)
        self.own_nodes = (self.own_u[0] // d, self.own_u[1] // d)
        self.sub = self.loc = None

    def cells_of_nodes(self, node_mask=None, p_mask=None):
        V = self.G.space
        sel = np.zeros(V.cell_dofs2.shape[0], dtype=bool)
        if node_mask is not None:
            sel |= node_mask[V.cell_dofs2].any(axis=1)
        if p_mask is not None:
            sel |= p_mask[V.cell_dofs1].any(axis=1)
        return sel

    def owned_cells(self):
        """Cells that touch an owned velocity node or an owned pressure dof:
        every cell that contributes to an owned row."""
        V = self.G.space
        nm = np.zeros(V.nn, dtype=bool)
        nm[self.own_nodes[0]:self.own_nodes[1]] = True
        pm = np.zeros(V.n_p, dtype=bool)
        pm[self.own_p[0]:self.own_p[1]] = True
        return np.nonzero(self.cells_of_nodes(nm, pm))[0]


class PartitionedProblem(object):
    """A ``FlowProblem`` of which this rank assembles its rows only.

    ``factory(level=None, **kw)`` builds the problem class with the caller's
    parameters (``skeleton`` / ``local_of`` are passed through ``kw``;
    ``level`` selects a coarser mesh of the same hierarchy).  The object
    offers the attributes the solver stack reads of a ``FlowProblem``
    (``space`` is the GLOBAL space: numbering, index sets, sizes); matrices
    are global-shaped and row-sparse, vectors global-length with the owned
    entries filled."""

    partitioned = True

    def __init__(self, factory, rank, size, host=None):
        self.rank, self.size = int(rank), int(size)
        self.host = host if host is not None else HostComm()
        self._factory = factory
        self.G = G = factory(skeleton=True)
        self.space = V = G.space
        self.hierarchy = G.hierarchy
        for k in ("nu", "variant", "nls", "idt", "pcdr", "stabilize",
                  "coarse_stabilize"):
            setattr(self, k, getattr(G, k))
        self.t = 0.0
        self._u0 = np.zeros(V.n_u)
        self.bc_u_idx, self.bc_p_idx = G.bc_u_idx, G.bc_p_idx
        self.bc_p_val = G.bc_p_val
        self.robin_edges = G.robin_edges
        self._lift = _RowLift()
        self.fine = self._make_level(G, partitioned=True)
        self._levels = {}
        V.interpolations = self.interpolations
        V.partitioned_producer = self

    # -- levels --------------------------------------------------------------
    def _make_level(self, G, partitioned):
        lev = _Level(G, self.size, self.rank, partitioned)
        if partitioned:
            lev.sub = SubSpace(G.space, lev.owned_cells())
            lev.loc = self._local_problem(G, lev.sub)
        return lev

    def _local_problem(self, G, sub):
        loc = object.__new__(type(G))
        for k, v in G.__dict__.items():
            if isinstance(v, (int, float, str, bool)) or v is None:
                setattr(loc, k, v)
        type(G).__mro__[-2].__init__(loc, G.hierarchy, local_of=(G, sub),
                                     **G._init_kw)
        return loc

    def level(self, l, whole=False):
        """Level ``l`` of the hierarchy (finest = ``len(meshes) - 1``);
        ``whole``: never partitioned (the coarsest level of a cycle: its
        explicit inverse is replicated whatever its size)."""
        L = len(self.hierarchy.meshes) - 1
        if l == L:
            return self.fine
        key = (l, bool(whole))
        if key not in self._levels:
            Gl = self._factory(level=l, skeleton=True)
            part = Gl.space.n_u > replicate_below() and not whole
            lev = self._make_level(Gl, part)
            if not part:
                lev.loc = self._factory(level=l)      # whole (small) problem
            self._levels[key] = lev
        return self._levels[key]

    # -- state ---------------------------------------------------------------
    @property
    def u0(self):
        return self._u0

    @u0.setter
    def u0(self, v):
        self._u0 = np.asarray(v, dtype=np.float64)

    def bc_u_values(self, t):
        return self.G.bc_u_values(t)

    def initial_guess(self):
        V = self.space
        return np.zeros(V.n_u), np.zeros(V.n_p)

    def norm(self, b):
        """2-norm of a vector whose owned entries only are filled."""
        b = np.asarray(b)
        return float(np.sqrt(self.host.sum(float(np.dot(b, b)))))

    # -- lifting -------------------------------------------------------------
    def _uu(self, lev, M, tag):
        n = lev.G.space.n_u
        return self._lift(M, lev.sub.u_g, lev.sub.u_g, lev.own_u, (n, n),
                          (id(lev), tag))

    def _pp(self, lev, M, tag):
        n = lev.G.space.n_p
        return self._lift(M, lev.sub.p_g, lev.sub.p_g, lev.own_p, (n, n),
                          (id(lev), tag))

    def _up(self, lev, M, tag):
        V = lev.G.space
        return self._lift(M, lev.sub.u_g, lev.sub.p_g, lev.own_u,
                          (V.n_u, V.n_p), (id(lev), tag))

    def _pu(self, lev, M, tag):
        V = lev.G.space
        return self._lift(M, lev.sub.p_g, lev.sub.u_g, lev.own_p,
                          (V.n_p, V.n_u), (id(lev), tag))

    @staticmethod
    def _owned_vec(n, own, idx_g, v_loc):
        out = np.zeros(n)
        a, b = np.searchsorted(idx_g, own)
        out[own[0]:own[1]] = v_loc[a:b]
        return out

    # -- constant operators --------------------------------------------------
    @property
    def A01(self):
        return self._up(self.fine, self.fine.loc.A01, "A01")

    @property
    def A10(self):
        return self._pu(self.fine, self.fine.loc.A10, "A10")

    @property
    def Mp(self):
        return self._pp(self.fine, self.fine.loc.Mp, "Mp")

    @property
    def Ap(self):
        return self._pp(self.fine, self.fine.loc.Ap, "Ap")

    def Kp(self, xu):
        lev = self.fine
        return self._pp(lev, lev.loc.Kp(np.asarray(xu)[lev.sub.u_g]), "Kp")

    def Mu(self):
        return self._uu(self.fine, self.fine.loc.Mu(), "Mu")

    def Rp(self):
        """Owned rows of ``B diag(Mu)^-1 B^T`` (``FlowProblem.Rp``;
        field_split_backend.py:142-166).  The product is a sum over velocity
        rows: every rank forms the terms of ITS rows of ``B^T`` - they reach
        pressure rows beside its cut too - and ``HostComm.sum_rows`` hands
        each row's terms to its owner (what ``MatTransposeMatMult`` does on
        an MPI matrix)."""
        lev = self.fine
        a, b = lev.own_u
        d = self.Mu().diagonal()[a:b]
        s = np.zeros(self.space.n_u)
        s[a:b] = np.sqrt(np.abs(1.0 / d))
        T = sp.diags(s) @ self.A01
        return self.host.sum_rows((T.T @ T).tocsr(),
                                  cut(self.space.n_p, self.size, 1))

    # -- one linearisation ---------------------------------------------------
    def _sync(self, lev, t=None):
        loc = lev.loc
        loc.t = self.t if t is None else t
        if lev.sub is not None and self.idt:
            loc.u0 = self._u0[lev.sub.u_g] if lev is self.fine else loc.u0

    def linearise(self, xu, xp):
        lev = self.fine
        sub, V = lev.sub, self.space
        self._sync(lev)
        L = lev.loc.linearise(np.asarray(xu)[sub.u_g], np.asarray(xp)[sub.p_g])
        out = {"A00": self._uu(lev, L["A00"], "A00"),
               "A01": self._up(lev, L["A01"], "A01"),
               "A10": self._pu(lev, L["A10"], "A10"),
               "bu": self._owned_vec(V.n_u, lev.own_u, sub.u_g, L["bu"]),
               "bp": self._owned_vec(V.n_p, lev.own_p, sub.p_g, L["bp"])}
        if "P00" in L:
            out["P00"] = self._uu(lev, L["P00"], "P00")
        return out

    # -- multigrid inputs ----------------------------------------------------
    def interpolations(self):
        if not hasattr(self, "_interp"):
            self._interp = PartitionedInterpolations(self)
        return self._interp

    def _finest_keys(self):
        if not hasattr(self, "_fkeys"):
            self._fkeys = _coord_keys(self.space.node_coords, self.space)
        return self._fkeys

    def injected_wind(self, xu, lev):
        """The iterate at the nodes of a coarser level's LOCAL space (nested
        P2 spaces: every coarse node is a finest node) - the composition of
        the level-to-level injections of ``FlowProblem.
        coarse_velocity_operators``."""
        V, d = self.space, self.space.dim
        Vl = lev.loc.space
        keys = _coord_keys(Vl.node_coords, self.space)
        fk = self._finest_keys()
        pos = np.searchsorted(fk, keys)
        assert np.array_equal(fk[pos], keys), "levels are not nested"
        return np.asarray(xu).reshape(-1, d)[pos].ravel()

    def coarse_velocity_operators(self, xu, nlev):
        """Re-discretised velocity blocks of the ``nlev - 1`` levels below the
        finest one, coarsest first (``-pc_mg_galerkin none``, PETSc's PCMG
        default): partitioned levels as owned rows, replicated ones whole.
        No communication: every level is assembled from the iterate."""
        L = len(self.hierarchy.meshes) - 1
        ops = []
        for l in range(L - 1, L - nlev, -1):
            lev = self.level(l, whole=(l == L - nlev + 1))
            self._sync(lev)
            x_l = self.injected_wind(xu, lev)
            lin = lev.loc.linearise(x_l, np.zeros(lev.loc.space.n_p))
            M = lin.get("P00", lin["A00"])
            ops.append(self._uu(lev, M, "cA00") if lev.partitioned else M)
        return ops[::-1]


def _coord_keys(coords, V):
    """One integer per node, ordered like the numbering (the lexicographic
    order of ``TaylorHood``): 20 bits per axis on a 2^-16 lattice."""
    lo = getattr(V, "_key_origin", None)
    if lo is None:
        lo = V._key_origin = np.floor(V.node_coords.min(axis=0)) - 1.0
    q = np.round((coords - lo) * 65536.0).astype(np.int64)
    key = np.zeros(coords.shape[0], dtype=np.int64)
    for a in V.axes:
        key = (key << 20) | q[:, a]
    return key


class PartitionedInterpolations(object):
    """Prolongation chains of a partitioned problem.  Levels below the finest
    one are small (1/2^d of the one above each): their prolongations are built
    whole on every rank; the FINEST prolongation exists as the rows of this
    rank's local nodes (owned + halo: the columns of its operator rows) plus
    the rows in the support of its owned coarse dofs (what the restriction
    rows ``pcd_mg_set_level_local`` takes are read from)."""

    def __init__(self, pp):
        from .multigrid import prolongations
        H, Vf = pp.hierarchy, pp.space
        L = len(H.meshes) - 1
        self.velocity, self.pressure = [None], [None]
        for l in range(1, L):
            P2, P1 = prolongations(H.space(l - 1), H.space(l), H.parents[l])
            self.velocity.append(interleave(P2, Vf.dim))
            self.pressure.append(P1)
        if L >= 1:
            P2, P1 = self._finest(pp, H.space(L - 1), Vf, H.parents[L])
            self.velocity.append(interleave(P2, Vf.dim))
            self.pressure.append(P1)

    @staticmethod
    def _finest(pp, Vc, Vf, parent):
        lev, R, r = pp.fine, pp.size, pp.rank
        d = Vf.dim
        # fine cells: those of the local space, and those in the support of
        # the coarse dofs this rank owns at level L-1 (with every cell that
        # shares a node with them, so each row is the global build's)
        cu, cp = cut(Vc.n_u, R, d), cut(Vc.n_p, R, 1)
        part_c = Vc.n_u > replicate_below()
        nm = np.zeros(Vc.nn, dtype=bool)
        pm = np.zeros(Vc.n_p, dtype=bool)
        if part_c:
            nm[cu[r] // d:cu[r + 1] // d] = True
            pm[cp[r]:cp[r + 1]] = True
        csel = nm[Vc.cell_dofs2].any(axis=1) | pm[Vc.cell_dofs1].any(axis=1)
        fsel = csel[parent]
        fsel[lev.sub.cells] = True
        fn = np.zeros(Vf.nn, dtype=bool)
        fn[Vf.cell_dofs2[fsel].ravel()] = True
        fsel |= fn[Vf.cell_dofs2].any(axis=1)
        cells = np.nonzero(fsel)[0]
        return prolongation_rows(Vc, Vf, parent, cells)

    def chain(self, field, nlevels=None):
        full = self.velocity if field == "u" else self.pressure
        if nlevels is None or nlevels >= len(full):
            return list(full)
        return [None] + full[len(full) - nlevels + 1:]


def prolongation_rows(Vc, Vf, parent, cells):
    """``multigrid.prolongations`` for the fine ``cells`` only (ascending):
    global-shaped (scalar P2, P1) prolongations whose populated rows are the
    dofs of those cells.  A row computed from any cell that contains its node
    is complete; the value kept is the first cell's in global order, as in the
    global build - so rows whose cells are ALL selected are bitwise its."""
    mc, mf = Vc.mesh, Vf.mesh
    d = Vf.dim
    par = parent[cells]
    pc = mc.vertices[mc.cells[par]]
    pf = mf.vertices[mf.cells[cells]]
    mids = np.stack([0.5 * (pf[:, i] + pf[:, j]) for i, j in Vf.local_edges],
                    axis=1)
    pts = np.concatenate([pf, mids], axis=1)
    T = np.stack([pc[:, k + 1] - pc[:, 0] for k in range(d)], axis=2)
    Tinv = small_det_inv(T)[1]
    l1d = np.matmul(pts - pc[:, None, 0, :], Tinv.transpose(0, 2, 1))
    lam = np.concatenate([1.0 - l1d.sum(axis=2, keepdims=True), l1d], axis=2)
    na, nvl = Vf.na, Vf.nvl
    phi, _ = _p2_basis(lam.reshape(-1, nvl), Vc.local_edges, grad=False)
    phi = phi.reshape(lam.shape[0], na, na)
    rows = np.repeat(Vf.cell_dofs2[cells][:, :, None], na, axis=2)
    cols = np.repeat(Vc.cell_dofs2[par][:, None, :], na, axis=1)
    P2 = _unique_entries(rows, cols, phi, (Vf.nn, Vc.nn))
    rows1 = np.repeat(Vf.cell_dofs1[cells][:, :, None], nvl, axis=2)
    cols1 = np.repeat(Vc.cell_dofs1[par][:, None, :], nvl, axis=1)
    P1 = _unique_entries(rows1, cols1, lam[:, :nvl, :], (Vf.n_p, Vc.n_p))
    return P2, P1


def partitioned(cls, rank, size, host=None, **kw):
    """``cls(**kw)`` (``BackwardStep`` / ``Cavity`` / ``Cavity3D``; ``level``
    among ``kw``) as a :class:`PartitionedProblem` of rank ``rank``."""
    base = dict(kw)
    lvl = base.pop("level")

    def factory(level=None, **extra):
        return cls(lvl if level is None else level, **dict(base, **extra))
    return PartitionedProblem(factory, rank, size, host)
