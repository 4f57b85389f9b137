"""Benchmark flow problems: forms, boundary conditions, linearised systems.

``BackwardStep`` restates the reference demos' problem definition
(``demo/navier-stokes-pcd/demo_navier-stokes-pcd.py:56-137``: boundary
markers 0/1/2 = no-slip/inlet/outlet, parabolic inflow ``4y(1-y)``, PCD
Dirichlet BC on the inlet for BRM1 and on the outlet for BRM2, Robin term on
the inlet in ``kp`` for BRM2).  ``Cavity`` is the enclosed lid-driven flow
BASELINE.json names (no counterpart in the reference).

The linear systems are produced in the Newton-update form DOLFIN's
``NewtonSolver`` uses (``fenapack/nonlinear_solvers.py:53-60``):
``J(x) dx = F(x)``, ``x <- x - dx``, with Dirichlet conditions imposed
symmetrically as ``SystemAssembler`` does (``fenapack/assembling.py:98-100,
151-155``).
"""

import numpy as np
import scipy.sparse as sp

from .mesh import Mesh, unit_square_mesh
from .mesh import _LSHAPE_VERTICES, _LSHAPE_CELLS
from .multigrid import (MeshHierarchy, CubeHierarchy, Interpolations,
                        injection_map)
from .taylor_hood import TaylorHood
from .. import _host



def _matvec(A, x):
    """``A @ x``; large operators through the threaded native SpMV (bitwise
    scipy's row sums) - scipy's own is one thread, and at config 5's size the
    five products of a residual were ~1.5 of the 3.4 s of a linearisation."""
    if A.nnz > 400000 and not _host.use_numpy():
        return _host.SpMV(A)(x)
    return A @ x

class _Dirichlet(object):
    """Symmetric elimination of a dof set on fixed-pattern CSR blocks."""

    def __init__(self, n, idx):
        self.idx = np.asarray(idx, dtype=np.int64)
        self.flag = np.zeros(n, dtype=bool)
        self.flag[self.idx] = True
        self._cache = {}

    def _masks(self, M, rows_bc, cols_bc, tag):
        key = (tag, M.nnz, M.shape)
        if key not in self._cache:
            rows = np.repeat(np.arange(M.shape[0]), np.diff(M.indptr))
            kill = np.zeros(M.nnz, dtype=bool)
            if rows_bc:
                kill |= self.flag[rows]
            if cols_bc:
                kill |= self.flag[M.indices]
            diag = None
            if rows_bc and cols_bc:
                diag = np.nonzero((rows == M.indices) & self.flag[rows])[0]
            self._cache[key] = (kill, diag)
        return self._cache[key]

    def square(self, M, diag_values=None):
        """Zero bc rows and columns, unit diagonal (pattern preserved);
        ``diag_values`` (per dof) replaces the unit diagonal."""
        kill, diag = self._masks(M, True, True, "sq")
        data = np.where(kill, 0.0, M.data)
        data[diag] = 1.0 if diag_values is None else \
            diag_values[M.indices[diag]]
        return sp.csr_matrix((data, M.indices, M.indptr), shape=M.shape)

    def rows(self, M):
        kill, _ = self._masks(M, True, False, "r")
        return sp.csr_matrix((np.where(kill, 0.0, M.data), M.indices,
                              M.indptr), shape=M.shape)

    def cols(self, M):
        kill, _ = self._masks(M, False, True, "c")
        return sp.csr_matrix((np.where(kill, 0.0, M.data), M.indices,
                              M.indptr), shape=M.shape)


class FlowProblem(object):
    """Common machinery; subclasses define geometry and boundary data."""

    def __init__(self, hierarchy, nu, variant="BRM1", nls="picard", dt=None,
                 pcdr=False, stabilize=False, dirichlet_diag="unit",
                 coarse_stabilize=False, skeleton=False, local_of=None):
        """``skeleton``: numbering and boundary classification only - no
        operator is assembled (what every rank of a partitioned build holds
        of the WHOLE problem).  ``local_of = (G, sub)``: the problem on the
        sub-space ``sub`` (``partition.SubSpace``: the cells this rank
        assembles) of the skeleton ``G``; its boundary sets are G's, mapped
        to the local numbering (``fem/partition.py``)."""
        assert variant in ("BRM1", "BRM2")
        assert nls in ("picard", "newton")
        self.hierarchy = hierarchy
        self._init_kw = dict(nu=nu, variant=variant, nls=nls, dt=dt, pcdr=pcdr,
                             stabilize=stabilize, dirichlet_diag=dirichlet_diag,
                             coarse_stabilize=coarse_stabilize)
        if local_of is not None:
            self.space = V = local_of[1].V
        else:
            self.space = V = TaylorHood(hierarchy.finest)
        self.nu = float(nu)
        self.variant = variant
        self.nls = nls
        self.idt = 0.0 if dt is None else 1.0 / float(dt)
        self.pcdr = pcdr
        self.stabilize = stabilize
        # SUPG on the re-discretised COARSE multigrid levels only (cells with
        # Peclet number > 1 there; the finest level and the operator itself
        # stay unstabilised): keeps `-pc_mg_galerkin none` hierarchies stable
        self.coarse_stabilize = coarse_stabilize
        self.t = 0.0
        # [ext DOLFIN] SystemAssembler writes 1 on the diagonal of a Dirichlet
        # row once per cell tensor, so the assembled diagonal (and the
        # right-hand side entry) of such a row is the number of cells sharing
        # the dof.  "multiplicity" reproduces that scaling (it changes the
        # residual norms the nonlinear stopping test sees, not the solution).
        assert dirichlet_diag in ("unit", "multiplicity")
        mult = np.bincount(V.cell_dofs2.ravel(), minlength=V.nn).astype(float)
        self._bc_mult = np.repeat(mult, V.dim) \
            if dirichlet_diag == "multiplicity" else np.ones(V.n_u)
        if local_of is not None:
            self._localise_boundary(*local_of)
        else:
            self._classify_boundary()
        self.u0 = np.zeros(V.n_u)
        if skeleton:
            return
        self.bc_u = _Dirichlet(V.n_u, self.bc_u_idx)
        self.bc_p = _Dirichlet(V.n_p, self.bc_p_idx)
        # constant operators (assembling.py:98-106: ap, mp, mu, gp constant)
        self._A01_raw = V.assemble_A01()
        self._A10_raw = V.assemble_A10()
        self.A01 = self.bc_u.rows(self._A01_raw)
        self.A10 = self.bc_u.cols(self._A10_raw)
        self.Mp = V.assemble_Mp(1.0 / self.nu)
        self.Ap = self.bc_p.square(V.assemble_Ap())
        self._Mmass = None
        if self.idt:
            self._Mmass = V.assemble_Mu(1.0)
        V.interpolations = self.interpolations     # for pc_type mg

    def _localise_boundary(self, G, sub):
        """Boundary sets of the skeleton ``G`` in the numbering of the
        sub-space (the sub-mesh's own boundary contains the artificial cut)."""
        self._bc_nodes = sub.local_nodes(G._bc_nodes)
        for name in ("_inlet_nodes",):
            if hasattr(G, name):
                setattr(self, name, sub.local_nodes(getattr(G, name)))
        self.bc_u_idx = self._velocity_dofs(self._bc_nodes)
        keep, self.bc_p_idx = sub.local_pdofs(G.bc_p_idx, with_mask=True)
        self.bc_p_val = np.asarray(G.bc_p_val)[keep]
        for name in ("robin_edges", "inlet_edges", "outlet_edges"):
            if hasattr(G, name):
                setattr(self, name, sub.local_facets(getattr(G, name)))

    def interpolations(self):
        """Prolongation chains for the multigrid inner solves (lazy)."""
        if not hasattr(self, "_interp"):
            self._interp = Interpolations(self.hierarchy, self.space)
        return self._interp

    def coarse_velocity_operators(self, xu, nlev):
        """Re-discretised velocity blocks of the preconditioner matrix on the
        ``nlev - 1`` levels below the finest one (coarsest first), for
        ``pc_mg_galerkin none``: the iterate is injected to the coarse nodes
        and every level is assembled with its own SUPG parameter, so coarse
        levels stay stable at cell Peclet numbers > 1 where Galerkin products
        of the fine operator are not (BASELINE config 3)."""
        chain = self.interpolations().velocity
        L = len(chain) - 1                              # finest level index
        ops, x_l = [], xu
        if not hasattr(self, "_coarse_problems"):
            self._coarse_problems = {}
        if not hasattr(self, "_inject"):
            self._inject = {}
        for l in range(L, L - nlev + 1, -1):            # level l -> l - 1
            if l not in self._inject:
                self._inject[l] = injection_map(chain[l], self.space.dim)
            d = self.space.dim
            x_c = x_l.reshape(-1, d)[self._inject[l]].ravel()   # injection
            if l - 1 not in self._coarse_problems:
                self._coarse_problems[l - 1] = self._same_problem_on_level(l - 1)
            pc = self._coarse_problems[l - 1]
            pc.t = self.t
            lin = pc.linearise(x_c, np.zeros(pc.space.n_p))
            ops.append(lin.get("P00", lin["A00"]))
            x_l = x_c
        return ops[::-1]

    def _same_problem_on_level(self, level):
        kw = dict(nu=self.nu, variant=self.variant, nls=self.nls,
                  pcdr=self.pcdr,
                  stabilize=self.stabilize or self.coarse_stabilize)
        if self.idt:
            kw["dt"] = 1.0 / self.idt
        return type(self)(level, **kw)

    # -- helpers -----------------------------------------------------------
    def _edge_dofs_u(self, edges):
        """Velocity-local dofs (both components) living on ``edges``."""
        V, m = self.space, self.space.mesh
        nodes = np.unique(np.concatenate([
            V._rank[m.edges[edges].ravel()], V._rank[V.nv + edges]]))
        return nodes

    def _edge_dofs_p(self, edges):
        V, m = self.space, self.space.mesh
        return np.unique(V._pnum[m.edges[edges].ravel()])

    def nodal_velocity(self, xu):
        return xu.reshape(-1, self.space.dim)

    def _velocity_dofs(self, nodes):
        """All components of the given P2 nodes (interleaved numbering)."""
        d = self.space.dim
        return (d * np.asarray(nodes)[:, None] + np.arange(d)).ravel()

    # -- operators refreshed every nonlinear iteration ----------------------
    def Kp(self, xu):
        V = self.space
        U = self.nodal_velocity(xu)
        robin = self.robin_edges if self.variant == "BRM2" else None
        idt = 0.0 if self.pcdr else self.idt
        return V.assemble_Kp(self.nu, U, idt=idt, robin_edges=robin)

    def Mu(self):
        """``mu = idt * (u, v)`` of the PCDR demo
        (demo_unsteady-navier-stokes-pcdr.py:137)."""
        return self.space.assemble_Mu(self.idt if self.idt else 1.0)

    def linearise(self, xu, xp):
        """Blocks and right-hand side of ``J dx = F`` at the iterate ``x``.

        Returns dict with A00, A01, A10 (BCs applied), optional P00
        (stabilised 00-block for the preconditioner), bu, bp.
        """
        V = self.space
        U = self.nodal_velocity(xu)
        A00p = V.assemble_A00(self.nu, U, idt=self.idt, newton=False)
        Fu = _matvec(A00p, xu) + _matvec(self._A01_raw, xp)
        if self.idt:
            Fu -= self.idt * _matvec(self._Mmass, self.u0)
        Fp = _matvec(self._A10_raw, xu)
        if self.nls == "newton":
            A00 = V.assemble_A00(self.nu, U, idt=self.idt, newton=True)
        else:
            A00 = A00p
        g = self.bc_u_values(self.t)
        d = np.zeros(V.n_u)
        d[self.bc_u_idx] = xu[self.bc_u_idx] - g
        Fu = Fu - _matvec(A00, d)
        Fp = Fp - _matvec(self._A10_raw, d)
        Fu[self.bc_u_idx] = self._bc_mult[self.bc_u_idx] * d[self.bc_u_idx]
        out = {"A00": self.bc_u.square(A00, self._bc_mult), "A01": self.A01,
               "A10": self.A10, "bu": Fu, "bp": Fp}
        if self.stabilize:
            delta = V.supg_delta(U, self.nu)
            P00 = V.assemble_A00(self.nu, U, idt=self.idt,
                                 newton=(self.nls == "newton"), delta=delta)
            out["P00"] = self.bc_u.square(P00, self._bc_mult)
        return out

    def initial_guess(self):
        V = self.space
        return np.zeros(V.n_u), np.zeros(V.n_p)

    def Rp(self):
        """``B diag(Mu)^-1 B^T`` with BC-constrained ``B^T`` = A01
        (field_split_backend.py:142-166; math.rst:157-166)."""
        d = self.Mu().diagonal()
        s = np.sqrt(np.abs(1.0 / d))
        T = sp.diags(s) @ self.A01
        R = (T.T @ T).tocsr()
        R.sort_indices()
        return R


class BackwardStep(FlowProblem):
    """Reference geometry (L-shape) with inflow/outflow."""

    def __init__(self, level, nu=0.02, **kw):
        self.level = level
        base = Mesh(_LSHAPE_VERTICES.copy(), _LSHAPE_CELLS.copy())
        FlowProblem.__init__(self, MeshHierarchy(base, level), nu, **kw)

    def _classify_boundary(self):
        V, m = self.space, self.space.mesh
        mid = m.edge_midpoints()[m.boundary_edges]
        inlet = m.boundary_edges[np.abs(mid[:, 0] + 1.0) < 1e-12]
        outlet = m.boundary_edges[np.abs(mid[:, 0] - 5.0) < 1e-12]
        walls = np.setdiff1d(m.boundary_edges, np.concatenate([inlet,
                                                               outlet]))
        self.inlet_edges, self.outlet_edges = inlet, outlet
        self.robin_edges = inlet                       # ds(1)
        # DOLFIN marks all boundary facets 0 first, then 1/2 override
        # (demo :68-71); bc0 on marker 0 and bc1 on marker 1, bc1 last.
        wall_nodes = self._edge_dofs_u(walls)
        inlet_nodes = self._edge_dofs_u(inlet)
        nodes = np.union1d(wall_nodes, inlet_nodes)
        self._bc_nodes = nodes
        self._inlet_nodes = inlet_nodes
        self.bc_u_idx = self._velocity_dofs(nodes)
        pe = inlet if self.variant == "BRM1" else outlet
        self.bc_p_idx = self._edge_dofs_p(pe)
        self.bc_p_val = np.zeros(self.bc_p_idx.size)

    def inflow_scale(self, t):
        if self.idt:
            return 1.0 - np.exp(-5.0 * t)      # unsteady demo :85-86
        return 1.0

    def bc_u_values(self, t):
        V = self.space
        nodes = self._bc_nodes
        val = np.zeros((nodes.size, 2))
        is_in = np.isin(nodes, self._inlet_nodes)
        y = V.node_coords[nodes[is_in], 1]
        val[is_in, 0] = self.inflow_scale(t) * 4.0 * y * (1.0 - y)
        return val.ravel()


class Cavity(FlowProblem):
    """Lid-driven unit-square cavity (watertight lid), ``nu = 1/Re``."""

    def __init__(self, level, nu=0.01, n=None, **kw):
        self.level = level
        base = unit_square_mesh(5 if n is None else n)
        FlowProblem.__init__(self, MeshHierarchy(base, level), nu, **kw)

    def _classify_boundary(self):
        V, m = self.space, self.space.mesh
        nodes = self._edge_dofs_u(m.boundary_edges)
        self._bc_nodes = nodes
        self.bc_u_idx = self._velocity_dofs(nodes)
        self.robin_edges = np.zeros(0, dtype=np.int64)
        # enclosed flow has no inlet/outlet: pin the Laplacian at one vertex
        # so that A_p is SPD (hard part 2)
        c = V.p_coords
        self.bc_p_idx = np.array([int(np.argmin(c[:, 0] + c[:, 1]))])
        self.bc_p_val = np.zeros(1)

    def bc_u_values(self, t):
        V = self.space
        xy = V.node_coords[self._bc_nodes]
        val = np.zeros((xy.shape[0], 2))
        lid = (np.abs(xy[:, 1] - 1.0) < 1e-12) & (xy[:, 0] > 1e-12) \
            & (xy[:, 0] < 1.0 - 1e-12)
        val[lid, 0] = 1.0
        return val.ravel()


class Cavity3D(FlowProblem):
    """Lid-driven unit-cube cavity (BASELINE config 5), P2/P1 on the Kuhn
    triangulation of ``n0 * 2**level`` cubes per side; lid ``z = 1`` moves
    with ``u = (1, 0, 0)`` (watertight: the lid's rim belongs to the walls)."""

    def __init__(self, level, nu=0.01, n0=2, **kw):
        self.level, self.n0 = level, n0
        FlowProblem.__init__(self, CubeHierarchy(n0, level), nu, **kw)

    def _same_problem_on_level(self, level):
        kw = dict(nu=self.nu, variant=self.variant, nls=self.nls,
                  pcdr=self.pcdr, n0=self.n0,
                  stabilize=self.stabilize or self.coarse_stabilize)
        if self.idt:
            kw["dt"] = 1.0 / self.idt
        return Cavity3D(level, **kw)

    def _classify_boundary(self):
        V, m = self.space, self.space.mesh
        nodes = np.unique(np.concatenate([
            V._rank[m.boundary_vertices], V._rank[V.nv + m.boundary_edges]]))
        self._bc_nodes = nodes
        self.bc_u_idx = self._velocity_dofs(nodes)
        self.robin_edges = np.zeros(0, dtype=np.int64)
        c = V.p_coords
        self.bc_p_idx = np.array([int(np.argmin(c.sum(axis=1)))])
        self.bc_p_val = np.zeros(1)

    def bc_u_values(self, t):
        V = self.space
        xyz = V.node_coords[self._bc_nodes]
        val = np.zeros((xyz.shape[0], 3))
        eps = 1e-12
        lid = (np.abs(xyz[:, 2] - 1.0) < eps) \
            & (xyz[:, 0] > eps) & (xyz[:, 0] < 1 - eps) \
            & (xyz[:, 1] > eps) & (xyz[:, 1] < 1 - eps)
        val[lid, 0] = 1.0
        return val.ravel()


class Channel3D(FlowProblem):
    """Square duct on the unit cube: inflow through ``x = 0`` (marker 1),
    do-nothing outflow through ``x = 1`` (marker 2), no-slip walls - the
    reference demo's in/outflow set-up (demo_navier-stokes-pcd.py:56-87) in
    three dimensions, where its forms are dimension-free: BRM1 pins the
    pressure operators on the inlet, BRM2 on the outlet and adds the Robin
    term ``-(1/nu) (u.n) p q ds(1)`` to ``kp`` (:131-135).  Kuhn triangulation
    of ``n0 * 2**level`` cubes per side."""

    def __init__(self, level, nu=0.02, n0=2, **kw):
        self.level, self.n0 = level, n0
        FlowProblem.__init__(self, CubeHierarchy(n0, level), nu, **kw)

    def _same_problem_on_level(self, level):
        kw = dict(nu=self.nu, variant=self.variant, nls=self.nls,
                  pcdr=self.pcdr, n0=self.n0,
                  stabilize=self.stabilize or self.coarse_stabilize)
        if self.idt:
            kw["dt"] = 1.0 / self.idt
        return Channel3D(level, **kw)

    def _face_nodes_u(self, faces):
        """P2 nodes (vertices and edge midpoints) of the given boundary faces."""
        V, m = self.space, self.space.mesh
        fv = m.boundary_faces[faces]
        e = np.concatenate([m.edge_index(fv[:, 0], fv[:, 1]),
                            m.edge_index(fv[:, 0], fv[:, 2]),
                            m.edge_index(fv[:, 1], fv[:, 2])])
        return np.unique(np.concatenate([V._rank[fv.ravel()],
                                         V._rank[V.nv + e]]))

    def _classify_boundary(self):
        V, m = self.space, self.space.mesh
        cx = m.boundary_face_centroids()[:, 0]
        allf = np.arange(m.boundary_faces.shape[0])
        inlet, outlet = allf[np.abs(cx) < 1e-12], allf[np.abs(cx - 1.0) < 1e-12]
        walls = np.setdiff1d(allf, np.concatenate([inlet, outlet]))
        self.inlet_edges, self.outlet_edges = inlet, outlet      # (boundary FACES in 3-D)
        self.robin_edges = inlet                                 # ds(1)
        # (bc0 on the walls first, bc1 on the inlet last: demo :68-71)
        wall_nodes, inlet_nodes = self._face_nodes_u(walls), self._face_nodes_u(inlet)
        nodes = np.union1d(wall_nodes, inlet_nodes)
        self._bc_nodes = nodes
        self._inlet_nodes = np.setdiff1d(inlet_nodes, wall_nodes)   # (the rim is wall: no-slip)
        self.bc_u_idx = self._velocity_dofs(nodes)
        pe = inlet if self.variant == "BRM1" else outlet
        self.bc_p_idx = np.unique(V._pnum[m.boundary_faces[pe].ravel()])
        self.bc_p_val = np.zeros(self.bc_p_idx.size)

    def inflow_scale(self, t):
        if self.idt:
            return 1.0 - np.exp(-5.0 * t)
        return 1.0

    def bc_u_values(self, t):
        V = self.space
        nodes = self._bc_nodes
        val = np.zeros((nodes.size, 3))
        is_in = np.isin(nodes, self._inlet_nodes)
        y, z = V.node_coords[nodes[is_in], 1], V.node_coords[nodes[is_in], 2]
        val[is_in, 0] = self.inflow_scale(t) * 16.0 * y * (1 - y) * z * (1 - z)
        return val.ravel()
