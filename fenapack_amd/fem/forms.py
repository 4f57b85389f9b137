"""Form / Function / DirichletBC objects standing in for the UFL + DOLFIN
objects the reference passes around (no FEniCS exists here).  A ``Form`` is
"assemblable on the mixed space W": like the reference's forms it yields a
matrix in the MIXED numbering, from which ``PCDInterface`` extracts field
blocks through index sets (``fenapack/field_split_backend.py:240-241,
285-291``)."""

import numpy as np
import scipy.sparse as sp


class Function(object):
    """A mixed P2^2 x P1 function: host vector in the mixed numbering."""

    def __init__(self, V):
        self.V = V
        self._x = np.zeros(V.ndof)
        self.version = 0

    def function_space(self):
        return self.V

    def vector(self):
        return self._x

    def assign(self, other):
        self._x[:] = other._x if isinstance(other, Function) else other
        self.version += 1

    def touch(self):
        self.version += 1

    def split(self):
        return self._x[self.V.is_u], self._x[self.V.is_p]


class DirichletBC(object):
    """Boundary values as a {mixed dof: value} map - what
    ``DirichletBC::get_boundary_values`` returns (SubfieldBC.h:103-105)."""

    def __init__(self, V, dofs, values):
        self.V = V
        self.dofs = np.asarray(dofs, dtype=np.int64)
        self.values = np.asarray(values, dtype=np.float64)

    def get_boundary_values(self):
        return dict(zip(self.dofs.tolist(), self.values.tolist()))


class Form(object):
    """One named form of a :class:`FlowProblem` (``kind`` in
    a|a_pc|L|mp|kp|ap|mu|fp|gp)."""

    def __init__(self, kind, problem, w):
        self.kind, self.problem, self.w = kind, problem, w

    def function_space(self):
        return self.problem.space


def _embed_pp(V, M):
    """p-p block -> matrix on W (zero elsewhere)."""
    M = sp.coo_matrix(M)
    return sp.csr_matrix((M.data, (V.is_p[M.row], V.is_p[M.col])),
                         shape=(V.ndof, V.ndof))


def _embed_uu(V, M):
    M = sp.coo_matrix(M)
    return sp.csr_matrix((M.data, (V.is_u[M.row], V.is_u[M.col])),
                         shape=(V.ndof, V.ndof))


class FormAssembler(object):
    """Caches one linearisation per state of ``w`` so that the system matrix,
    the right-hand side and the preconditioner matrix of one Newton step come
    from a single pass (DOLFIN's SystemAssembler assembles them together)."""

    def __init__(self, problem, w):
        self.problem, self.w = problem, w
        self._version = None
        self._lin = None

    def linearisation(self):
        if self._version != self.w.version:
            xu, xp = self.w.split()
            self._lin = self.problem.linearise(xu, xp)
            self._version = self.w.version
        return self._lin

    def assemble(self, form):
        pb, V = self.problem, self.problem.space
        k = form.kind
        if k in ("a", "a_pc"):
            L = self.linearisation()
            A00 = L["P00"] if (k == "a_pc" and "P00" in L) else L["A00"]
            return V.monolithic(A00, L["A01"], L["A10"])
        if k == "L":
            L = self.linearisation()
            return V.to_mixed(L["bu"], L["bp"])
        if k == "mp":
            return _embed_pp(V, pb.Mp)
        if k == "ap":
            return _embed_pp(V, pb.Ap)      # symmetric PCD BCs already in
        if k == "kp":
            return _embed_pp(V, pb.Kp(self.w.split()[0]))
        if k == "mu":
            return _embed_uu(V, pb.Mu())
        if k == "gp":
            L = self.linearisation()
            return V.monolithic(0.0 * L["A00"], L["A01"], 0.0 * L["A10"])
        raise AttributeError("Form '%s' not available" % k)


def navier_stokes_forms(problem, w=None):
    """The demo's forms and BCs (demo_navier-stokes-pcd.py:76-142), bound to
    the iterate ``w``.  Returns a dict ready for ``PCDAssembler(**forms)``."""
    V = problem.space
    w = Function(V) if w is None else w
    fa = FormAssembler(problem, w)
    mk = lambda kind: _BoundForm(kind, problem, w, fa)
    # coarse-level operators at the current iterate (pc_mg_galerkin none)
    V.coarse_velocity_operators = lambda nlev: \
        problem.coarse_velocity_operators(w.split()[0], nlev)
    bc_u = DirichletBC(V, V.is_u[problem.bc_u_idx],
                       problem.bc_u_values(problem.t))
    bc_p = DirichletBC(V, V.is_p[problem.bc_p_idx], problem.bc_p_val)
    forms = dict(a=mk("a"), L=mk("L"), bcs=[bc_u],
                 a_pc=mk("a_pc") if problem.stabilize else None,
                 mp=mk("mp"), kp=mk("kp"), ap=mk("ap"), bcs_pcd=[bc_p])
    if problem.pcdr:
        forms["mu"] = mk("mu")
    return w, forms


class _BoundForm(Form):
    def __init__(self, kind, problem, w, fa):
        Form.__init__(self, kind, problem, w)
        self._fa = fa

    def assemble(self):
        return self._fa.assemble(self)
