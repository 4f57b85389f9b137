"""Newton/Picard driver glue: ``PCDNewtonSolver`` and ``PCDNonlinearProblem``
with the call shapes of ``fenapack/nonlinear_solvers.py:28-112``.  The Newton
loop itself is DOLFIN's in the reference ([ext] ``dolfin::NewtonSolver``);
here it is a small host loop: residual criterion, ``J dx = F``, ``x -= dx``.
The one-shot ``solver_setup`` semantic (operators + ``init_pcd`` at the first
iteration of the first solve only, ``:63-78``) is kept."""

import numpy as np

from .assembling import PCDAssembler
from .petsc import Mat


class PCDNonlinearProblem(object):
    """Interface between ``PCDNewtonSolver`` and a ``PCDAssembler``
    (``fenapack/nonlinear_solvers.py:85-112``)."""

    def __init__(self, pcd_assembler):
        assert isinstance(pcd_assembler, PCDAssembler)
        self.pcd_assembler = pcd_assembler

    def F(self, b, x):
        self.pcd_assembler.rhs_vector(b, x)

    def J(self, A, x):
        self.pcd_assembler.system_matrix(A)

    def J_pc(self, P, x):
        self.pcd_assembler.pc_matrix(P)


class PCDNewtonSolver(object):
    def __init__(self, solver, pcd_pc_class=None):
        self._solver = solver
        self._pcd_pc_class = pcd_pc_class
        self.parameters = {"relative_tolerance": 1e-9,
                           "absolute_tolerance": 1e-10,
                           "maximum_iterations": 50,
                           "relaxation_parameter": 1.0,
                           "error_on_nonconvergence": True}
        self._A, self._P = Mat(), Mat()
        self._krylov_iterations = 0
        self.krylov_history = []
        self.residual_history = []

    def linear_solver(self):
        return self._solver

    def krylov_iterations(self):
        """Outer GMRES iterations accumulated over the last ``solve``
        (``fenapack/__init__.py:44-56``)."""
        return self._krylov_iterations

    def solver_setup(self, A, P, nonlinear_problem, iteration):
        if iteration > 0 or getattr(self, "_initialized", False):
            return
        self._initialized = True
        P = A if not P.isAssembled() else P
        self._solver.set_operators(A, P)
        self._solver.init_pcd(nonlinear_problem.pcd_assembler,
                              self._pcd_pc_class)

    def solve(self, problem, x, on_update=None, final_residual=True):
        """``x``: host vector of the iterate, updated in place.
        ``on_update()`` is invoked after every update of ``x`` (it plays the
        role of DOLFIN forms seeing the new coefficient values).
        ``final_residual=False``: when the iteration limit is reached the
        residual at the last iterate is not assembled (a caller that
        continues the iteration by other means evaluates it itself)."""
        prm = self.parameters
        self._krylov_iterations = 0
        self.krylov_history, self.residual_history = [], []
        n = x.size
        b, dx = np.zeros(n), np.zeros(n)
        # a partitioned producer fills the owned entries of b only: its norm
        # reduces over the ranks (PETSc's VecNorm on a distributed Vec)
        norm = getattr(problem, "norm", None) or np.linalg.norm
        problem.F(b, x)
        r0 = r = float(norm(b))
        self.residual_history.append(r)
        # a zero first residual is a solved problem (never 0/0 below)
        it, converged = 0, (r < prm["absolute_tolerance"] or r == 0.0)
        while not converged and it < prm["maximum_iterations"]:
            problem.J(self._A, x)
            problem.J_pc(self._P, x)
            self.solver_setup(self._A, self._P, problem, it)
            dx[:] = 0.0
            its = self._solver.solve(dx, b)
            self._krylov_iterations += its
            self.krylov_history.append(its)
            x -= prm["relaxation_parameter"] * dx
            if on_update is not None:
                on_update()
            it += 1
            if not final_residual and it >= prm["maximum_iterations"]:
                break
            problem.F(b, x)
            r = float(norm(b))
            self.residual_history.append(r)
            converged = (r < prm["absolute_tolerance"] or r == 0.0
                         or r / r0 < prm["relative_tolerance"])
        if not converged and prm["error_on_nonconvergence"]:
            raise RuntimeError("Newton solver did not converge: |r|/|r0| = "
                               "%g after %d iterations"
                               % (r / r0 if r0 else float("inf"), it))
        return it, converged
