"""End-to-end drivers written against the public API exactly as the reference
demos are (``demo/navier-stokes-pcd/demo_navier-stokes-pcd.py:140-180``,
``demo/unsteady-navier-stokes-pcd/demo_unsteady-navier-stokes-pcd.py:
149-208``): forms -> PCDAssembler -> PCDNonlinearProblem; PCDKrylovSolver with
PETSc-style options; PCDNewtonSolver.solve."""

import time

from . import (PCDAssembler, PCDKrylovSolver, PCDNewtonSolver,
               PCDNonlinearProblem, PETScOptions)
from .fem.forms import navier_stokes_forms


def mass_matrix_bounds(dim):
    """Eigenvalue bounds of ``diag(M)^-1 M`` for P1: [1/2, 2] on triangles
    (the demos' setting, valid in 2D only:
    demo/navier-stokes-pcd/documentation.rst:143-147), [1/2, 5/2] on
    tetrahedra (Wathen 1987)."""
    return "0.5, 2.0" if dim == 2 else "0.5, 2.5"


def multigrid_inner_options(prefix="", cycles_u=1, cycles_p=1, smooth=2,
                            mp_its=5, pcdr=False, galerkin_u=True, dim=2,
                            two_grid_p=True, algebraic=False):
    """The reference's "iterative" configuration (demo_navier-stokes-pcd.py:
    152-165: Richardson + one/two multigrid cycles for A00 and Ap, Chebyshev +
    Jacobi for Mp) with hypre BoomerAMG replaced by the engine's geometric
    multigrid (``pc_type mg``)."""
    S = PETScOptions.set
    for key, cycles in (("fieldsplit_u_", cycles_u),
                        ("fieldsplit_p_PCD_Ap_", cycles_p)) \
            + ((("fieldsplit_p_PCD_Rp_", cycles_p),) if pcdr else ()):
        S(prefix + key + "ksp_type", "richardson")
        S(prefix + key + "ksp_max_it", cycles)
        # algebraic=True: the hierarchy is built from the matrix alone
        # (-pc_type gamg, fenapack_amd/amg.py) - what a caller with matrices
        # from another mesh / assembler uses, as the reference uses hypre
        S(prefix + key + "pc_type", "gamg" if algebraic else "mg")
        S(prefix + key + "mg_levels_ksp_max_it", smooth)
    if two_grid_p and not algebraic:
        # Pressure Laplacian: finest level + explicit coarse inverse, nothing
        # in between.  On an MI355X a kernel on <= 10^5 rows costs a fixed
        # 4-6 us whatever it does, and the intermediate levels of this SPD
        # solve buy no outer iterations for their ~18 launches
        S(prefix + "fieldsplit_p_PCD_Ap_pc_mg_skip_levels", "all")
    if not galerkin_u:
        # re-discretised (SUPG-stabilised) coarse velocity operators instead
        # of Galerkin products: needed at cell Peclet numbers > 1
        S(prefix + "fieldsplit_u_pc_mg_galerkin", "none")
    S(prefix + "fieldsplit_p_PCD_Mp_ksp_type", "chebyshev")
    S(prefix + "fieldsplit_p_PCD_Mp_ksp_max_it", mp_its)
    S(prefix + "fieldsplit_p_PCD_Mp_ksp_chebyshev_eigenvalues",
      mass_matrix_bounds(dim))
    S(prefix + "fieldsplit_p_PCD_Mp_pc_type", "jacobi")


def default_inner_options(prefix="", a00_its=60, a00_ratio=0.01, ap_rtol=1e-8,
                          ap_its=10000, mp_its=5, pcdr=False, dim=2):
    """North-star inner solvers expressed with the reference's option names
    (demo_navier-stokes-pcd.py:151-165 sets the same keys)."""
    S = PETScOptions.set
    S(prefix + "fieldsplit_u_ksp_type", "chebyshev")
    S(prefix + "fieldsplit_u_ksp_max_it", a00_its)
    S(prefix + "fieldsplit_u_pc_type", "jacobi")
    S(prefix + "fieldsplit_u_ksp_chebyshev_esteig",
      "0,%g,0,1.1" % a00_ratio)
    S(prefix + "fieldsplit_p_PCD_Ap_ksp_type", "cg")
    S(prefix + "fieldsplit_p_PCD_Ap_ksp_max_it", ap_its)
    S(prefix + "fieldsplit_p_PCD_Ap_ksp_rtol", ap_rtol)
    S(prefix + "fieldsplit_p_PCD_Ap_pc_type", "jacobi")
    S(prefix + "fieldsplit_p_PCD_Mp_ksp_type", "chebyshev")
    S(prefix + "fieldsplit_p_PCD_Mp_ksp_max_it", mp_its)
    S(prefix + "fieldsplit_p_PCD_Mp_ksp_chebyshev_eigenvalues",
      mass_matrix_bounds(dim))
    S(prefix + "fieldsplit_p_PCD_Mp_pc_type", "jacobi")
    if pcdr:
        S(prefix + "fieldsplit_p_PCD_Rp_ksp_type", "cg")
        S(prefix + "fieldsplit_p_PCD_Rp_ksp_max_it", ap_its)
        S(prefix + "fieldsplit_p_PCD_Rp_ksp_rtol", ap_rtol)
        S(prefix + "fieldsplit_p_PCD_Rp_pc_type", "jacobi")


def make_solver(problem, prefix="", gmres_rtol=1e-6, restart=150,
                newton_rtol=1e-5, max_newton=25, device=0, comm=None):
    """Wire up the solver stack for ``problem``; returns (w, nls, nlp)."""
    w, forms = navier_stokes_forms(problem)
    assembler = PCDAssembler(**forms)
    nlp = PCDNonlinearProblem(assembler)
    if getattr(problem, "partitioned", False):
        # this rank's rows only (fem/partition.py): residual norms reduce
        nlp.norm = problem.norm
    linear_solver = PCDKrylovSolver(comm=comm, device=device)
    if prefix:
        linear_solver.set_options_prefix(prefix)
    linear_solver.parameters["relative_tolerance"] = gmres_rtol
    linear_solver.parameters["maximum_iterations"] = 600
    PETScOptions.set(prefix + "ksp_gmres_restart", restart)
    cls = ("PCDRPC_" if problem.pcdr else "PCDPC_") + problem.variant
    PETScOptions.set(prefix + "fieldsplit_p_pc_python_type",
                     "fenapack." + cls)
    linear_solver.set_from_options()
    nls = PCDNewtonSolver(linear_solver)
    nls.parameters["relative_tolerance"] = newton_rtol
    nls.parameters["maximum_iterations"] = max_newton
    return w, nls, nlp


def solve_steady(problem, **kw):
    """Run the steady demo; returns a stats dict (M2 = GMRES its/Newton step)."""
    w, nls, nlp = make_solver(problem, **kw)
    nls.parameters["error_on_nonconvergence"] = False
    t0 = time.time()
    its, converged = nls.solve(nlp, w.vector(), on_update=w.touch)
    return {"w": w, "newton_its": its, "converged": converged,
            "krylov_its": nls.krylov_iterations(),
            "krylov_per_step": list(nls.krylov_history),
            "residuals": list(nls.residual_history),
            "time": time.time() - t0, "solver": nls}


def solve_unsteady(problem, dt, t_end, **kw):
    """Backward-Euler time loop of the unsteady demo (:188-208)."""
    w, nls, nlp = make_solver(problem, **kw)
    V = problem.space
    t, steps, krylov, newton = 0.0, 0, 0, 0
    per_step, newton_per_step, residuals = [], [], []
    t0 = time.time()
    while t < t_end - 0.1 * dt:
        t += dt
        steps += 1
        problem.t = t                           # inflow.t = t
        w.touch()
        n_it, _ = nls.solve(nlp, w.vector(), on_update=w.touch)
        krylov += nls.krylov_iterations()
        newton += n_it
        per_step.append(nls.krylov_iterations())
        newton_per_step.append(list(nls.krylov_history))
        residuals.append(list(nls.residual_history))
        problem.u0 = w.split()[0].copy()        # w0.assign(w)
        w.touch()
    return {"w": w, "steps": steps, "krylov_its": krylov,
            "krylov_per_step": per_step, "newton_its": newton,
            "krylov_per_newton": newton_per_step, "residuals": residuals,
            "time": time.time() - t0, "ndof": V.ndof}
