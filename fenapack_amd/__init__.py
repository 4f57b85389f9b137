"""fenapack_amd - MI355X-native PCD preconditioner-apply engine behind the
fenapack operator API.

Re-exports the public names of ``fenapack/__init__.py:35-40``.  The package
holds the host-side mirror of the reference interface for the PCD /
fieldsplit apply path and the HIP engine (``csrc/``) behind a C ABI
(``include/pcd_engine.h``); the product fails loudly when the HIP library is
missing and never falls back to CPU arithmetic.
"""

__version__ = "0.1.0"

from fenapack_amd.field_split import PCDKSP, PCDKrylovSolver
from fenapack_amd.assembling import PCDAssembler, PCDForm
from fenapack_amd.nonlinear_solvers import (PCDNewtonSolver,
                                            PCDNonlinearProblem)
from fenapack_amd.preconditioners import (PCDPC_BRM1, PCDPC_BRM2,
                                          PCDRPC_BRM1, PCDRPC_BRM2)
from fenapack_amd.stabilization import StabilizationParameterSD
from fenapack_amd.petsc import PETScOptions
from fenapack_amd.timing import Timer, timed, timings, list_timings
from fenapack_amd import _guard

# bench.py, tools/*, demo/* and the test suite run under the resident-set
# watchdog (a host application that imports this package does not, unless it
# sets FENAPACK_AMD_WATCHDOG=1): two GPU boxes were lost to host allocations
# of this repository's own scripts
_guard.autostart()


def limit_blas_threads(n=None):
    """Cap the threads of the BLAS libraries numpy / scipy have loaded.

    The host producer's dense work is thousands of SMALL calls (norms and dots
    of the power iterations, 18 x 36 reference-tensor GEMMs, coarse inverses
    of a few hundred rows).  OpenBLAS's default on a many-core host - one
    thread per core, up to its build limit - wakes the whole pool for each of
    them: measured on a 2 x 64-core EPYC 9575F, the level-6 bench set-up takes
    4.8 s at the default and 2.9 s with 8 (or 1) BLAS threads
    (``profiles/r03_setup_by_blas_threads.txt``).  Called once at import with
    ``FENAPACK_AMD_BLAS_THREADS`` (0 / unset in a host application = leave its
    pools alone; the scripts of this repository default to 8); an
    ``OPENBLAS_NUM_THREADS`` the user set wins - and is the better route
    (``bench.py`` and ``tests/conftest.py`` set it before numpy is imported):
    resizing a pool at run time makes OpenBLAS start new threads, and threads
    started after an OpenMP runtime has BOUND the main thread
    (``OMP_PROC_BIND``) inherit its one-core mask - eight BLAS threads on one
    core took this set-up from 3 s to 50 s, so nothing is resized then."""
    import os
    if n is None:
        if os.environ.get("OPENBLAS_NUM_THREADS"):
            return None
        if any(os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES",
                                           "GOMP_CPU_AFFINITY",
                                           "KMP_AFFINITY")):
            return None
        # opt-in for a host application (its BLAS pools are its own); the
        # scripts of this repository (bench.py, tools/, demo/) get 8
        n = int(os.environ.get("FENAPACK_AMD_BLAS_THREADS",
                               "8" if _guard.main_script_is_ours() else "0"))
    if n <= 0:
        return None
    try:
        import scipy.linalg                   # (loads scipy's own OpenBLAS)
        import scipy.sparse.linalg            # noqa: F401
        from threadpoolctl import threadpool_limits
    except ImportError:                       # a speed matter only
        return None
    return threadpool_limits(limits=min(n, os.cpu_count() or n),
                             user_api="blas")


_blas_limit = limit_blas_threads()

__all__ = ["PCDKSP", "PCDKrylovSolver", "PCDAssembler", "PCDForm",
           "PCDNewtonSolver", "PCDNonlinearProblem", "PCDPC_BRM1",
           "PCDPC_BRM2", "PCDRPC_BRM1", "PCDRPC_BRM2", "StabilizationParameterSD",
           "PETScOptions", "Timer", "timed", "timings", "list_timings"]
