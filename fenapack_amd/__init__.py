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

__all__ = ["PCDKSP", "PCDKrylovSolver", "PCDAssembler", "PCDForm",
           "PCDNewtonSolver", "PCDNonlinearProblem", "PCDPC_BRM1",
           "PCDPC_BRM2", "PCDRPC_BRM1", "PCDRPC_BRM2", "StabilizationParameterSD",
           "PETScOptions", "Timer", "timed", "timings", "list_timings"]
