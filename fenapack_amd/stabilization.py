"""``StabilizationParameterSD``: streamline-diffusion parameter of the
reference (``fenapack/stabilization.py:87-118``; C++ formula ``:66-67``):

    Pe = 0.5 * |w| * h * rho / nu;   delta = Pe > 1 ? 0.5*h*(1 - 1/Pe)/|w| : 0

as a piecewise-constant (DG0) field, with ``h`` = DOLFIN's ``Cell::h()``
(twice the circumradius) and ``w`` evaluated at the cell midpoint.  The
reference JIT-compiles a ``dolfin::Expression``; here it is a host-side
callable of the fixed-form assembler - it only changes the values of the
00-block of the preconditioner matrix (``demo_navier-stokes-pcd.py:122-125``),
never a kernel."""

import numpy as np


class _DeltaSD(object):
    def __init__(self, space, wind, viscosity, density):
        self.V, self.wind = space, wind
        self.nu, self.rho = float(viscosity), float(density)

    def cell_values(self):
        """delta per cell for the current wind (nodal P2 velocity (nn, dim),
        or a ``Function`` whose velocity part is used)."""
        w = self.wind
        d = self.V.dim
        U = w.split()[0].reshape(-1, d) if hasattr(w, "split") \
            else np.asarray(w).reshape(-1, d)
        return self.V.supg_delta(U, self.nu, self.rho)

    __call__ = cell_values


def StabilizationParameterSD(wind, viscosity, density=None):
    """Returns the SD stabilisation parameter bound to ``wind`` (a mixed
    ``Function`` or nodal velocity array living on a ``TaylorHood`` space
    available as ``wind.function_space()`` / ``wind.V``)."""
    if density is None:
        density = 1.0
    V = wind.function_space() if hasattr(wind, "function_space") else wind.V
    return _DeltaSD(V, wind, viscosity, density)
