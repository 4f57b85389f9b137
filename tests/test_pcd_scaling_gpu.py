"""GPU suite, part 4: counterpart of the reference's bench/integration test
``test/bench/test_pcd_scaling.py:194-256`` - combinations of nonlinear solver
x PCD variant on refined L-shape meshes; like the reference the only assertion
on the solve path is convergence (plus a sanity bound on the iteration
count).  ``ls`` = the multigrid counterpart of the reference's "iterative"."""
import pytest

from fenapack_amd import PETScOptions
from fenapack_amd.driver import multigrid_inner_options, solve_steady
from fenapack_amd.fem import BackwardStep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nls", ["picard", "newton"])
@pytest.mark.parametrize("pcd_variant", ["BRM1", "BRM2"])
@pytest.mark.parametrize("level", [2, 3])
def test_scaling_mesh(nls, pcd_variant, level):
    PETScOptions.clear()
    multigrid_inner_options(cycles_u=2, cycles_p=2)
    pb = BackwardStep(level, nu=0.02, variant=pcd_variant, nls=nls)
    out = solve_steady(pb, newton_rtol=1e-5, gmres_rtol=1e-6)
    PETScOptions.clear()
    assert out["converged"], (nls, pcd_variant, level, out["residuals"])
    assert max(out["krylov_per_step"]) < 200
