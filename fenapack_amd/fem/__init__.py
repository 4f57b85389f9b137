"""Fixed-form P2/P1 finite-element input producer for the PCD engine."""
from .mesh import (Mesh, TetMesh, lshape_mesh, unit_square_mesh,
                   cavity_mesh, unit_cube_mesh)
from .taylor_hood import TaylorHood, FixedPattern
from .problems import (FlowProblem, BackwardStep, Cavity, Cavity3D,
                       Channel3D)
from .forms import Function, DirichletBC, Form, navier_stokes_forms
