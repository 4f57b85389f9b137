"""``PCDAssembler`` / ``PCDForm``: container of the forms and boundary
conditions that define the linear problem and the PCD operators - same
constructor signature, method names, ``const``/``phantom`` defaults and error
behaviour as ``fenapack/assembling.py:27-230``.  Assembly itself is delegated
to the fixed-form P2/P1 producer in :mod:`fenapack_amd.fem` (UFL/FFC/DOLFIN
are out of scope: SURVEY 2, row 8)."""

from .petsc import Mat


class PCDForm(object):
    """Form wrapper recording whether the operator stays constant across
    outer iterations and whether it is a *phantom* (taken from the system
    matrix instead of being assembled): ``fenapack/assembling.py:192-230``."""

    def __init__(self, form, const=False, phantom=False):
        assert isinstance(const, bool)
        self._form = form
        self.constant = const
        self.phantom = phantom

    def dolfin_form(self):
        return self._form

    def is_constant(self):
        return self.constant

    def is_phantom(self):
        return self.phantom


class PCDAssembler(object):
    """Mirror of ``fenapack.assembling.PCDAssembler`` (``:35-189``).

    Defaults as in the reference (``:98-106``): ``ap``, ``mp``, ``mu``, ``gp``
    constant; ``fp``, ``kp`` re-assembled every outer iteration; ``gp`` is a
    phantom taken from the 01-block of the system matrix."""

    def __init__(self, a, L, bcs, a_pc=None, mp=None, mu=None, ap=None,
                 fp=None, kp=None, gp=None, bcs_pcd=[]):
        self._a, self._a_pc = a, a_pc
        self._bcs = bcs
        self._bcs_pcd = bcs_pcd
        self._forms = {
            "L": PCDForm(L),
            "ap": PCDForm(ap, const=True),
            "mp": PCDForm(mp, const=True),
            "mu": PCDForm(mu, const=True),
            "fp": PCDForm(fp),
            "kp": PCDForm(kp),
            "gp": PCDForm(gp, const=True, phantom=True),
        }

    def get_pcd_form(self, key):
        form = self._forms.get(key)
        if form is None or (form.dolfin_form() is None
                            and not form.is_phantom()):
            raise AttributeError("Form '%s' requested by PCD not available"
                                 % key)
        return form

    def get_dolfin_form(self, key):
        return self.get_pcd_form(key).dolfin_form()

    def function_space(self):
        return self.get_dolfin_form("L").function_space()

    # -- linear system ------------------------------------------------------
    def rhs_vector(self, b, x=None):
        """Residual with BCs applied (``:127-136``); ``b`` is a host array."""
        b[:] = self.get_dolfin_form("L").assemble()

    def system_matrix(self, A):
        A.set(self._a.assemble())

    def pc_matrix(self, P):
        if self._a_pc is not None:
            P.set(self._a_pc.assemble())

    # -- PCD operators (each on the mixed space, like the reference) --------
    def ap(self, Ap):
        Ap.set(self.get_dolfin_form("ap").assemble())

    def mp(self, Mp):
        Mp.set(self.get_dolfin_form("mp").assemble())

    def mu(self, Mu):
        Mu.set(self.get_dolfin_form("mu").assemble())

    def fp(self, Fp):
        Fp.set(self.get_dolfin_form("fp").assemble())

    def kp(self, Kp):
        Kp.set(self.get_dolfin_form("kp").assemble())

    def gp(self, Bt):
        Bt.set(self.get_dolfin_form("gp").assemble())

    def pcd_bcs(self):
        if getattr(self, "_bcs_pcd", None) is None:
            raise AttributeError("BCs requested by PCD not available")
        return self._bcs_pcd
