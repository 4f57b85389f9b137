"""The four PCD Schur-complement preconditioner contexts, with the PCPYTHON
protocol and the attribute names of ``fenapack/preconditioners.py`` -
``create / setFromOptions / init_pcd / setUp / apply`` - but with the whole
``apply`` body executed as ONE device-resident call behind the C ABI
(``pcd_apply``): no Python<->device crossing per Vec/Mat/KSP operation.

Deviation from the reference, stated once: the reference's default inner
solver is an exact Cholesky factorisation (``preconditioners.py:42-49``).
Sparse factorisations have no place on this path; the default here is
Jacobi-preconditioned CG converged to ``rtol = 1e-10`` (overridable through
the same option names, e.g. ``-..._PCD_Mp_ksp_type chebyshev``).
"""

from . import _cabi as c
from .petsc import KSP, PC
from .timing import timed


def _buffer(v, writable=False):
    """(buffer, mem flag) behind a vector argument of ``apply``:

    * this package's device ``Vec`` (``.t`` = torch tensor) or a bare torch
      CUDA tensor -> device pointer, the call is enqueued on the engine's
      stream;
    * a petsc4py ``Vec`` (what PETSc hands a PCPYTHON context: it has
      ``getArray``) -> its host array, the engine copies in and out and
      synchronises (a PETSc built ``--with-hip`` can pass the device pointer
      instead: INTEGRATION.md);
    * a numpy array -> host pointer."""
    t = getattr(v, "t", None)
    if t is not None:
        v = t
    if hasattr(v, "data_ptr"):
        return v, (c.MEM_DEVICE if v.is_cuda else c.MEM_HOST)
    if hasattr(v, "getArray"):                      # petsc4py.PETSc.Vec
        return (v.getArray() if writable
                else v.getArray(readonly=True)), c.MEM_HOST
    return v, c.MEM_HOST


class BasePCDPC(object):
    """Base python context for PCD preconditioners
    (``fenapack/preconditioners.py:25-85``)."""

    variant = None

    def create(self, pc):
        self.ksp_Ap = self.create_default_ksp(pc.comm)
        self.ksp_Mp = self.create_default_ksp(pc.comm)
        options_prefix = pc.getOptionsPrefix() or ""
        self.ksp_Ap.setOptionsPrefix(options_prefix + "PCD_Ap_")
        self.ksp_Mp.setOptionsPrefix(options_prefix + "PCD_Mp_")

    def setFromOptions(self, pc):
        self.ksp_Ap.setFromOptions()
        self.ksp_Mp.setFromOptions()

    @staticmethod
    def create_default_ksp(comm):
        """Jacobi-PCG converged tightly - the device stand-in for the
        reference's PREONLY + CHOLESKY default."""
        ksp = KSP().create(comm)
        ksp.setType(KSP.Type.CG)
        ksp.pc.setType(PC.Type.JACOBI)
        ksp.setTolerances(rtol=1e-10, max_it=10000)
        return ksp

    def get_work_vecs(self, v, num):
        """``num`` work vecs duplicated from ``v``; cached forever, count
        frozen (``preconditioners.py:52-61``).  The engine owns the device
        work vectors of the fused path; these serve ``apply_by_parts``."""
        cached = self.__dict__.get("_work_vecs")
        if cached is None:
            cached = self._work_vecs = tuple(v.duplicate()
                                             for _ in range(num))
        elif len(cached) != num:
            raise ValueError("Changing number of work vecs not allowed")
        return cached

    def init_pcd(self, pcd_interface):
        if hasattr(self, "interface"):
            raise RuntimeError("Reinitialization of PCDPC not allowed")
        self.interface = pcd_interface

    def setUp(self, pc):
        # Mp/Ap: assembled, extracted and handed to the engine once (const)
        self.interface.setup_ksp_Mp(self.ksp_Mp)
        self.interface.setup_ksp_Ap(self.ksp_Ap)
        # Kp: re-assembled into the existing device matrix when not constant
        Kp = self.interface.setup_mat_Kp(mat=getattr(self, "mat_Kp", None))
        if Kp is not None:
            self.mat_Kp = Kp
            self.mat_Kp.setOptionsPrefix((pc.getOptionsPrefix() or "")
                                         + "PCD_Kp_")
        self.bcs_applier = self.interface.apply_pcd_bcs
        self.interface._subbcs_upload()
        self._engine_setup()

    def _engine_setup(self):
        self.interface.engine.setup()

    def apply(self, pc, x, y):
        """``y = -S^-1 x``: the fused HIP path (x borrowed, y overwritten).
        ``x`` / ``y``: this package's device ``Vec``, a torch tensor, a numpy
        array or a petsc4py ``Vec`` (see ``_buffer``)."""
        xb, mx = _buffer(x)
        yb, my = _buffer(y, writable=True)
        if mx != my:
            raise TypeError("apply: x and y must both live on the device or "
                            "both on the host")
        self.interface.engine.apply(xb, yb, mx)

    def apply_by_parts(self, pc, x, y):
        """Testing aid: the same operator assembled from per-operation ABI
        calls (``pcd_inner_solve``, ``pcd_spmv``, ``pcd_apply_bc``) in the
        order of the reference bodies (``preconditioners.py:124-135, 158-169,
        239-252, 285-298``); must agree with the fused :meth:`apply`."""
        laplace_first = self.variant in ("BRM1", "RBRM1")
        work = self.get_work_vecs(x, 1 if laplace_first else 2)
        if laplace_first:                 # -Mp^-1 (x + Kp Ap^-1 bc(x))
            rhs = x.copy(result=work[0])
            self.bcs_applier(rhs)
            self.ksp_Ap.solve(rhs, y)
            self.mat_Kp.mult(y, rhs)
            rhs.axpy(1.0, x)
            self.ksp_Mp.solve(rhs, y)
            spare = rhs
        else:                             # -(Mp^-1 x + Ap^-1 bc(Kp Mp^-1 x))
            self.ksp_Mp.solve(x, y)
            self.mat_Kp.mult(y.copy(result=work[0]), work[1])
            self.bcs_applier(work[1])
            self.ksp_Ap.solve(work[1], work[0])
            y.axpy(1.0, work[0])
            spare = work[0]
        if self.variant.startswith("R"):  # reaction term - Rp^-1 x
            self.ksp_Rp.solve(x, spare)
            y.axpy(1.0, spare)
        y.scale(-1.0)


class PCDPC_BRM1(BasePCDPC):
    r"""``y = -M_p^{-1} (I + K_p A_p^{-1}) x`` with the subfield BC applied to
    the right-hand side of the Laplace solve only
    (``fenapack/preconditioners.py:89-135``)."""
    variant = "BRM1"

    @timed("FENaPack: PCDPC_BRM1 apply")
    def apply(self, pc, x, y):
        return BasePCDPC.apply(self, pc, x, y)


class PCDPC_BRM2(BasePCDPC):
    r"""``y = -(I + A_p^{-1} K_p) M_p^{-1} x``
    (``fenapack/preconditioners.py:139-169``)."""
    variant = "BRM2"

    @timed("FENaPack: PCDPC_BRM2 apply")
    def apply(self, pc, x, y):
        return BasePCDPC.apply(self, pc, x, y)


class BasePCDRPC(BasePCDPC):
    """Base context of the PCDR (reaction) variants
    (``fenapack/preconditioners.py:173-207``)."""

    def create(self, pc):
        super(BasePCDRPC, self).create(pc)
        self.ksp_Rp = self.create_default_ksp(pc.comm)
        options_prefix = pc.getOptionsPrefix() or ""
        self.ksp_Rp.setOptionsPrefix(options_prefix + "PCD_Rp_")

    def setFromOptions(self, pc):
        super(BasePCDRPC, self).setFromOptions(pc)
        self.ksp_Rp.setFromOptions()

    def _engine_setup(self):
        pass                    # deferred until Rp is in place

    def setUp(self, pc):
        super(BasePCDRPC, self).setUp(pc)
        Mu = self.interface.setup_mat_Mu(mat=getattr(self, "mat_Mu", None))
        if Mu is not None:
            self.mat_Mu = Mu
            self.mat_Mu.setOptionsPrefix((pc.getOptionsPrefix() or "")
                                         + "PCD_Mu_")
        Bt = self.interface.setup_mat_Bt(mat=getattr(self, "mat_Bt", None))
        if Bt is not None:
            self.mat_Bt = Bt
            self.mat_Bt.setOptionsPrefix((pc.getOptionsPrefix() or "")
                                         + "PCD_Bt_")
        self.interface.setup_ksp_Rp(self.ksp_Rp, self.mat_Mu, self.mat_Bt)
        self.interface.engine.setup()


class PCDRPC_BRM1(BasePCDRPC):
    r"""``y = -R_p^{-1} x - M_p^{-1} (I + K_p A_p^{-1}) x``
    (``fenapack/preconditioners.py:211-252``)."""
    variant = "RBRM1"

    @timed("FENaPack: PCDRPC_BRM1 apply")
    def apply(self, pc, x, y):
        return BasePCDRPC.apply(self, pc, x, y)


class PCDRPC_BRM2(BasePCDRPC):
    r"""``y = -R_p^{-1} x - (I + A_p^{-1} K_p) M_p^{-1} x``
    (``fenapack/preconditioners.py:256-298``)."""
    variant = "RBRM2"

    @timed("FENaPack: PCDRPC_BRM2 apply")
    def apply(self, pc, x, y):
        return BasePCDRPC.apply(self, pc, x, y)


PCD_CLASSES = {"PCDPC_BRM1": PCDPC_BRM1, "PCDPC_BRM2": PCDPC_BRM2,
               "PCDRPC_BRM1": PCDRPC_BRM1, "PCDRPC_BRM2": PCDRPC_BRM2}
