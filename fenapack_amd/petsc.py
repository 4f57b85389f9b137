"""Duck-typed stand-ins for the petsc4py objects the reference touches.

No PETSc exists on either box, so the host layer exposes only the ~25
petsc4py methods fenapack calls (complete list: SURVEY 8b), backed by device
buffers (torch tensors as plumbing) and by the engine behind the C ABI.  If a
real petsc4py is ever present, the ``PCDPC_*`` classes keep the PCPYTHON
protocol (``create/setFromOptions/setUp/apply``) and can be handed to
``pc.setPythonContext`` unchanged.
"""

import os

import numpy as np
import scipy.sparse as sp

from . import _cabi as c
from . import _host


# --------------------------------------------------------------- options DB
class _OptionsDB(dict):
    """The global PETSc options database (``PETScOptions.set``)."""


_options = _OptionsDB()


class PETScOptions(object):
    """``dolfin.PETScOptions`` look-alike (demo_navier-stokes-pcd.py:147-165)."""

    @staticmethod
    def set(name, value=""):
        _options[name.lstrip("-")] = str(value)

    @staticmethod
    def clear(name=None):
        if name is None:
            _options.clear()
        else:
            _options.pop(name.lstrip("-"), None)


class Options(object):
    """``PETSc.Options(prefix)`` (field_split.py:110)."""

    def __init__(self, prefix=""):
        self.prefix = prefix or ""

    def _get(self, name):
        return _options.get(self.prefix + name)

    def hasName(self, name):
        return self._get(name) is not None

    def getString(self, name, default=None):
        v = self._get(name)
        return default if v is None else v

    def getInt(self, name, default=None):
        v = self._get(name)
        return default if v is None else int(v)

    def getReal(self, name, default=None):
        v = self._get(name)
        return default if v is None else float(v)

    def setValue(self, name, value):
        _options[self.prefix + name] = str(value)


# -------------------------------------------------------------------- Vec
def _torch():
    import torch
    return torch


class Vec(object):
    """fp64 vector resident in HBM (a 1-D torch tensor on the engine's GPU)."""

    def __init__(self, array=None, size=None, device="cuda:0"):
        torch = _torch()
        if array is not None:
            if isinstance(array, torch.Tensor):
                self.t = array
            else:
                self.t = torch.as_tensor(np.asarray(array, dtype=np.float64),
                                         device=device)
        else:
            self.t = torch.zeros(int(size), dtype=torch.float64,
                                 device=device)

    # the methods preconditioners.py:58,128-135 uses
    def duplicate(self):
        return Vec(_torch().zeros_like(self.t))

    def copy(self, result=None):
        if result is None:
            return Vec(self.t.clone())
        result.t.copy_(self.t)
        return result

    def axpy(self, alpha, x):
        self.t.add_(x.t, alpha=alpha)

    def scale(self, alpha):
        self.t.mul_(alpha)

    def reciprocal(self):
        # VecReciprocal: zero entries stay zero (rows a rank does not hold)
        t = self.t
        nz = t != 0
        t[nz] = 1.0 / t[nz]

    def sqrtabs(self):
        self.t.abs_().sqrt_()

    def getSize(self):
        return self.t.numel()

    def getArray(self):
        return self.t.cpu().numpy()

    def setArray(self, a):
        self.t.copy_(_torch().as_tensor(np.asarray(a, dtype=np.float64)))

    def norm(self):
        return float(self.t.norm())

    def data_ptr(self):
        return self.t.data_ptr()


class IS(object):
    """Index set: sorted global dof numbers (``dofmap_dofs_is``,
    _field_split_utils.py:39-50)."""

    def __init__(self, indices, comm=None):
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.comm = comm

    def getIndices(self):
        return self.indices

    def getSize(self):
        return self.indices.size


class Mat(object):
    """Host CSR (scipy) handle; the engine copies it to HBM on hand-over."""

    class Option(object):
        SPD = "spd"

    def __init__(self, A=None, comm=None):
        self.A = None if A is None else sp.csr_matrix(A)
        self.comm = comm
        self._prefix = None
        self._spd = False
        self.state = 0          # bumped on every (re)assembly

    @property
    def type(self):
        return None if self.A is None else "seqaij"

    def isAssembled(self):
        return self.A is not None

    def set(self, A):
        self.A = A if sp.isspmatrix_csr(A) else sp.csr_matrix(A)
        self.state += 1
        return self

    def getSize(self):
        return self.A.shape

    def setOptionsPrefix(self, p):
        self._prefix = p

    def getOptionsPrefix(self):
        return self._prefix

    def setOption(self, opt, flag):
        if opt == Mat.Option.SPD:
            self._spd = bool(flag)

    def getDiagonal(self, result=None):
        d = self.A.diagonal()
        if result is None:
            return Vec(d, device="cpu")
        result.setArray(d)
        return result

    def getVecLeft(self):
        # Mats are host objects, so are the vectors derived from them
        return Vec(size=self.A.shape[0], device="cpu")

    def duplicate(self):
        return Mat(self.A * 0.0, self.comm)

    def copy(self, result=None):
        if result is None:
            return Mat(self.A.copy(), self.comm)
        result.set(self.A.copy())
        return result

    def diagonalScale(self, L=None, R=None):
        if L is not None:
            self.A = sp.diags(L.getArray()) @ self.A
        if R is not None:
            self.A = self.A @ sp.diags(R.getArray())
        self.A = self.A.tocsr()
        self.state += 1

    def transposeMatMult(self, B, result=None):
        C = (self.A.T @ B.A).tocsr()
        C.sort_indices()
        if result is None:
            return Mat(C, self.comm)
        return result.set(C)

    def createSubMatrix(self, isrow, iscol=None, submat=None):
        iscol = isrow if iscol is None else iscol
        S = self.A[isrow.indices][:, iscol.indices].tocsr()
        S.sort_indices()
        if submat is None:
            return Mat(S, self.comm)
        return submat.set(S)

    def mult(self, x, y):
        raise NotImplementedError(
            "Mat.mult on a bare Mat: operators live in an engine; use the "
            "DeviceMat returned by PCDInterface.setup_mat_*")


class DeviceMat(Mat):
    """A Mat that has been handed to an engine slot: ``mult`` is the HIP CSR
    SpMV (preconditioners.py:131,164)."""

    def __init__(self, engine, which, A, comm=None):
        Mat.__init__(self, A, comm)
        self.engine, self.which = engine, which
        self._rows = None
        if getattr(engine, "local_handover", False):
            # this rank's rows only (pcd_set_csr_local); all of these live on
            # the pressure space
            self._rows = engine.row_range(self.A.shape[0])
            r0, r1 = self._rows
            engine.set_csr_local(which, self.A[r0:r1], self.A.shape)
        else:
            engine.set_csr(which, self.A)

    def update(self, A):
        self.set(A)
        if self._rows is not None:
            r0, r1 = self._rows
            ip = self.A.indptr
            self.engine.update_values(self.which, self.A.data[ip[r0]:ip[r1]])
        else:
            self.engine.update_values(self.which, self.A.data)

    def mult(self, x, y):
        self.engine.spmv(self.which, x.t, y.t, c.MEM_DEVICE)


# ------------------------------------------------------------------ KSP / PC
_UNSUPPORTED_PC = ("lu", "cholesky", "ilu", "icc", "ml", "bjacobi", "asm",
                   "sor")
_SUPPORTED_PC = ("none", "jacobi", "mg")
# algebraic multigrid: -pc_type gamg = smoothed aggregation built from the
# matrix (fenapack_amd/amg.py) feeding the engine's cycle; -pc_type hypre (the
# reference's demo_navier-stokes-pcd.py:155-160) has no device counterpart and
# is served by the same algebraic hierarchy, with a note on stderr
_ALGEBRAIC_PC = ("gamg", "hypre")
_warned_hypre = [False]


class PC(object):
    class Type(object):
        NONE, JACOBI, LU, CHOLESKY, FIELDSPLIT, PYTHON, MG = \
            "none", "jacobi", "lu", "cholesky", "fieldsplit", "python", "mg"

    class Side(object):
        LEFT, RIGHT = "left", "right"

    class CompositeType(object):
        SCHUR = "schur"

    class SchurFactType(object):
        UPPER = "upper"

    class SchurPreType(object):
        USER = "user"

    def __init__(self, ksp):
        self.ksp = ksp
        self.type = PC.Type.JACOBI
        self.comm = ksp.comm
        self._ctx = None
        self._fs = {}
        self._is = None
        self._subksp = None
        # [ext PETSc] PCMG: prolongation chain, levels, smoother settings
        self._mg_chain = None
        self.mg_levels = None
        self.mg_coarse_eq_limit = 2000     # like -pc_gamg_coarse_eq_limit
        # -pc_mg_skip_levels: intermediate levels (1 = first above the
        # coarsest) left out of the cycle, "all" = two-grid (finest level +
        # explicit coarse inverse); their interpolations are composed
        self.mg_skip_levels = ()
        self.mg_smooth_its = 2
        # -pc_mg_smoothdown / -pc_mg_smoothup: different pre- and
        # post-smoothing counts (default: mg_levels_ksp_max_it for both)
        self.mg_smooth_down = None
        self.mg_smooth_up = None
        self.mg_esteig = (0.0, 0.1, 0.0, 1.1)
        self.mg_galerkin = True            # -pc_mg_galerkin both | none
        self._mg_ops_cb = None
        self._mg_pushed = None
        # -pc_mg_fuse_nnz: levels whose pre-composed up-sweep operator holds
        # at most this many stored entries (of the scalar stencil for an
        # F (x) I operator) run as three launches instead of nu_pre + nu_post
        # + 3 (compose.vcycle_level; 0 = never).  Such levels are
        # latency-bound, not bandwidth-bound: DESIGN.md 4.  Its rows must
        # also suit a kernel: at most mg_fuse_row_nnz entries on average (the
        # stream kernel then needs <= 3-4 passes through its LDS tile), or few
        # enough rows for the wave-per-row kernel (3-D Galerkin levels have
        # 400+ entries per row of W_u: measured 125 us composed against
        # 28 us step by step on a 35 937-node level, profiles/r02_n_*)
        self.mg_fuse_nnz = 8000000
        self.mg_fuse_row_nnz = 200
        self.mg_fuse_wave_nodes = 8192
        self.mg_fuse_rows = 160000         # never even try above this size
        # -pc_type gamg: the chain comes from the matrix (amg.py), built once
        # per pattern and kept while values change
        self.mg_algebraic = False
        self.mg_gamg_threshold = 0.02
        # a coupled (Newton) velocity block: "scalar" = aggregate the mean of
        # its diagonal blocks, P = P_s (x) I_d (what the device producer
        # refreshes); "block" = block-norm graph, general P (amg.py)
        self.mg_gamg_coupled = "scalar"

    def setMGOperators(self, callback):
        """``callback(nlev)`` -> operators of the ``nlev - 1`` coarse levels,
        coarsest first (used with ``-pc_mg_galerkin none``)."""
        self._mg_ops_cb = callback

    def setMGInterpolations(self, chain):
        """``chain[l]`` maps level l-1 to l, ``chain[0] is None``
        (PCMGSetInterpolation)."""
        self._mg_chain = list(chain)

    def setType(self, t):
        self.type = t

    def getOptionsPrefix(self):
        return self.ksp.getOptionsPrefix()

    def setFactorSolverType(self, t):
        self._factor = t

    # fieldsplit configuration (field_split.py:54-57)
    def setFieldSplitType(self, t):
        self._fs["type"] = t

    def setFieldSplitSchurFactType(self, t):
        self._fs["fact"] = t

    def setFieldSplitSchurPreType(self, t, pre=None):
        self._fs["pre"] = t

    def setFieldSplitIS(self, *fields):
        self._is = [(name, iset) for name, iset in fields]

    def getFieldSplitSubKSP(self):
        if self._subksp is None:
            raise RuntimeError("PC not set up: call setUp() after "
                               "setFieldSplitIS() (PETSc issue #160)")
        return self._subksp

    def setUp(self):
        if self.type == PC.Type.FIELDSPLIT and self._subksp is None:
            if not self._is:
                raise RuntimeError("fieldsplit PC needs setFieldSplitIS")
            prefix = self.getOptionsPrefix() or ""
            subs = []
            for name, _ in self._is:
                k = KSP(self.comm)
                k.setOptionsPrefix("%sfieldsplit_%s_" % (prefix, name))
                subs.append(k)
            self._subksp = tuple(subs)
        elif self.type == PC.Type.PYTHON and self._ctx is not None:
            self._ctx.setUp(self)

    def setPythonContext(self, ctx):
        self._ctx = ctx
        if hasattr(ctx, "create"):
            ctx.create(self)

    def getPythonContext(self):
        return self._ctx


class KSP(object):
    """Inner/outer Krylov solver *description*; the arithmetic is the
    engine's.  Supported inner types: preonly, richardson, chebyshev, cg with
    pc none|jacobi; anything else is rejected loudly at setFromOptions."""

    class Type(object):
        PREONLY, RICHARDSON, CHEBYSHEV, CG, GMRES = \
            "preonly", "richardson", "chebyshev", "cg", "gmres"

    def __init__(self, comm=None):
        self.comm = comm
        self.type = KSP.Type.CG
        self.pc = PC(self)
        self.pc_side = PC.Side.LEFT
        self._prefix = ""
        self._ops = (Mat(), Mat())
        self.max_it = 10000
        self.rtol = 1e-5
        self.atol = 1e-50
        self.norm_type = "default"
        self.cheb_eigs = None          # (emin, emax) or None -> estimate
        self.cheb_esteig = (0.0, 0.1, 0.0, 1.1)
        # -ksp_chebyshev_precompose k: Chebyshev + Jacobi with a FIXED step
        # count on a small operator is applied as <= k sparse factors composed
        # on the host (compose.chebyshev_factors) instead of max_it dependent
        # launches; 0 = step by step
        # -ksp_cg_single_reduction [ext PETSc]: both inner products of a CG
        # iteration in one reduction (one 16-byte all-reduce per iteration on
        # several ranks instead of two)
        self.cg_single_reduction = False
        self.cheb_precompose = 2
        # measured on an MI355X (profiles/r02_c_timeline.txt): at 103 041 rows
        # the two factors of Chebyshev(5) on a P1 mass matrix (5.0x and 2.65x
        # its entries) take 28 us against 25 us for the five steps - streaming
        # bytes already matter there; composition pays below ~3 10^4 rows
        self.cheb_precompose_rows = 30000
        self.restart = 30
        self.engine = None
        self.slot = None
        self.its = 0

    def create(self, comm=None):
        self.comm = comm
        return self

    def setType(self, t):
        self.type = t

    def setPCSide(self, side):
        self.pc_side = side

    def setOptionsPrefix(self, prefix):
        self._prefix = prefix or ""

    def getOptionsPrefix(self):
        return self._prefix

    def setOperators(self, A, P=None):
        self._ops = (A, A if P is None else P)

    def getOperators(self):
        return self._ops

    def setTolerances(self, rtol=None, atol=None, max_it=None):
        if rtol is not None:
            self.rtol = rtol
        if atol is not None:
            self.atol = atol
        if max_it is not None:
            self.max_it = max_it

    def setFromOptions(self):
        o = Options(self._prefix)
        t = o.getString("ksp_type")
        if t is not None:
            if t not in ("preonly", "richardson", "chebyshev", "cg", "gmres"):
                raise ValueError("%sksp_type %s is not supported by the HIP "
                                 "engine" % (self._prefix, t))
            self.type = t
        p = o.getString("pc_type")
        if p is not None:
            if p in _UNSUPPORTED_PC:
                raise ValueError(
                    "%spc_type %s has no device counterpart; use jacobi "
                    "(north star: Jacobi-CG / Chebyshev-Jacobi), mg or gamg"
                    % (self._prefix, p))
            if p in _ALGEBRAIC_PC:
                if p == "hypre" and not _warned_hypre[0]:
                    import sys
                    sys.stderr.write(
                        "fenapack_amd: -%spc_type hypre: BoomerAMG does not "
                        "exist on this path; using the engine's algebraic "
                        "multigrid (smoothed aggregation, = -pc_type gamg)\n"
                        % self._prefix)
                    _warned_hypre[0] = True
                self.pc.setType("mg")
                self.pc.mg_algebraic = True
            else:
                self.pc.setType(p)
                if p == "mg":
                    self.pc.mg_algebraic = False
        th = o.getReal("pc_gamg_threshold")
        if th is not None:
            self.pc.mg_gamg_threshold = th
        self.pc.mg_gamg_coupled = o.getString("pc_gamg_coupled",
                                              self.pc.mg_gamg_coupled)
        self.pc.mg_levels = o.getInt("pc_mg_levels", self.pc.mg_levels)
        self.pc.mg_coarse_eq_limit = o.getInt("pc_mg_coarse_eq_limit",
                                              self.pc.mg_coarse_eq_limit)
        skip = o.getString("pc_mg_skip_levels")
        if skip:
            self.pc.mg_skip_levels = "all" if str(skip).strip() == "all" \
                else tuple(int(t) for t in
                           str(skip).replace(",", " ").split())
        g = o.getString("pc_mg_galerkin")
        if g is not None:
            if g not in ("both", "none"):
                raise ValueError("%spc_mg_galerkin %s: use both | none"
                                 % (self._prefix, g))
            self.pc.mg_galerkin = g == "both"
        self.pc.mg_smooth_its = o.getInt("mg_levels_ksp_max_it",
                                         self.pc.mg_smooth_its)
        self.pc.mg_smooth_down = o.getInt("pc_mg_smoothdown",
                                          self.pc.mg_smooth_down)
        self.pc.mg_smooth_up = o.getInt("pc_mg_smoothup",
                                        self.pc.mg_smooth_up)
        self.pc.mg_fuse_nnz = o.getInt("pc_mg_fuse_nnz", self.pc.mg_fuse_nnz)
        self.cheb_precompose = o.getInt("ksp_chebyshev_precompose",
                                        self.cheb_precompose)
        e = o.getString("mg_levels_ksp_chebyshev_esteig")
        if e is not None:
            self.pc.mg_esteig = tuple(float(v)
                                      for v in e.replace(",", " ").split())
        self.max_it = o.getInt("ksp_max_it", self.max_it)
        self.rtol = o.getReal("ksp_rtol", self.rtol)
        self.atol = o.getReal("ksp_atol", self.atol)
        self.norm_type = o.getString("ksp_norm_type", self.norm_type)
        sr = o.getString("ksp_cg_single_reduction")
        if sr is not None:
            self.cg_single_reduction = str(sr).lower() in ("1", "true", "yes",
                                                           "")
        self.restart = o.getInt("ksp_gmres_restart", self.restart)
        e = o.getString("ksp_chebyshev_eigenvalues")
        if e is not None:
            lo, hi = (float(v) for v in e.replace(",", " ").split())
            self.cheb_eigs = (lo, hi)
        e = o.getString("ksp_chebyshev_esteig")
        if e is not None:
            self.cheb_esteig = tuple(float(v)
                                     for v in e.replace(",", " ").split())

    # -- binding to an engine slot ---------------------------------------
    def bind(self, engine, slot):
        self.engine, self.slot = engine, slot

    def _chebyshev_bounds(self):
        if self.cheb_eigs is not None:
            return self.cheb_eigs
        A = self._ops[1].A
        if not hasattr(self, "_cheb_warm"):
            self._cheb_warm = {}
        prod = _producer(self.engine)
        emax = estimate_emax(A, jacobi=(self.pc.type == "jacobi"),
                             warm=self._cheb_warm,
                             reduce=None if prod is None else prod.host.sum)
        a, b, cc, d = self.cheb_esteig
        return (b * emax, d * emax)        # emin estimate taken as 0

    def _push_multigrid(self):
        """Galerkin coarse operators, smoother bounds and the explicit coarse
        inverse, computed on the host and handed to ``pcd_mg_*``.  Patterns
        are fixed, so later calls only refresh values."""
        from .fem.multigrid import galerkin_chain, coarse_inverse
        pc = self.pc
        if pc._mg_chain is None and pc.mg_algebraic:
            from .amg import smoothed_aggregation_chain, PartitionedSA
            blk = getattr(self.engine, "velocity_block", 2) \
                if self.slot == c.KSP_A00 else 1
            prod0 = _producer(self.engine)
            if prod0 is not None:
                # this rank's rows only: aggregation rank by rank, coarse
                # levels cut where the aggregates fall (amg.PartitionedSA)
                A0 = sp.csr_matrix(self._ops[1].A)
                F0 = A0 if blk == 1 else _host.kron_factor(A0, blk)
                if F0 is None:
                    raise RuntimeError(
                        "%spc_type gamg with a partitioned producer needs the "
                        "F (x) I_d velocity block (Picard)" % self._prefix)
                r0, r1 = self.engine.row_range(A0.shape[0],
                                               velocity=self.slot == c.KSP_A00)
                psa = PartitionedSA(
                    F0, (r0 // blk, r1 // blk), prod0.host, block=blk,
                    theta=pc.mg_gamg_threshold,
                    coarse_rows=pc.mg_coarse_eq_limit,
                    replicate_rows=int(os.environ.get("PCD_REPLICATE_BELOW",
                                                      "60000")))
                pc._mg_psa = psa
                pc.setMGInterpolations(
                    [None] + [P if blk == 1 else _host.kron_expand(P, blk)
                              for P in psa.chain()[1:]])
            else:
                pc.setMGInterpolations(smoothed_aggregation_chain(
                    self._ops[1].A, block=blk,
                    coarse_rows=pc.mg_coarse_eq_limit,
                    theta=pc.mg_gamg_threshold,
                    coupled=pc.mg_gamg_coupled))
        if pc._mg_chain is None:
            raise RuntimeError("%spc_type mg needs interpolations "
                               "(pc.setMGInterpolations)" % self._prefix)
        chain = pc._mg_chain
        nlev = pc.mg_levels
        if nlev is None:
            # the coarsest level is the LARGEST one whose explicit inverse
            # stays small: a dense SpMV of a few MB costs less than the ~7
            # short launches of one more level
            nlev = len(chain)
            while nlev > 1 and chain[len(chain) - nlev + 1].shape[0] \
                    <= pc.mg_coarse_eq_limit:
                nlev -= 1
        psa = getattr(pc, "_mg_psa", None)
        if psa is not None and pc.mg_levels is None:
            # (the partitioned hierarchy replicates - and could invert - only
            # what it gathered: never cut it above its first replicated level)
            nlev = max(nlev, len(psa.part) + 1)
        if nlev < len(chain):
            chain = [None] + chain[len(chain) - nlev + 1:]
        lvl_off = 0 if psa is None else psa.nlevels - len(chain)
        skip = range(1, len(chain) - 1) if pc.mg_skip_levels == "all" \
            else pc.mg_skip_levels
        for k in sorted(skip, reverse=True):
            # drop level k: level k+1 interpolates straight from level k-1
            if not 1 <= k < len(chain) - 1:
                raise ValueError("%spc_mg_skip_levels %d: no such intermediate "
                                 "level" % (self._prefix, k))
            sa, sb = _scalar_of(chain[k + 1]), _scalar_of(chain[k])
            if sa is not None and sb is not None and sa[1] == sb[1]:
                merged = _host.kron_expand(sp.csr_matrix(sa[0] @ sb[0]), sa[1])
            else:
                merged = sp.csr_matrix(chain[k + 1] @ chain[k])
                merged.sort_indices()
            chain = chain[:k] + [merged] + chain[k + 2:]
        # The velocity block of the preconditioner matrix is F (x) I_d (the
        # Picard operator couples no components) and so are its prolongations:
        # products, bounds, the coarse inverse and the composed levels are
        # computed on the scalar factor - the same numbers (row sums run in
        # the same order) at 1/d of the work - and expanded for the hand-over.
        blk = getattr(self.engine, "velocity_block", 2) \
            if self.slot == c.KSP_A00 else 1
        A = self._ops[1].A
        chain_s = ops_s = None
        # partitioned producer: operators of partitioned levels hold this
        # rank's rows only (fem/partition.py) - reductions through its HostComm
        prod = _producer(self.engine)
        rep_limit = int(os.environ.get("PCD_REPLICATE_BELOW", "60000"))
        if psa is not None:
            if pc.mg_skip_levels:
                raise ValueError("%spc_mg_skip_levels with a partitioned "
                                 "algebraic hierarchy" % self._prefix)
            psa_part = psa.partitioned_levels()
            rowsparse = lambda l, n: psa_part[l + lvl_off]
        else:
            rowsparse = lambda l, n: prod is not None and (
                l == len(chain) - 1 or n > rep_limit)
        if blk > 1:
            facs = [_scalar_of(P) for P in chain[1:]]
            if all(f is not None and f[1] == blk for f in facs):
                F = _host.kron_factor(A, blk)
                if F is not None:
                    chain_s = [None] + [f[0] for f in facs]
        if pc.mg_galerkin and psa is not None:
            F = A if blk == 1 else _host.kron_factor(A, blk)
            ops_s = psa.operators(F)[lvl_off:]
            chain_s = [None] + [P if blk == 1 else _scalar_of(P)[0]
                                for P in chain[1:]]
            ops = ops_s if blk == 1 else \
                [_host.kron_expand(o, blk) for o in ops_s[:-1]] \
                + [sp.csr_matrix(A)]
            if blk == 1:
                chain_s = ops_s = None
        elif pc.mg_galerkin:
            red = None
            if prod is not None:
                # this rank's terms of every coarse operator: summed over the
                # ranks where the level is replicated, handed to the rows'
                # owners where it is partitioned (the engine's rule: more
                # than PCD_REPLICATE_BELOW rows; level 0 - an explicit
                # inverse - is whole whatever its size)
                from .fem.partition import cut

                def red(C, l):
                    n = C.shape[0]
                    if l >= 1 and n > rep_limit and prod.size > 1:
                        return prod.host.sum_rows(
                            C, cut(n, prod.size, blk)), True
                    return prod.host.sum(C), False
            if red is not None:
                ops = galerkin_chain(A, chain, reduce_level=red)
            elif chain_s is not None:
                ops_s = galerkin_chain(F, chain_s)
                ops = [_host.kron_expand(o, blk) for o in ops_s[:-1]] \
                    + [sp.csr_matrix(A)]
            else:
                ops = galerkin_chain(A, chain)
        else:
            if pc._mg_ops_cb is None:
                raise RuntimeError("%spc_mg_galerkin none needs coarse "
                                   "operators (pc.setMGOperators)"
                                   % self._prefix)
            ops = [sp.csr_matrix(o) for o in pc._mg_ops_cb(len(chain))] \
                + [sp.csr_matrix(A)]
            assert len(ops) == len(chain)
            if chain_s is not None:
                ops_s = [_host.kron_factor(o, blk) for o in ops[:-1]] + [F]
                if any(o is None for o in ops_s):
                    chain_s = ops_s = None
        a, b, cc, d = pc.mg_esteig
        bounds = [None]
        if not hasattr(pc, "_mg_warm"):
            pc._mg_warm = {}
        for l in range(1, len(ops)):
            red = prod.host.sum if rowsparse(l, ops[l].shape[0]) else None
            if ops_s is not None:
                emax = estimate_emax(ops_s[l], iters=12, block=blk,
                                     warm=pc._mg_warm.setdefault(l, {}),
                                     reduce=red)
            else:
                emax = estimate_emax(ops[l], iters=12,
                                     warm=pc._mg_warm.setdefault(l, {}),
                                     reduce=red)
            bounds.append((b * emax, d * emax))
        if ops[0].shape[0] > 30000:
            # (an explicit inverse of that size is tens of GB - e.g. a smoothed-
            # aggregation chain that stalled on a 3-D P2 operator)
            raise ValueError(
                "%spc_type %s: the coarsest level has %d rows - too large for "
                "the explicit coarse inverse (limit 30000); the hierarchy "
                "stalled or has too few levels"
                % (self._prefix, "gamg" if pc.mg_algebraic else "mg",
                   ops[0].shape[0]))
        C = coarse_inverse(ops[0], blk)
        eng, slot, L = self.engine, self.slot, len(ops)
        # kept for statistics (roofline bytes) and for the CPU baseline
        nu_pre = pc.mg_smooth_its if pc.mg_smooth_down is None \
            else pc.mg_smooth_down
        nu_post = pc.mg_smooth_its if pc.mg_smooth_up is None \
            else pc.mg_smooth_up
        pc.mg_data = {"ops": ops, "chain": chain, "bounds": bounds, "C": C,
                      "nu": nu_pre, "nu_post": nu_post}
        sig = (L, nu_pre, nu_post, tuple(o.nnz for o in ops))
        local = getattr(eng, "local_handover", False)
        if local:
            # partitioned levels go over as this rank's rows (the rule of
            # pcd_mg_set_level: more than PCD_REPLICATE_BELOW rows, and the
            # finest level always); replicated ones whole
            limit = int(os.environ.get("PCD_REPLICATE_BELOW", "60000"))
            vel = slot == c.KSP_A00
            cuts = [None] * L
            if psa is not None:
                part = [psa_part[l + lvl_off] for l in range(L)]
                part[0] = False
                for l in range(L):
                    cl = psa.level_cuts()[l + lvl_off]
                    if part[l] and cl is not None:
                        cuts[l] = np.asarray(cl, dtype=np.int64) * blk
            else:
                part = [l == L - 1 or ops[l].shape[0] > limit
                        for l in range(L)]
            me = getattr(prod, "rank", None)
            rng_ = [None if not part[l] else
                    (int(cuts[l][me]), int(cuts[l][me + 1]))
                    if cuts[l] is not None else
                    eng.row_range(ops[l].shape[0], velocity=vel)
                    for l in range(L)]
        # (what a checker needs to put this rank's rows back together:
        # oracle.mirror_partitioned)
        pc.mg_data["rows"] = rng_ if local else None
        if pc._mg_pushed != sig:
            eng.mg_begin(slot, L, nu_pre, nu_post)
            if local:
                for l in range(L - 1):
                    if cuts[l] is not None:
                        eng.mg_set_level_cuts(slot, l, ops[l].shape[0], cuts[l])
            # finest first: a level's engine numbering is inherited from the
            # level above it through the prolongation (pcd_reorder.hpp)
            for l in range(L - 1, 0, -1):
                if local and part[l]:
                    r0, r1 = rng_[l]
                    P = sp.csr_matrix(chain[l])
                    R_rows = None
                    if part[l - 1]:
                        c0, c1 = rng_[l - 1]
                        R_rows = _host.transpose(P)[c0:c1]
                    eng.mg_set_level_local(
                        slot, l, ops[l].shape[0],
                        sp.csr_matrix(ops[l])[r0:r1] if l < L - 1 else None,
                        P[r0:r1], R_rows, *bounds[l])
                else:
                    eng.mg_set_level(slot, l, ops[l] if l < L - 1 else None,
                                     chain[l], *bounds[l])
            eng.mg_set_level(slot, 0, C)
            pc._mg_pushed = sig
        else:
            eng.mg_update_values(slot, 0, C.data)
            for l in range(1, L):
                vals = ops[l].data if l < L - 1 else None
                if vals is not None and local and part[l]:
                    ip = ops[l].indptr
                    vals = vals[ip[rng_[l][0]]:ip[rng_[l][1]]]
                eng.mg_update_values(slot, l, vals, *bounds[l])
        self._push_fused_levels(ops, chain, bounds, ops_s, chain_s, blk)

    def _push_fused_levels(self, ops, chain, bounds, ops_s=None, chain_s=None,
                           blk=1):
        """Small levels of the cycle as pre-composed operators (composed on
        the scalar factors ``ops_s`` / ``chain_s`` when there are any)."""
        from .compose import vcycle_level
        pc, eng, slot = self.pc, self.engine, self.slot
        pc.mg_fused = []
        nu_pre, nu_post = pc.mg_data["nu"], pc.mg_data["nu_post"]
        if not getattr(eng.L, "hip", False) or pc.mg_fuse_nnz <= 0 \
                or nu_pre < 1 or nu_post < 1:
            return
        prod = _producer(eng)
        rep_limit = int(os.environ.get("PCD_REPLICATE_BELOW", "60000"))
        for l in range(1, len(ops)):
            A = ops[l]
            if prod is not None and (l == len(ops) - 1
                                     or A.shape[0] > rep_limit):
                break       # this rank's rows only: nothing to compose from
            grow = 3 if nu_post == 1 else 6     # nnz(W_u) / nnz(A), at least
            if A.shape[0] // blk > pc.mg_fuse_rows \
                    or grow * (A.nnz // blk) > pc.mg_fuse_nnz:
                break       # larger levels are bandwidth-bound: no point in
                #             composing W_u just to measure it
            if ops_s is not None:
                Wd, Wu = vcycle_level(ops_s[l], chain_s[l], bounds[l][0],
                                      bounds[l][1], nu_pre, nu_post)
                nnz_f, nodes = Wu.nnz, ops_s[l].shape[0]
            else:
                Wd, Wu = vcycle_level(A, chain[l], bounds[l][0], bounds[l][1],
                                      nu_pre, nu_post)
                nnz_f, nodes = Wu.nnz // blk, A.shape[0] // blk
            avg = nnz_f / float(max(nodes, 1))
            suits_stream = avg <= pc.mg_fuse_row_nnz
            suits_wave = nodes <= pc.mg_fuse_wave_nodes or avg >= 300.0
            if nnz_f > pc.mg_fuse_nnz or not (suits_stream or suits_wave):
                break
            if ops_s is not None:
                Wd, Wu = _host.kron_expand(Wd, blk), _host.kron_expand(Wu, blk)
            eng.mg_set_fused(slot, l, Wd, Wu)
            pc.mg_fused.append((l, Wd.nnz, Wu.nnz))

    def push_settings(self):
        """Translate the PETSc-style description into ``pcd_set_inner``."""
        pc = self.pc.type if self.pc.type in _SUPPORTED_PC else None
        if pc is None:
            raise ValueError("%spc_type %s is not supported by the HIP engine"
                             % (self._prefix, self.pc.type))
        if pc == "mg":
            if self.type not in ("preonly", "richardson"):
                raise ValueError("%spc_type mg runs under ksp_type preonly or "
                                 "richardson" % self._prefix)
            self._push_multigrid()
            self.engine.set_inner(self.slot, self.type, "mg",
                                  1 if self.type == "preonly" else self.max_it,
                                  0.0, 0.5, 2.0)
            return
        lo, hi = 0.5, 2.0
        rtol = self.rtol
        if self.type == "chebyshev":
            lo, hi = self._chebyshev_bounds()
            # what the engine holds from here on (an estimate re-run later -
            # a warm power iteration goes on from where it stopped - is a
            # slightly different interval, i.e. another polynomial)
            self.cheb_bounds_pushed = (lo, hi)
        if self.type != "cg" or self.norm_type == "none":
            rtol = 0.0                      # fixed iteration count
        max_it = 1 if self.type == "preonly" else self.max_it
        self.precomposed = None
        if self.type == "chebyshev" and pc == "jacobi" \
                and self.cheb_precompose > 0 and 2 <= max_it <= 8 \
                and getattr(self.engine.L, "hip", False) \
                and _producer(self.engine) is None \
                and self._ops[1].A is not None \
                and self._ops[1].A.shape[0] <= self.cheb_precompose_rows:
            # a fixed number of Chebyshev-Jacobi steps from a zero guess is a
            # polynomial in D^-1 A: hand over its factors (the step count is
            # kept for the statistics: these ARE max_it steps)
            from .compose import chebyshev_factors
            F = chebyshev_factors(self._ops[1].A, lo, hi, max_it,
                                  max_factors=self.cheb_precompose)
            self.engine.set_inner_factors(self.slot, F)
            self.engine.set_inner(self.slot, "preonly", "explicit", max_it,
                                  0.0, lo, hi)
            self.precomposed = [W.nnz for W in F]
            return
        self.engine.set_inner(self.slot, self.engine_type, pc, max_it, rtol,
                              lo, hi)

    @property
    def engine_type(self):
        """Name of the solver as the engine knows it."""
        if self.type == "cg" and self.cg_single_reduction:
            return "cgsr"
        return self.type

    def setUp(self):
        if self.engine is not None and self.slot is not None:
            self.push_settings()

    def solve(self, b, x):
        self.engine.inner_solve(self.slot, b.t, x.t, c.MEM_DEVICE)


def estimate_emax(A, jacobi=True, iters=20, seed=0, warm=None, block=1,
                  reduce=None):
    """Largest eigenvalue (modulus) of ``D^-1 A`` by power iteration on the
    host - the stand-in for PETSc's ``-ksp_chebyshev_esteig`` [ext PETSc].
    ``warm``: a dict that carries the iterate between calls (re-estimation
    after a value refresh then needs only a few steps).  ``block`` > 1: ``A``
    is the scalar factor of ``A (x) I_block``; the iteration is the expanded
    operator's (same start vector, same sums) on ``block`` interleaved
    vectors at once.  ``reduce``: ``A`` holds this rank's rows only (a
    partitioned producer: global shape, other rows empty); the product is
    completed by ``reduce`` (sum over ranks of vectors with disjoint
    supports - exact), so every rank iterates on the same global vector."""
    A = sp.csr_matrix(A)
    n = A.shape[0] * block
    d = A.diagonal().copy()
    d[d == 0.0] = 1.0
    dinv = 1.0 / d if jacobi else np.ones_like(d)
    if block > 1:
        dinv = np.repeat(dinv, block)
    v = None if warm is None else warm.get("v")
    cold = v is None or v.size != n
    if cold:
        v = np.random.default_rng(seed).standard_normal(n)
    full = iters
    if not cold:
        iters = max(3, iters // 4)
    lam, best, it = 1.0, 0.0, 0
    if A.nnz * block > 4000000 and not _host.use_numpy():
        # large operators: the threaded native SpMV (same row sums as scipy's)
        apply = _host.SpMV(A, dinv, nvec=block)
    elif block > 1:
        apply = lambda u: dinv * (A @ u.reshape(-1, block)).ravel()
    else:
        apply = lambda u: dinv * (A @ u)
    if reduce is not None:
        apply_rows = apply
        apply = lambda u: reduce(apply_rows(u))
    while it < iters:
        v = v / np.linalg.norm(v)
        w = apply(v)
        lam = np.linalg.norm(w)
        v = w
        it += 1
        # non-normal operators (convection, SUPG) make |D^-1 A v| / |v|
        # oscillate instead of increasing: keep the envelope - an
        # over-estimate only slows the smoother, an under-estimate breaks it
        if cold and it <= 2:
            continue
        best = max(best, lam)
        # a warm start is only trusted while the estimate stays near the last
        # one; after a large change of the operator (Stokes -> Oseen step)
        # iterate as long as a cold start would
        if it == iters and iters < full and \
                abs(lam - warm.get("lam", lam)) > 0.1 * lam:
            iters = full
    lam = max(best, lam) if best > 0.0 else lam
    if warm is not None:
        warm["v"] = v
        warm["lam"] = float(lam)
    return float(lam)


def _producer(engine):
    """The partitioned producer (``fem/partition.PartitionedProblem``) whose
    rows this engine received, or ``None`` (global hand-over)."""
    prod = getattr(engine, "producer", None)
    return prod if hasattr(prod, "host") and hasattr(prod, "fine") else None


def _scalar_of(P):
    """``(F, block)`` when ``P`` was built as ``F (x) I_block``
    (``_host.kron_expand``), else ``None``."""
    F = getattr(P, "kron_scalar", None)
    return None if F is None else (F, int(P.kron_block))


class Sys(object):
    @staticmethod
    def getVersion():
        return (0, 0, 0)

    @staticmethod
    def getVersionInfo():
        return {"release": False}

    @staticmethod
    def pushErrorHandler(name):
        return None
