"""ctypes binding of the C ABI declared in ``include/pcd_engine.h``.

The product binds ``libpcd_hip.so`` (prefix ``pcd_``) and nothing else, and
raises if it is missing.  (The test oracle exports the same signatures under
``pcdo_``; the subclass that binds it lives in ``oracle/__init__.py`` - test
infrastructure - not here.)
"""

import ctypes as C
import os

import numpy as np

# enums of include/pcd_engine.h
BRM1, BRM2, RBRM1, RBRM2 = 1, 2, 3, 4
MAT_AP, MAT_MP, MAT_KP, MAT_RP, MAT_A00, MAT_A01, MAT_A = range(7)
KSP_AP, KSP_MP, KSP_RP, KSP_A00 = range(4)
PREONLY, RICHARDSON, CHEBYSHEV, CG, CG_SR = range(5)
PC_NONE, PC_JACOBI, PC_MG, PC_EXPLICIT = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
INFO_N_U, INFO_N_P, INFO_ITS_AP, INFO_ITS_MP, INFO_ITS_RP, INFO_ITS_A00, \
    INFO_NUM_PCD_APPLY, INFO_NUM_FS_APPLY, INFO_GMRES_ITS, \
    INFO_GMRES_RNORM, INFO_N_U_LOCAL, INFO_N_P_LOCAL, INFO_A00_COMPONENTS, \
    INFO_A00_ROWS_PER_WG, INFO_RANKS = range(15)
INFO_REORDERED = 15          # HIP engine only: +1 velocity, +2 pressure renumbered
INFO_LAUNCHES = 64           # HIP engine only: kernel launches of this host thread
INFO_PEER_CALLS, INFO_BOOT_CALLS = 65, 66   # exchanges / reductions: peer kernels, bootstrap
INFO_A00_KERNEL = 69         # 0 csr, 1 stream, 2 multi-component stream, 3 tiles, 4 lane-major tiles
INFO_PEER_DECLINED = 68      # halo channels that did not fit the peer arena (bootstrap path)
INFO_A00_MODEL_BYTES = 67    # bytes one Chebyshev step on A00 moves by construction
INFO_NNZ_BASE = 16

KSP_TYPES = {"preonly": PREONLY, "richardson": RICHARDSON,
             "chebyshev": CHEBYSHEV, "cg": CG, "cgsr": CG_SR}
PC_TYPES = {"none": PC_NONE, "jacobi": PC_JACOBI, "mg": PC_MG,
            "explicit": PC_EXPLICIT}
VARIANTS = {"BRM1": BRM1, "BRM2": BRM2, "RBRM1": RBRM1, "RBRM2": RBRM2}

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)

# name -> (argtypes after the handle, needs handle)
_SIGNATURES = {
    "destroy": [],
    "set_csr": [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                C.c_void_p],
    "update_values": [C.c_int, C.c_void_p, C.c_int],
    "set_system": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_int64, C.c_void_p, C.c_int64, C.c_void_p],
    "update_system": [C.c_void_p, C.c_void_p, C.c_int],
    "set_bc": [C.c_int64, C.c_void_p, C.c_void_p],
    "set_inner": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                  C.c_double],
    "mg_begin": [C.c_int, C.c_int, C.c_int, C.c_int],
    "mg_set_level": [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_double, C.c_double],
    "mg_update_values": [C.c_int, C.c_int, C.c_void_p, C.c_double,
                         C.c_double, C.c_int],
    "setup": [],
    "apply": [C.c_void_p, C.c_void_p, C.c_int],
    "fieldsplit_apply": [C.c_void_p, C.c_void_p, C.c_int],
    "gmres_solve": [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                    C.c_int, C.c_int, C.POINTER(C.c_int), _f64p],
    "spmv": [C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "inner_solve": [C.c_int, C.c_void_p, C.c_void_p, C.c_int],
    "apply_bc": [C.c_void_p, C.c_int],
    "get_info": [C.c_int, _f64p],
    "synchronize": [],
}
# entry points only the HIP library has
_HIP_ONLY = {
    "set_reorder": [C.c_int],
    # rank-local hand-over (partitioned runs)
    "set_csr_local": [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                      C.c_void_p, C.c_void_p],
    "row_range": [C.c_int, C.c_int64, C.c_void_p, C.c_void_p],
    "set_system_local": [C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                         C.c_void_p, C.c_void_p, C.c_void_p],
    "mg_set_level_cuts": [C.c_int, C.c_int, C.c_int64, C.c_void_p],
    "mg_set_level_local": [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_double, C.c_double],
    "set_stream": [C.c_void_p],
    "comm_init": [C.c_int, C.c_int, C.c_void_p],
    "comm_init_threads": [C.c_int, C.c_int, C.POINTER(C.c_void_p)],
    "comm_init_host": [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "graph_enable": [C.c_int],
    "probe_a00_step": [C.c_void_p, C.c_void_p, C.c_int, _f64p,
                       C.POINTER(C.c_int)],
    "bandwidth_probe": [C.c_int, C.c_int64, C.c_int, _f64p],
    "set_velocity_block": [C.c_int],
    # pre-composed inner solves
    "mg_set_fused": [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                     C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                     C.c_void_p, C.c_void_p],
    "set_inner_factor": [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                         C.c_void_p, C.c_void_p],
    # device operator producer
    "fe_begin": [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                 C.c_void_p, C.c_void_p],
    "fe_set_level": [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                     C.c_void_p, C.c_void_p],
    "fe_set_level_galerkin": [C.c_int, C.c_int64, C.c_int64, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p],
    "fe_set_level_product": [C.c_int, C.c_int64, C.c_int64] + [C.c_void_p] * 12,
    "fe_set_level_product_rows": [C.c_int, C.c_int64, C.c_int64, C.c_int64]
    + [C.c_void_p] * 12 + [C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int,
                           C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                           C.c_int64, C.c_int64],
    "fe_set_residual_rows": [C.c_int],
    "fe_set_supg": [C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_int,
                    C.c_void_p, C.c_void_p, C.c_void_p],
    "fe_bind_pattern": [C.c_int, C.c_int64, C.c_void_p, C.c_void_p],
    "fe_set_newton": [C.c_int, C.c_void_p],
    "fe_bind_system": [C.c_void_p],
    "fe_bind_kp": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double],
    "fe_set_kp_const": [C.c_void_p],
    "fe_set_rows": [C.c_int, C.c_int64, C.c_int64],
    "fe_set_kp_rows": [C.c_int64, C.c_int64],
    "fe_bind_robin": [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                      C.c_void_p],
    "fe_bind_mg": [C.c_int, C.c_double, C.c_double, C.c_int],
    "fe_bind_coarse_inverse": [C.c_int64, C.c_void_p, C.c_void_p],
    "fe_update": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
    "fe_bind_residual": [C.c_void_p] * 6 + [C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_void_p,
                                            C.c_double],
    "fe_set_bc_values": [C.c_void_p],
    "fe_set_previous": [C.c_void_p, C.c_int],
    "fe_residual": [C.c_void_p, C.c_void_p, C.c_int, _f64p],
    "fe_picard_solve": [C.c_void_p, C.c_int, C.c_double, C.c_double,
                        C.c_double, C.c_int, C.c_double, C.c_double,
                        C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int),
                        C.c_void_p, C.c_void_p, C.POINTER(C.c_int)],
    "fe_get_level_values": [C.c_int, C.c_void_p],
    "fe_get_newton_values": [C.c_int, C.c_void_p],
    "fe_get_kp_values": [C.c_void_p],
    "fe_get_bounds": [C.c_int, _f64p, _f64p],
}

#: every symbol include/pcd_engine.h declares (checked by the CPU test-suite)
DECLARED_SYMBOLS = (["pcd_create", "pcd_last_error", "pcd_comm_unique_id",
                     "pcd_dist_probe"]
                    + ["pcd_" + k for k in _SIGNATURES]
                    + ["pcd_" + k for k in _HIP_ONLY])


class EngineError(RuntimeError):
    """Raised for any nonzero status crossing the C ABI (SURVEY 8b Errors)."""


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):           # torch tensor
        return a.data_ptr()
    return int(a)


def _out(a, n=None, what="output"):
    """Pointer of an OUTPUT / in-place vector.  The engine writes ``n``
    doubles through it, so anything but a writeable C-contiguous float64
    buffer of at least that length would be silent memory corruption."""
    if isinstance(a, np.ndarray):
        if a.dtype != np.float64 or not a.flags.c_contiguous \
                or not a.flags.writeable or a.ndim != 1:
            raise EngineError("%s vector must be a writeable C-contiguous 1-d "
                              "float64 array (got %s, contiguous=%s, "
                              "writeable=%s)" % (what, a.dtype,
                                                 a.flags.c_contiguous,
                                                 a.flags.writeable))
        if n is not None and a.size < n:
            raise EngineError("%s vector holds %d entries, the engine writes "
                              "%d" % (what, a.size, n))
        return a.ctypes.data
    if hasattr(a, "data_ptr"):           # torch tensor
        import torch
        if a.dtype != torch.float64 or not a.is_contiguous():
            raise EngineError("%s tensor must be contiguous float64 (got %s)"
                              % (what, a.dtype))
        if n is not None and a.numel() < n:
            raise EngineError("%s tensor holds %d entries, the engine writes "
                              "%d" % (what, a.numel(), n))
        return a.data_ptr()
    if a is None:
        raise EngineError("%s vector is None" % what)
    return int(a)                        # raw pointer: the caller's business


def _in(a, mem):
    """Pointer of an INPUT vector; host arrays are normalised to contiguous
    float64 (the returned object must be kept alive by the caller)."""
    if mem == MEM_HOST and not hasattr(a, "data_ptr") \
            and not isinstance(a, int):
        a = _f64(a)
    elif hasattr(a, "data_ptr"):
        import torch
        if a.dtype != torch.float64 or not a.is_contiguous():
            raise EngineError("input tensor must be contiguous float64 "
                              "(got %s)" % a.dtype)
    return a


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Library(object):
    """The loaded ``libpcd_hip.so``: every symbol of pcd_engine.h bound."""

    prefix = "pcd_"
    hip = True

    def __init__(self, path):
        if not os.path.exists(path):
            raise EngineError(
                "%s not found: build it first (python -c 'import "
                "__graft_entry__ as g; g.build()')" % path)
        self.path = path
        prefix, hip = self.prefix, self.hip
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL if hip else C.DEFAULT_MODE)
        self.fn = {}
        sigs = dict(_SIGNATURES)
        if hip:
            sigs.update(_HIP_ONLY)
        for name, args in sigs.items():
            f = getattr(self.lib, prefix + name)
            f.argtypes = [C.c_void_p] + args
            f.restype = C.c_int
            self.fn[name] = f
        self.create = getattr(self.lib, prefix + "create")
        self.create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
        self.create.restype = C.c_int
        self.last_error = getattr(self.lib, prefix + "last_error")
        self.last_error.argtypes = []
        self.last_error.restype = C.c_char_p
        if hip:
            self.comm_unique_id = getattr(self.lib, prefix + "comm_unique_id")
            self.comm_unique_id.argtypes = [C.c_void_p]
            self.comm_unique_id.restype = C.c_int

    def error(self):
        msg = self.last_error()
        return msg.decode("utf-8", "replace") if msg else ""


class Engine(object):
    """One engine handle = one PCD context (+ its fieldsplit/GMRES shell)."""

    def __init__(self, library, variant="BRM1", device=0):
        self.L = library
        self._h = C.c_void_p()
        v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
        rc = library.create(C.byref(self._h), v, int(device))
        if rc:
            raise EngineError("create failed (%d): %s" % (rc, library.error()))
        self.variant = v
        self.shapes = {}
        #: set by ``set_system_local``: operators and multigrid levels are
        #: handed over as this rank's rows (petsc.DeviceMat, _push_multigrid)
        self.local_handover = False

    # -- plumbing -----------------------------------------------------------
    def _call(self, name, *args):
        if not self._h:
            raise EngineError("engine handle already destroyed")
        rc = self.L.fn[name](self._h, *args)
        if rc:
            raise EngineError("%s%s failed (%d): %s"
                              % (self.L.prefix, name, rc, self.L.error()))

    def destroy(self):
        if self._h:
            self.L.fn["destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    # -- operators ----------------------------------------------------------
    def set_csr(self, which, A):
        """``A``: scipy CSR (or anything with indptr/indices/data/shape)."""
        ip, ix, dv = _i32(A.indptr), _i32(A.indices), _f64(A.data)
        self.shapes[which] = A.shape
        self._call("set_csr", which, A.shape[0], A.shape[1], _ptr(ip),
                   _ptr(ix), _ptr(dv))

    def set_reorder(self, mode):
        """Engine renumbering: "none" | "auto" | "always" (before set_system)."""
        self._call("set_reorder", {"none": 0, "auto": 1, "always": 2}[mode])

    def row_range(self, n_global, velocity=False):
        """Rows ``[r0, r1)`` of a field this rank owns."""
        r0, r1 = C.c_int64(), C.c_int64()
        self._call("row_range", int(bool(velocity)), int(n_global),
                   C.byref(r0), C.byref(r1))
        return int(r0.value), int(r1.value)

    def set_csr_local(self, which, A_rows, shape):
        """Rank-local hand-over: ``A_rows`` = this rank's rows (scipy CSR,
        GLOBAL column ids), ``shape`` = global shape."""
        ip, ix, dv = _i32(A_rows.indptr), _i32(A_rows.indices), \
            _f64(A_rows.data)
        self.shapes[which] = tuple(shape)
        self._call("set_csr_local", which, shape[0], shape[1],
                   A_rows.shape[0], _ptr(ip), _ptr(ix), _ptr(dv))

    def update_values(self, which, vals, mem=MEM_HOST):
        if mem == MEM_HOST:
            vals = _f64(vals)
        self._call("update_values", which, _ptr(vals), mem)

    def set_system(self, A, is_u, is_p, P=None):
        ip, ix, dv = _i32(A.indptr), _i32(A.indices), _f64(A.data)
        pv = None
        if P is not None:
            assert P.nnz == A.nnz
            pv = _f64(P.data)
        iu, ipp = _i32(is_u), _i32(is_p)
        self.n_u, self.n_p = iu.size, ipp.size
        self._call("set_system", A.shape[0], _ptr(ip), _ptr(ix), _ptr(dv),
                   _ptr(pv), iu.size, _ptr(iu), ipp.size, _ptr(ipp))
        self.local_handover = False

    def set_system_local(self, A_rows, rows, n, is_u, is_p, P_rows=None):
        """Rank-local hand-over of the system: ``A_rows`` = this rank's rows
        of the monolithic matrix (scipy CSR, the caller's GLOBAL column ids),
        ``rows`` their global indices - velocity rows of ``row_range`` first,
        then the pressure rows; the index sets whole."""
        ip, ix, dv = _i32(A_rows.indptr), _i32(A_rows.indices), \
            _f64(A_rows.data)
        pv = None
        if P_rows is not None:
            assert P_rows.nnz == A_rows.nnz
            pv = _f64(P_rows.data)
        iu, ipp, rw = _i32(is_u), _i32(is_p), _i32(rows)
        self.n_u, self.n_p = iu.size, ipp.size
        self._call("set_system_local", int(n), iu.size, _ptr(iu), ipp.size,
                   _ptr(ipp), rw.size, _ptr(rw), _ptr(ip), _ptr(ix), _ptr(dv),
                   _ptr(pv))
        self.local_handover = True

    def update_system(self, vals, pvals=None, mem=MEM_HOST):
        if mem == MEM_HOST:
            vals = _f64(vals)
            pvals = None if pvals is None else _f64(pvals)
        self._call("update_system", _ptr(vals), _ptr(pvals), mem)

    def set_bc(self, idx, vals):
        idx, vals = _i32(idx), _f64(vals)
        self._call("set_bc", idx.size, _ptr(idx), _ptr(vals))

    def set_inner(self, slot, ksp_type, pc_type="jacobi", max_it=1,
                  rtol=0.0, emin=0.5, emax=2.0):
        k = KSP_TYPES[ksp_type] if isinstance(ksp_type, str) else ksp_type
        p = PC_TYPES[pc_type] if isinstance(pc_type, str) else pc_type
        self._call("set_inner", slot, k, p, int(max_it), float(rtol),
                   float(emin), float(emax))

    def setup(self):
        self._call("setup")

    # -- multigrid ----------------------------------------------------------
    def mg_begin(self, slot, nlevels, nu_pre=2, nu_post=2):
        self._call("mg_begin", slot, int(nlevels), int(nu_pre), int(nu_post))

    def mg_set_level(self, slot, level, A=None, P=None, emin=0.0, emax=0.0):
        a = [0, None, None, None]
        if A is not None:
            ip, ix, dv = _i32(A.indptr), _i32(A.indices), _f64(A.data)
            a = [A.shape[0], _ptr(ip), _ptr(ix), _ptr(dv)]
        b = [0, 0, None, None, None]
        if P is not None:
            pp, px, pv = _i32(P.indptr), _i32(P.indices), _f64(P.data)
            b = [P.shape[0], P.shape[1], _ptr(pp), _ptr(px), _ptr(pv)]
        self._call("mg_set_level", slot, int(level), *(a + b),
                   float(emin), float(emax))

    def mg_set_level_cuts(self, slot, level, n, bounds):
        """Row cuts of a partitioned coarse level (``pcd_mg_set_level_cuts``):
        ``bounds`` has one more entry than there are ranks."""
        b = np.ascontiguousarray(bounds, dtype=np.int64)
        self._call("mg_set_level_cuts", slot, int(level), int(n), _ptr(b))

    def mg_set_level_local(self, slot, level, n, A_rows, P_rows, R_rows,
                           emin, emax):
        """Rank-local hand-over of a partitioned level: this rank's rows of
        the operator (``None`` on the finest level), of the prolongation, and
        of its transpose (``None`` when the level below is replicated)."""
        a = [0, None, None, None]
        keep = []
        if A_rows is not None:
            t = (_i32(A_rows.indptr), _i32(A_rows.indices), _f64(A_rows.data))
            keep.append(t)
            a = [A_rows.shape[0], _ptr(t[0]), _ptr(t[1]), _ptr(t[2])]
        else:
            a[0] = P_rows.shape[0]
        pt = (_i32(P_rows.indptr), _i32(P_rows.indices), _f64(P_rows.data))
        r = [0, None, None, None]
        if R_rows is not None:
            t = (_i32(R_rows.indptr), _i32(R_rows.indices), _f64(R_rows.data))
            keep.append(t)
            r = [R_rows.shape[0], _ptr(t[0]), _ptr(t[1]), _ptr(t[2])]
        self._call("mg_set_level_local", slot, int(level), int(n), *(
            a + [P_rows.shape[1], _ptr(pt[0]), _ptr(pt[1]), _ptr(pt[2])] + r
            + [float(emin), float(emax)]))

    def mg_update_values(self, slot, level, vals, emin=0.0, emax=0.0,
                         mem=MEM_HOST):
        if vals is not None and mem == MEM_HOST:
            vals = _f64(vals)
        self._call("mg_update_values", slot, int(level), _ptr(vals),
                   float(emin), float(emax), mem)

    def mg_set_fused(self, slot, level, Wd=None, Wu=None):
        """Pre-composed form of one level (``compose.vcycle_level``); ``None``
        drops it."""
        if Wd is None:
            self._call("mg_set_fused", slot, int(level), 0, 0, None, None,
                       None, 0, 0, None, None, None)
            return
        a = (_i32(Wd.indptr), _i32(Wd.indices), _f64(Wd.data))
        b = (_i32(Wu.indptr), _i32(Wu.indices), _f64(Wu.data))
        self._call("mg_set_fused", slot, int(level), Wd.shape[0], Wd.shape[1],
                   _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), Wu.shape[0],
                   Wu.shape[1], _ptr(b[0]), _ptr(b[1]), _ptr(b[2]))

    def set_inner_factors(self, slot, factors):
        """``x = factors[-1] @ ... @ factors[0] @ b`` (pc type "explicit")."""
        for k, W in enumerate(factors):
            ip, ix, dv = _i32(W.indptr), _i32(W.indices), _f64(W.data)
            self._call("set_inner_factor", slot, k, len(factors), W.shape[0],
                       _ptr(ip), _ptr(ix), _ptr(dv))

    # -- hot path -----------------------------------------------------------
    # Host-pointer calls carry GLOBAL vectors, so their lengths are known
    # here and outputs are checked; device-pointer calls of a partitioned
    # engine carry the rank's slice (length known to the engine only).
    def _n_p(self, mem):
        s = self.shapes.get(MAT_AP)
        return s[0] if (s and mem == MEM_HOST) else None

    def _n_sys(self, mem):
        n = getattr(self, "n_u", None)
        return n + self.n_p if (n is not None and mem == MEM_HOST) else None

    def apply(self, x, y, mem=MEM_HOST):
        x = _in(x, mem)
        self._call("apply", _ptr(x), _out(y, self._n_p(mem), "apply: y"), mem)

    def fieldsplit_apply(self, x, y, mem=MEM_HOST):
        x = _in(x, mem)
        self._call("fieldsplit_apply", _ptr(x),
                   _out(y, self._n_sys(mem), "fieldsplit_apply: y"), mem)

    def gmres_solve(self, b, x, mem=MEM_HOST, rtol=1e-6, atol=0.0,
                    restart=150, max_it=10000):
        its, rn = C.c_int(0), C.c_double(0.0)
        b = _in(b, mem)
        self._call("gmres_solve", _ptr(b),
                   _out(x, self._n_sys(mem), "gmres_solve: x"), mem,
                   float(rtol), float(atol), int(restart), int(max_it),
                   C.byref(its), C.byref(rn))
        return its.value, rn.value

    def spmv(self, which, x, y, mem=MEM_HOST):
        x = _in(x, mem)
        s = self.shapes.get(which)
        n = s[0] if (s and mem == MEM_HOST) else None
        if which == MAT_A:
            n = self._n_sys(mem)
        elif which in (MAT_A00, MAT_A01) and mem == MEM_HOST:
            n = getattr(self, "n_u", n)
        self._call("spmv", which, _ptr(x), _out(y, n, "spmv: y"), mem)

    def inner_solve(self, slot, b, x, mem=MEM_HOST):
        b = _in(b, mem)
        n = self._n_p(mem) if slot != KSP_A00 else (
            getattr(self, "n_u", None) if mem == MEM_HOST else None)
        self._call("inner_solve", slot, _ptr(b),
                   _out(x, n, "inner_solve: x"), mem)

    def apply_bc(self, x, mem=MEM_HOST):
        self._call("apply_bc", _out(x, self._n_p(mem), "apply_bc: x"), mem)

    def info(self, key):
        out = C.c_double(0.0)
        self._call("get_info", int(key), C.byref(out))
        return out.value

    def synchronize(self):
        self._call("synchronize")

    # -- HIP only -------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._call("set_stream", C.c_void_p(stream_ptr))

    def set_velocity_block(self, ncomp):
        self.velocity_block = int(ncomp)
        if self.L.hip:
            self._call("set_velocity_block", int(ncomp))

    def probe_a00_step(self, x, y, reps=3):
        """(us per launch, launches timed) of the fused Chebyshev step on the
        finest velocity operator INSIDE eager fieldsplit applies (device
        vectors)."""
        us, cnt = C.c_double(0.0), C.c_int(0)
        self._call("probe_a00_step", _ptr(x), _ptr(y), int(reps),
                   C.byref(us), C.byref(cnt))
        return us.value, cnt.value

    def graph_enable(self, on=True):
        self._call("graph_enable", int(bool(on)))

    def bandwidth_probe(self, kind="triad", nbytes=1 << 30, reps=5):
        """GB/s (reads + writes) of a streaming kernel of the library: copy,
        triad, read-only sweep, read-mostly (6 % writes)."""
        out = C.c_double(0.0)
        self._call("bandwidth_probe",
                   {"copy": 0, "triad": 1, "read": 2, "read_mostly": 3,
                    "read_nt": 4}[kind],
                   int(nbytes), int(reps), C.byref(out))
        return out.value

    # -- device operator producer (HIP only) ------------------------------------
    def fe_begin(self, dim, nlevels, qw, phi, dphi, psi):
        qw, phi, dphi, psi = _f64(qw), _f64(phi), _f64(dphi), _f64(psi)
        self._call("fe_begin", int(dim), int(nlevels), int(qw.size), _ptr(qw),
                   _ptr(phi), _ptr(dphi), _ptr(psi))

    def fe_set_level(self, level, dofs2, gradlam, measure, f_ptr, f_src,
                     f_const, f_keep, diag_pos, diag_val, inject, nn2):
        """Arrays already component-major (see include/pcd_engine.h)."""
        dofs2, f_ptr, f_src = _i32(dofs2), _i32(f_ptr), _i32(f_src)
        gradlam, measure = _f64(gradlam), _f64(measure)
        f_const, diag_val = _f64(f_const), _f64(diag_val)
        f_keep = np.ascontiguousarray(f_keep, dtype=np.uint8)
        diag_pos = _i32(diag_pos)
        inj = None if inject is None else _i32(inject)
        self._call("fe_set_level", int(level), int(measure.size), int(nn2),
                   _ptr(dofs2), _ptr(gradlam), _ptr(measure),
                   int(f_const.size), _ptr(f_ptr), _ptr(f_src), _ptr(f_const),
                   _ptr(f_keep), int(diag_pos.size), _ptr(diag_pos),
                   _ptr(diag_val), _ptr(inj))

    def fe_set_level_galerkin(self, level, b_ptr, b_src, b_w, c_ptr, c_src,
                              c_w):
        b_ptr = np.ascontiguousarray(b_ptr, dtype=np.int64)
        c_ptr = np.ascontiguousarray(c_ptr, dtype=np.int64)
        b_src, c_src = _i32(b_src), _i32(c_src)
        b_w, c_w = _f64(b_w), _f64(c_w)
        self._call("fe_set_level_galerkin", int(level), int(c_ptr.size - 1),
                   int(b_ptr.size - 1), _ptr(b_ptr), _ptr(b_src), _ptr(b_w),
                   _ptr(c_ptr), _ptr(c_src), _ptr(c_w))

    def fe_set_level_product(self, level, P, PT, f_indptr, f_indices,
                             b_indptr, b_indices, c_indptr, c_indices):
        """Galerkin level by the numeric sparse product on fixed patterns:
        scalar ``P`` (fine x coarse) and its transpose ``PT`` (scipy CSR,
        sorted), the pattern of the finer level's ``F`` and the structural
        patterns of ``B = F P`` and ``F_c = P^T B``."""
        keep = [_i32(P.indptr), _i32(P.indices), _f64(P.data),
                _i32(PT.indptr), _i32(PT.indices), _f64(PT.data),
                _i32(f_indptr), _i32(f_indices), _i32(b_indptr),
                _i32(b_indices), _i32(c_indptr), _i32(c_indices)]
        self._call("fe_set_level_product", int(level), int(P.shape[0]),
                   int(P.shape[1]), *[_ptr(a) for a in keep])

    def fe_set_level_product_rows(self, level, Pext, PT, f_indptr, f_indices,
                                  b_indptr, b_indices, t_indptr, t_indices,
                                  n_out, wire_len, sends, adds, pos,
                                  gather_off=0, gather_total=0, node_row0=0,
                                  n_node_rows=0):
        """Galerkin level BY ROWS (``pcd_fe_set_level_product_rows``):
        ``Pext`` (global shape: own + halo rows), ``PT`` = transpose of the
        own rows, the patterns, and the exchange lists."""
        keep = [_i32(Pext.indptr), _i32(Pext.indices), _f64(Pext.data),
                _i32(PT.indptr), _i32(PT.indices), _f64(PT.data),
                _i32(f_indptr), _i32(f_indices), _i32(b_indptr),
                _i32(b_indices), _i32(t_indptr), _i32(t_indices)]
        sends = np.ascontiguousarray(sends, dtype=np.int64).reshape(-1, 3)
        adds = np.ascontiguousarray(adds, dtype=np.int64).reshape(-1, 4)
        pos = _i32(pos)
        self._call("fe_set_level_product_rows", int(level),
                   int(len(f_indptr) - 1), int(Pext.shape[0]),
                   int(Pext.shape[1]), *([_ptr(a) for a in keep] + [
                       int(n_out), int(wire_len), int(sends.shape[0]),
                       _ptr(sends), int(adds.shape[0]), _ptr(adds), _ptr(pos),
                       int(gather_off), int(gather_total), int(node_row0),
                       int(n_node_rows)]))

    def fe_set_residual_rows(self, on=True):
        self._call("fe_set_residual_rows", int(bool(on)))

    def fe_set_kp_rows(self, entry_offset, nnz_global):
        self._call("fe_set_kp_rows", int(entry_offset), int(nnz_global))

    def fe_set_rows(self, level, node_row0, n_node_rows):
        self._call("fe_set_rows", int(level), int(node_row0), int(n_node_rows))

    def fe_set_supg(self, level, cell_h, nu, phi_mid, qw_s, phi_s, dphi_s):
        cell_h, phi_mid = _f64(cell_h), _f64(phi_mid)
        qw_s, phi_s, dphi_s = _f64(qw_s), _f64(phi_s), _f64(dphi_s)
        self._call("fe_set_supg", int(level), _ptr(cell_h), float(nu),
                   _ptr(phi_mid), int(qw_s.size), _ptr(qw_s), _ptr(phi_s),
                   _ptr(dphi_s))

    def fe_bind_pattern(self, level, indptr, indices):
        ip, ix = _i32(indptr), _i32(indices)
        self._call("fe_bind_pattern", int(level), int(ip.size - 1), _ptr(ip),
                   _ptr(ix))

    def fe_set_newton(self, level, pos):
        pos = _i32(np.asarray(pos).ravel())
        self._call("fe_set_newton", int(level), _ptr(pos))

    def fe_bind_system(self, sys_pos):
        sys_pos = np.ascontiguousarray(sys_pos, dtype=np.int64)
        self._call("fe_bind_system", _ptr(sys_pos))

    def fe_bind_kp(self, kp_ptr, kp_src, kp_const, scale):
        kp_ptr, kp_src = _i32(kp_ptr), _i32(kp_src)
        cst = None if kp_const is None else _f64(kp_const)
        self._call("fe_bind_kp", int(kp_ptr.size - 1), _ptr(kp_ptr),
                   _ptr(kp_src), _ptr(cst), float(scale))

    def fe_set_kp_const(self, kp_const):
        cst = None if kp_const is None else _f64(kp_const)
        self._call("fe_set_kp_const", _ptr(cst))

    def fe_bind_robin(self, nodes, normals, lengths, aff_pos, aff_ptr,
                      aff_src, aff_w):
        """``nodes`` (3, nb), ``normals`` (2, nb): component-major (3-D:
        (6, nb) and (3, nb), ``lengths`` = areas of the boundary faces)."""
        nodes, aff_pos, aff_src = _i32(nodes), _i32(aff_pos), _i32(aff_src)
        normals, lengths, aff_w = _f64(normals), _f64(lengths), _f64(aff_w)
        aff_ptr = np.ascontiguousarray(aff_ptr, dtype=np.int64)
        self._call("fe_bind_robin", int(lengths.size), _ptr(nodes),
                   _ptr(normals), _ptr(lengths), int(aff_pos.size),
                   _ptr(aff_pos), _ptr(aff_ptr), _ptr(aff_src), _ptr(aff_w))

    def fe_bind_mg(self, slot, emin_factor, emax_factor, iters=12):
        self._call("fe_bind_mg", int(slot), float(emin_factor),
                   float(emax_factor), int(iters))

    def fe_bind_coarse_inverse(self, indptr, indices):
        ip, ix = _i32(indptr), _i32(indices)
        self._call("fe_bind_coarse_inverse", int(ip.size - 1), _ptr(ip),
                   _ptr(ix))

    def fe_update(self, xu, v=None, ru=None, mem=MEM_HOST):
        self._call("fe_update", _ptr(xu), _ptr(v), _ptr(ru), mem)

    def fe_bind_residual(self, Bt, B, bc_idx, bc_mult, mass_vals=None,
                         idt=0.0):
        keep = [_i32(Bt.indptr), _i32(Bt.indices), _f64(Bt.data),
                _i32(B.indptr), _i32(B.indices), _f64(B.data),
                _i32(bc_idx), _f64(bc_mult),
                None if mass_vals is None else _f64(mass_vals)]
        self._call("fe_bind_residual", *[_ptr(a) for a in keep[:6]],
                   int(keep[6].size), _ptr(keep[6]), _ptr(keep[7]),
                   _ptr(keep[8]), float(idt))

    def fe_set_bc_values(self, g):
        g = _f64(g)
        self._call("fe_set_bc_values", _ptr(g))

    def fe_set_previous(self, u0, mem=MEM_HOST):
        if u0 is not None and mem == MEM_HOST:
            u0 = _f64(u0)
        self._call("fe_set_previous", _ptr(u0), mem)

    def fe_residual(self, x, b, mem=MEM_HOST):
        nrm = C.c_double(0.0)
        self._call("fe_residual", _ptr(x), _ptr(b), mem, C.byref(nrm))
        return nrm.value

    def fe_picard_solve(self, x, mem=MEM_HOST, r0=0.0, rtol=1e-9, atol=1e-10,
                        max_it=50, relax=1.0, lin_rtol=1e-6, lin_atol=0.0,
                        restart=150, lin_max_it=10000):
        """Returns (iterations, converged, GMRES counts, residual norms);
        ``x`` is updated in place."""
        n_it, conv = C.c_int(0), C.c_int(0)
        lin = np.zeros(max(max_it, 1), dtype=np.int32)
        res = np.zeros(max_it + 1)
        self._call("fe_picard_solve",
                   _out(x, self._n_sys(mem), "fe_picard_solve: x"), mem, float(r0), float(rtol),
                   float(atol), int(max_it), float(relax), float(lin_rtol),
                   float(lin_atol), int(restart), int(lin_max_it),
                   C.byref(n_it), _ptr(lin), _ptr(res), C.byref(conv))
        k = n_it.value
        return k, bool(conv.value), [int(v) for v in lin[:k]], \
            [float(v) for v in res[:k + 1]]

    def fe_level_values(self, level, nnz):
        out = np.empty(int(nnz))
        self._call("fe_get_level_values", int(level), _ptr(out))
        return out

    def fe_newton_values(self, level, nnz, dim):
        out = np.empty((dim * dim, int(nnz)))
        self._call("fe_get_newton_values", int(level), _ptr(out))
        return out

    def fe_kp_values(self, nnz):
        out = np.empty(int(nnz))
        self._call("fe_get_kp_values", _ptr(out))
        return out

    def fe_bounds(self, level):
        a, b = C.c_double(0.0), C.c_double(0.0)
        self._call("fe_get_bounds", int(level), C.byref(a), C.byref(b))
        return a.value, b.value

    def comm_init(self, rank, nranks, unique_id_bytes):
        _map_torch_rccl()
        buf = C.create_string_buffer(bytes(unique_id_bytes), 128)
        self._call("comm_init", int(rank), int(nranks), buf)

    def comm_init_threads(self, rank, nranks, group):
        """Test backend; ``group`` is a ``ctypes.c_void_p`` shared by all."""
        self._call("comm_init_threads", int(rank), int(nranks),
                   C.byref(group))

    def comm_init_host(self, rank, nranks, transport):
        """Communicator over a host transport (``pcd_comm_init_host``):
        ``transport.allreduce(array)`` sums a float64 array in place over the
        ranks, ``transport.exchange(sends, recvs)`` with lists of ``(peer,
        array)`` is one neighbour exchange (``parallel.TorchHostTransport``;
        an mpi4py communicator wraps the same way)."""
        def ar(ctx, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(int(count),))
                transport.allreduce(a)
                return 0
            except Exception:               # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def ex(ctx, ns, sp, sb, sc, nr, rp, rb, rc):
            try:
                sends = [(int(sp[i]), np.ctypeslib.as_array(
                    sb[i], shape=(int(sc[i]),)) if sc[i] else np.zeros(0))
                    for i in range(ns)]
                recvs = [(int(rp[i]), np.ctypeslib.as_array(
                    rb[i], shape=(int(rc[i]),)) if rc[i] else np.zeros(0))
                    for i in range(nr)]
                transport.exchange(sends, recvs)
                return 0
            except Exception:               # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        dp = C.POINTER(C.c_double)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_int64)
        EX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                         C.POINTER(dp), C.POINTER(C.c_int64), C.c_int,
                         C.POINTER(C.c_int), C.POINTER(dp),
                         C.POINTER(C.c_int64))
        # (the callbacks must outlive the communicator)
        self._host_cbs = (AR(ar), EX(ex), transport)
        self._call("comm_init_host", int(rank), int(nranks),
                   C.cast(self._host_cbs[0], C.c_void_p),
                   C.cast(self._host_cbs[1], C.c_void_p), None)

    # -- numpy conveniences used by tests ------------------------------------
    def apply_np(self, x):
        x = _f64(x)
        y = np.empty_like(x)
        self.apply(x, y)
        return y

    def fieldsplit_apply_np(self, x):
        x = _f64(x)
        y = np.empty_like(x)
        self.fieldsplit_apply(x, y)
        return y

    def spmv_np(self, which, x, nrows):
        x = _f64(x)
        y = np.empty(nrows)
        self.spmv(which, x, y)
        return y

    def inner_solve_np(self, slot, b):
        b = _f64(b)
        x = np.empty_like(b)
        self.inner_solve(slot, b, x)
        return x

    def gmres_np(self, b, **kw):
        b = _f64(b)
        x = np.zeros_like(b)
        its, rn = self.gmres_solve(b, x, **kw)
        return x, its, rn


def dist_probe(A, rank, nranks, even_rows=False, even_cols=False):
    """Row block / halo plan of ``rank`` for the global CSR ``A`` (host only).
    Returns dict(local=csr with ghost columns appended, row0, col0, nghost,
    send={peer: local indices}, recv={peer: (ghost_begin, ghost_end)})."""
    import scipy.sparse as sp
    lib = hip_library()
    f = lib.lib.pcd_dist_probe
    f.restype = C.c_int
    ip, ix, dv = _i32(A.indptr), _i32(A.indices), _f64(A.data)
    nr, nc, nnz = A.shape[0], A.shape[1], A.nnz
    counts = np.zeros(7, dtype=np.int64)
    orp, oc, ov = (np.zeros(nr + 1, np.int32), np.zeros(max(nnz, 1), np.int32),
                   np.zeros(max(nnz, 1)))
    sp_, so, si = (np.zeros(nranks, np.int32), np.zeros(nranks + 1, np.int32),
                   np.zeros(max(nc, 1), np.int32))
    rp_, ro = np.zeros(nranks, np.int32), np.zeros(nranks + 1, np.int32)
    args = [C.c_int64(nr), C.c_int64(nc)] + [C.c_void_p(_ptr(a))
                                             for a in (ip, ix, dv)] \
        + [C.c_int(rank), C.c_int(nranks), C.c_int(int(even_rows)),
           C.c_int(int(even_cols))] \
        + [C.c_void_p(_ptr(a)) for a in (counts, orp, oc, ov, sp_, so, si,
                                         rp_, ro)]
    rc = f(*args)
    if rc:
        raise EngineError("dist_probe failed (%d): %s" % (rc, lib.error()))
    nl, ncl, ng, ns, nrv, row0, col0 = (int(v) for v in counts)
    lnnz = int(orp[nl])
    local = sp.csr_matrix((ov[:lnnz], oc[:lnnz], orp[:nl + 1]),
                          shape=(nl, ncl + ng))
    return {"local": local, "row0": row0, "col0": col0, "ncols_owned": ncl,
            "nghost": ng,
            "send": {int(sp_[i]): si[so[i]:so[i + 1]].copy()
                     for i in range(ns)},
            "recv": {int(rp_[i]): (int(ro[i]), int(ro[i + 1]))
                     for i in range(nrv)}}


def _map_torch_rccl():
    """PyTorch ships its own ``librccl.so`` (SONAME ``librccl.so.1``).  The
    engine binds RCCL at run time and takes the copy that is already mapped;
    if it loads ROCm's copy FIRST and torch is imported later, the loader does
    not recognise the two as the same library, both end up in the process and
    their static state is torn down twice at exit (``double free or
    corruption``).  Importing torch before the first RCCL call keeps it at one
    copy whenever torch is installed."""
    try:
        import torch  # noqa: F401
    except ImportError:                   # a host without torch: ROCm's copy
        pass


def comm_unique_id():
    """128-byte ncclUniqueId (rank 0 creates it, the host broadcasts it)."""
    _map_torch_rccl()
    buf = C.create_string_buffer(128)
    lib = hip_library()
    rc = lib.comm_unique_id(buf)
    if rc:
        raise EngineError("comm_unique_id failed (%d): %s" % (rc, lib.error()))
    return buf.raw


_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIBRARY_PATH = os.path.join(_HERE, "lib", "libpcd_hip.so")
_hip_library = None


def hip_library():
    """The product library.  Fails loudly when it has not been built."""
    global _hip_library
    if _hip_library is None:
        # FENAPACK_AMD_HIP_LIB: another BUILD of the same HIP library
        # (compile-time A/B switches of the kernels, tools/tile_sweep.sh)
        _hip_library = Library(os.environ.get("FENAPACK_AMD_HIP_LIB")
                               or HIP_LIBRARY_PATH)
    return _hip_library
