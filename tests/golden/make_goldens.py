#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (``/root/reference`` must exist):

    python tests/golden/make_goldens.py [output directory]

(``tests/test_goldens_regenerate.py`` re-runs this into a temporary directory
and compares with the committed fixtures.)

The reference cannot be imported as a package here (``dolfin``, ``petsc4py``
are absent; ``fenapack/__init__.py:28-29`` needs them), but
``fenapack/preconditioners.py``, ``fenapack/utils.py`` and
``fenapack/field_split_backend.py`` can be loaded by file path under stub
modules - the same mock-module trick the reference's docs build uses
(``doc/source/conf.py:29-45``).  The four ``apply`` bodies and
``PCDInterface._build_approx_Ap`` then run unmodified against numpy-backed
fake ``Vec``/``Mat``/``KSP`` objects.  What this pins: operation order, signs,
where the subfield BC is applied, the R_p construction.  What it cannot pin:
PETSc's own arithmetic (the fakes below stand in for it).

Outputs are data only (CSR arrays, index lists, input/output vectors); no
reference source text is stored.
"""

import importlib.util
import os
import sys
import types

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("FENAPACK_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from fenapack_amd.fem import BackwardStep, Cavity          # noqa: E402
from oracle import reference_numpy as rn                   # noqa: E402


# ---------------------------------------------------------------- stub layer
def _install_stubs():
    dolfin = types.ModuleType("dolfin")
    dolfin.timed = lambda name: (lambda f: f)
    dolfin.MPI = types.SimpleNamespace(size=lambda comm: 1)
    dolfin.has_lu_solver_method = lambda m: m == "superlu"

    class _Timer(object):
        def __init__(self, *a):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    dolfin.Timer = _Timer
    dolfin.PETScMatrix = type("PETScMatrix", (), {})
    dolfin.DirichletBC = type("DirichletBC", (), {})
    dolfin.SystemAssembler = type("SystemAssembler", (), {})
    dolfin.assemble = lambda *a, **k: None
    petsc4py = types.ModuleType("petsc4py")
    PETSc = types.ModuleType("petsc4py.PETSc")
    PETSc.Sys = types.SimpleNamespace(
        getVersion=lambda: (3, 12, 0),
        getVersionInfo=lambda: {"release": True})
    PETSc.Comm = type("Comm", (), {})
    PETSc.IS = type("IS", (), {})
    PETSc.Mat = types.SimpleNamespace(Option=types.SimpleNamespace(SPD=1))
    petsc4py.PETSc = PETSc
    pkg = types.ModuleType("fenapack")
    pkg.__path__ = [os.path.join(REF, "fenapack")]
    fsu = types.ModuleType("fenapack._field_split_utils")
    fsu.SubfieldBC = type("SubfieldBC", (), {})
    sys.modules.update({"dolfin": dolfin, "petsc4py": petsc4py,
                        "petsc4py.PETSc": PETSc, "fenapack": pkg,
                        "fenapack._field_split_utils": fsu})

    def load(name):
        spec = importlib.util.spec_from_file_location(
            "fenapack." + name, os.path.join(REF, "fenapack", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["fenapack." + name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("utils")
    load("assembling")
    return load("preconditioners"), load("field_split_backend")


class Vec(object):
    """The petsc4py Vec methods the reference touches."""

    def __init__(self, a):
        self.a = np.array(a, dtype=np.float64)

    def copy(self, result=None):
        if result is None:
            return Vec(self.a)
        result.a[:] = self.a
        return result

    def duplicate(self):
        return Vec(np.zeros_like(self.a))

    def axpy(self, alpha, x):
        self.a += alpha * x.a

    def scale(self, alpha):
        self.a *= alpha

    def reciprocal(self):
        self.a = 1.0 / self.a

    def sqrtabs(self):
        self.a = np.sqrt(np.abs(self.a))


class Mat(object):
    def __init__(self, A):
        self.A = sp.csr_matrix(A)

    def mult(self, x, y):
        y.a[:] = self.A @ x.a

    def getSize(self):
        return self.A.shape

    def getVecLeft(self):
        return Vec(np.zeros(self.A.shape[0]))

    def getDiagonal(self, result=None):
        result.a[:] = self.A.diagonal()

    def duplicate(self):
        return Mat(self.A * 0.0)

    def copy(self, result=None):
        result.A = self.A.copy()
        return result

    def diagonalScale(self, L=None, R=None):
        if L is not None:
            self.A = sp.diags(L.a) @ self.A
        if R is not None:
            self.A = self.A @ sp.diags(R.a)

    def transposeMatMult(self, B, result=None):
        return Mat((self.A.T @ B.A).tocsr())

    def isAssembled(self):
        return True

    def setOptionsPrefix(self, p):
        pass


class KSP(object):
    def __init__(self, solve):
        self._solve = solve

    def solve(self, b, x):
        x.a[:] = self._solve(b.a)


# ----------------------------------------------------------------- the cases
ITER_CFG = {                     # fixed-iteration fakes (demo :157-165 shape)
    "Ap": ("cg", 8, 0.0),
    "Mp": ("chebyshev", 5, 0.5, 2.0),
    "Rp": ("cg", 6, 0.0),
}
DIRECT_CFG = {"Ap": ("direct",), "Mp": ("direct",), "Rp": ("direct",)}


def picard_state(pb, steps=2):
    """Matrices frozen at Picard iterate ``steps``."""
    import scipy.sparse.linalg as spla
    xu, xp = pb.initial_guess()
    for _ in range(steps):
        L = pb.linearise(xu, xp)
        A = sp.bmat([[L["A00"], L["A01"]], [L["A10"], None]]).tocsc()
        A = A + 1e-12 * sp.identity(A.shape[0], format="csc") \
            if isinstance(pb, Cavity) else A
        dx = spla.spsolve(A, np.concatenate([L["bu"], L["bp"]]))
        xu = xu - dx[:pb.space.n_u]
        xp = xp - dx[pb.space.n_u:]
    return xu, xp


def csr_dict(prefix, A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return {prefix + "_indptr": A.indptr.astype(np.int32),
            prefix + "_indices": A.indices.astype(np.int32),
            prefix + "_data": A.data.astype(np.float64),
            prefix + "_shape": np.array(A.shape, dtype=np.int64)}


def make_case(pre, fsb, name, pb, seed=0):
    V = pb.space
    xu, _ = picard_state(pb)
    Ap, Mp, Kp = pb.Ap, pb.Mp, pb.Kp(xu)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(V.n_p)
    bc_idx, bc_val = pb.bc_p_idx.astype(np.int32), pb.bc_p_val

    def bcs_applier(vec):        # SubfieldBC.apply == VecSetValues(INSERT)
        vec.a[bc_idx] = bc_val

    out = {"x": x, "bc_idx": bc_idx, "bc_val": bc_val,
           "n_u": np.int64(V.n_u), "n_p": np.int64(V.n_p)}
    out.update(csr_dict("Ap", Ap))
    out.update(csr_dict("Mp", Mp))
    out.update(csr_dict("Kp", Kp))

    # R_p by the reference's own PCDInterface._build_approx_Ap
    Mu = pb.Mu()
    iface = object.__new__(fsb.PCDInterface)
    Rp = iface._build_approx_Ap(Mat(Mu), Mat(pb.A01), None).A
    # scipy's product stores an entry unless its terms cancel to an EXACT
    # zero, so the stored pattern depends on rounding in the last bit; the
    # fixture keeps entries above 1e-13 of the largest one
    Rp.data[np.abs(Rp.data) < 1e-13 * np.abs(Rp.data).max()] = 0.0
    Rp.eliminate_zeros()
    Rp.sort_indices()
    assert abs(Rp - pb.Rp()).max() < 1e-10 * abs(Rp).max()
    out.update(csr_dict("Rp", Rp))
    out.update(csr_dict("A01", pb.A01))
    out["Mu_diag"] = Mu.diagonal()

    classes = {"BRM1": pre.PCDPC_BRM1, "BRM2": pre.PCDPC_BRM2,
               "RBRM1": pre.PCDRPC_BRM1, "RBRM2": pre.PCDRPC_BRM2}
    for tag, cfg in (("direct", DIRECT_CFG), ("iter", ITER_CFG)):
        sAp = rn.make_inner(Ap, cfg["Ap"])
        sMp = rn.make_inner(Mp, cfg["Mp"])
        # R_p = B D^-1 B^T of an ENCLOSED flow is singular (constant
        # pressures): an "exact" solve with it is noise, not a golden - and
        # whether the factorisation even completes depends on the last bit
        no_rp = tag == "direct" and isinstance(pb, Cavity)
        sRp = None if no_rp else rn.make_inner(Rp, cfg["Rp"])
        for var, cls in classes.items():
            if no_rp and var.startswith("R"):
                continue
            ctx = cls()
            ctx.ksp_Ap, ctx.ksp_Mp, ctx.ksp_Rp = KSP(sAp), KSP(sMp), KSP(sRp)
            ctx.mat_Kp = Mat(Kp)
            ctx.bcs_applier = bcs_applier
            xv, yv = Vec(x), Vec(np.zeros_like(x))
            ctx.apply(None, xv, yv)
            assert np.array_equal(xv.a, x), "x must be left untouched"
            # second call reuses the cached work vecs (preconditioners.py:52)
            y2 = Vec(np.zeros_like(x))
            ctx.apply(None, xv, y2)
            assert np.array_equal(yv.a, y2.a)
            hand = rn.pcd_apply(var, x, Ap, Mp, Kp, bc_idx, bc_val, sAp, sMp,
                                sRp)
            assert np.array_equal(hand, yv.a), (name, var, tag)
            out["y_%s_%s" % (var, tag)] = yv.a
    out["iter_cfg"] = np.array(repr(ITER_CFG))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s n_p=%5d  %7.1f kB" % (name, V.n_p,
                                        os.path.getsize(path) / 1024.0))


OUT = HERE


def main():
    global OUT
    if len(sys.argv) > 1:
        OUT = sys.argv[1]
        os.makedirs(OUT, exist_ok=True)
    pre, fsb = _install_stubs()
    cases = []
    for lvl in (0, 1, 2):
        cases.append(("pcd_lshape_l%d" % lvl,
                      lambda v, lvl=lvl: BackwardStep(lvl, nu=0.02, variant=v,
                                                      dt=0.2)))
    for lvl in (0, 1):
        cases.append(("pcd_cavity_l%d" % lvl,
                      lambda v, lvl=lvl: Cavity(lvl, nu=0.01, variant=v,
                                                dt=0.2)))
    for name, mk in cases:
        # BRM1 and BRM2 differ in the BC location and the Robin term in Kp:
        # one fixture per PCD BC flavour
        # (the enclosed cavity has neither inlet nor outlet: one flavour)
        for flavour in (("BRM1",) if "cavity" in name else ("BRM1", "BRM2")):
            make_case(pre, fsb, "%s_%s" % (name, flavour.lower()),
                      mk(flavour))


if __name__ == "__main__":
    main()
