"""``PCDInterface``: turns a ``PCDAssembler`` into the engine's inputs - the
p-p / u-u / u-p submatrices, the subfield BC index lists and the const vs.
non-const update cadence - mirroring ``fenapack/field_split_backend.py:30-342``
method for method.  Where the reference hands a submatrix to a PETSc KSP
(``ksp.setOperators`` + ``ksp.setUp()`` = factorisation), this hands it to an
engine slot (``pcd_set_csr`` + ``pcd_set_inner``)."""

import numpy as np

from . import _cabi as c
from .assembling import PCDAssembler
from .petsc import Mat, DeviceMat, IS, Vec
from .timing import Timer


class SubfieldBC(object):
    """Mixed-space Dirichlet BC mapped onto the contiguous subfield numbering
    (``fenapack/SubfieldBC.h:92-160``): for every owned subfield index ``i``
    whose global dof is in the BC map, emit ``(i + rank_offset, value)``."""

    def __init__(self, bc, subfield_is, rank_offset=0):
        bv = bc.get_boundary_values()
        gdofs = np.fromiter(bv.keys(), dtype=np.int64, count=len(bv))
        gvals = np.fromiter(bv.values(), dtype=np.float64, count=len(bv))
        order = np.argsort(gdofs)
        gdofs, gvals = gdofs[order], gvals[order]
        sub = subfield_is.getIndices().astype(np.int64)
        pos = np.searchsorted(gdofs, sub)
        pos[pos == gdofs.size] = 0
        hit = gdofs[pos] == sub if gdofs.size else np.zeros(sub.size, bool)
        self.indices = (np.nonzero(hit)[0] + rank_offset).astype(np.int32)
        self.values = gvals[pos[hit]]

    def is_homogeneous(self):
        return bool(np.all(self.values == 0.0))

    def get_boundary_values(self):
        return dict(zip(self.indices.tolist(), self.values.tolist()))


class PCDInterface(object):
    def __init__(self, pcd_assembler, A, is_u, is_p, deep_submats=False,
                 engine=None):
        assert isinstance(pcd_assembler, PCDAssembler)
        assert isinstance(is_u, IS) and isinstance(is_p, IS)
        assert isinstance(deep_submats, bool)
        self.assembler = pcd_assembler
        self.A = A
        self.is_u, self.is_p = is_u, is_p
        self.engine = engine
        self.scratch = {}
        self._bcs_uploaded = False

    # -- BCs (field_split_backend.py:62-64, 294-308) -------------------------
    def subfield_bc_arrays(self):
        """``(indices int32, values float64)`` of the PCD Dirichlet condition
        in the contiguous pressure numbering - what ``pcd_set_bc`` takes
        (``SubfieldBC.h:92-160`` for every bc of ``bcs_pcd``, merged the way
        ``apply_bcs`` applies them: one after the other with INSERT, so on a
        duplicate index the last bc wins; ``field_split_backend.py:294-308``)."""
        subbcs = getattr(self, "_subbcs", None)
        if subbcs is None:
            bcs = self.assembler.pcd_bcs()
            bcs = bcs if isinstance(bcs, (list, tuple)) else [bcs]
            self._subbcs = subbcs = [SubfieldBC(bc, self.is_p) for bc in bcs]
        merged = {}
        for bc in subbcs:
            merged.update(bc.get_boundary_values())
        idx = np.fromiter(merged.keys(), dtype=np.int32, count=len(merged))
        val = np.fromiter(merged.values(), dtype=np.float64,
                          count=len(merged))
        return idx, val

    def _subbcs_upload(self):
        if not self._bcs_uploaded:
            self.engine.set_bc(*self.subfield_bc_arrays())
            self._bcs_uploaded = True

    def apply_pcd_bcs(self, vec):
        """Apply bcs to an intermediate pressure vector of the PCD PC."""
        self._subbcs_upload()
        self.engine.apply_bc(vec.t, c.MEM_DEVICE)

    # -- KSP operators -------------------------------------------------------
    def setup_ksp_Ap(self, ksp):
        self.setup_ksp(ksp, self.assembler.ap, self.is_p, c.MAT_AP, c.KSP_AP,
                       spd=True,
                       const=self.assembler.get_pcd_form("ap").is_constant())

    def setup_ksp_Mp(self, ksp):
        self.setup_ksp(ksp, self.assembler.mp, self.is_p, c.MAT_MP, c.KSP_MP,
                       spd=True,
                       const=self.assembler.get_pcd_form("mp").is_constant())

    def setup_mat_Kp(self, mat=None):
        if mat is None or not self.assembler.get_pcd_form("kp").is_constant():
            return self._assemble_operator_deep(self.assembler.kp, self.is_p,
                                                submat=mat, which=c.MAT_KP)

    def setup_mat_Mu(self, mat=None):
        if mat is None or not self.assembler.get_pcd_form("mu").is_constant():
            return self._assemble_operator_deep(self.assembler.mu, self.is_u,
                                                submat=mat, which=None)

    def setup_mat_Bt(self, mat=None):
        if mat is None or not self.assembler.get_pcd_form("gp").is_constant():
            if self.assembler.get_pcd_form("gp").is_phantom():
                # Bt is the 01-block of the system matrix (with velocity BCs)
                return self._get_deep_submat(self.A, self.is_u, self.is_p,
                                             submat=mat)
            return self._assemble_operator_deep(self.assembler.gp, self.is_u,
                                                self.is_p, submat=mat,
                                                which=None)

    def setup_ksp_Rp(self, ksp, Mu, Bt):
        mat = ksp.getOperators()[0]
        const = self.assembler.get_pcd_form("mu").is_constant() \
            and self.assembler.get_pcd_form("gp").is_constant()
        if mat.type is None or not mat.isAssembled() or not const:
            first = not isinstance(mat, DeviceMat)
            R = self._build_approx_Ap(Mu, Bt, None if first else mat)
            R.setOption(Mat.Option.SPD, True)
            if first:
                mat = DeviceMat(self.engine, c.MAT_RP, R.A, R.comm)
            else:
                mat.update(R.A)
            mat.setOptionsPrefix(ksp.getOptionsPrefix())
            ksp.setOperators(mat, mat)
            ksp.bind(self.engine, c.KSP_RP)
            self._give_interpolations(ksp)
            with Timer("FENaPack: {} setup".format(
                    ksp.getOptionsPrefix() or "")):      # :138
                ksp.setUp()

    def _build_approx_Ap(self, Mu, Bt, mat=None):
        """``R_p = B diag(Mu)^-1 B^T`` built exactly like
        field_split_backend.py:142-166 (host side, setup time)."""
        diagMu, = self.get_work_vecs_from_square_mat(Mu, 1)
        Ap, = self.get_work_mats(Bt, 1)
        Mu.getDiagonal(result=diagMu)
        diagMu.reciprocal()
        diagMu.sqrtabs()
        Bt.copy(result=Ap)
        Ap.diagonalScale(L=diagMu)
        R = Ap.transposeMatMult(Ap)
        prod = getattr(getattr(self, "engine", None), "producer", None)
        if prod is not None and getattr(prod, "size", 1) > 1:
            # rows of B^T by ranks: the product is this rank's TERMS of all
            # rows; each row's terms go to its owner (MatTransposeMatMult on
            # an MPI matrix does the same)
            from .fem.partition import cut
            n = R.A.shape[0]
            R.set(prod.host.sum_rows(R.A, cut(n, prod.size, 1)))
        return R

    def _give_interpolations(self, ksp):
        """pc_type mg on a pressure-space KSP: hand over the P1 chain."""
        if ksp.pc.type == "mg" and ksp.pc._mg_chain is None \
                and not ksp.pc.mg_algebraic:
            V = self.assembler.function_space()
            ksp.pc.setMGInterpolations(V.interpolations().chain("p"))

    def _cached(self, name, num, factory):
        items = self.__dict__.get(name)
        if items is None:
            items = tuple(factory() for _ in range(num))
            setattr(self, name, items)
        elif len(items) != num:
            raise ValueError("Changing number of %s not allowed"
                             % name.strip("_").replace("_", " "))
        return items

    def get_work_vecs_from_square_mat(self, M, num):
        rows, cols = M.getSize()
        assert rows == cols
        return self._cached("_work_vecs", num, M.getVecLeft)

    def get_work_mats(self, M, num):
        return self._cached("_work_mats", num, M.duplicate)

    def setup_ksp(self, ksp, assemble_func, iset, which, slot, spd=False,
                  const=False):
        """Assemble into the operator of ``ksp`` if not yet assembled; when
        the form is not constant, re-assemble into the existing submatrix."""
        mat = ksp.getOperators()[0]
        if mat.type is None or not mat.isAssembled():
            work = Mat()
            assemble_func(work)
            sub = self._get_deep_submat(work, iset)
            mat = DeviceMat(self.engine, which, sub.A, sub.comm)
            mat.setOption(Mat.Option.SPD, spd)
            mat.setOptionsPrefix(ksp.getOptionsPrefix())
            ksp.setOperators(mat, mat)
            ksp.bind(self.engine, slot)
            self._give_interpolations(ksp)
            with Timer("FENaPack: {} setup".format(
                    ksp.getOptionsPrefix() or "")):      # :254
                ksp.setUp()
        elif not const:
            work = Mat()
            assemble_func(work)
            mat.update(self._get_deep_submat(work, iset).A)
            with Timer("FENaPack: {} setup".format(
                    ksp.getOptionsPrefix() or "")):      # :262
                ksp.setUp()

    def _assemble_operator_deep(self, assemble_func, isrow, iscol=None,
                                submat=None, which=None):
        work = Mat()
        assemble_func(work)
        sub = self._get_deep_submat(work, isrow, iscol)
        if which is None:
            return sub if submat is None else submat.set(sub.A)
        if submat is None:
            return DeviceMat(self.engine, which, sub.A, sub.comm)
        submat.update(sub.A)
        return submat

    @staticmethod
    def _get_deep_submat(mat, isrow, iscol=None, submat=None):
        return mat.createSubMatrix(isrow, iscol, submat=submat)
