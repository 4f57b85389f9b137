"""Device operator producer: the operators that change with the iterate are
assembled in HBM by the engine (``pcd_fe_*``) instead of on the host.

What the reference re-assembles through DOLFIN every nonlinear iteration
(``fenapack/assembling.py:151-155`` system matrix, ``:165-171`` ``kp``; the
residual ``F`` of ``fenapack/nonlinear_solvers.py:85-112``) is, for the fixed
P2/P1 Picard forms of ``demo_navier-stokes-pcd.py:104-137``, a function of the
velocity iterate only:

* the velocity block ``F x I_d`` with ``F = nu K + idt M + C(w)`` on every
  multigrid level (coarse levels re-discretised with the injected iterate,
  ``-pc_mg_galerkin none``, or Galerkin products of the finest one); with the
  Newton linearisation (``--nls newton``, ``demo_navier-stokes-pcd.py:42,
  113-116``) the coupled block ``F x I_d + N(w)``, ``N_ij = (phi_b d_j w_i,
  phi_a)``, on the pattern ``(pattern of F) x ones(d, d)``,
* ``Kp = (1/nu) (w . grad p, q) [+ idt/nu (p, q)]``,
* the residual part ``(F_unconstrained x I_d) (x_u - d)``.

This module builds the static plans (element -> CSR-entry contribution lists,
Dirichlet masks, injection maps, positions in the caller's system values) from
the host spaces once, hands them to the engine, and drives the Picard loop
with ``pcd_fe_update`` + the outer GMRES; per iteration the host only applies
constant blocks to vectors and inverts the coarsest scalar operator.
"""

import os
import time

import numpy as np
import scipy.sparse as sp

from . import _cabi as c
from .fem.multigrid import dense_csr, injection_map

__all__ = ["DeviceProducer", "DevicePicardSolver", "DeviceNonlinearSolver",
           "solve_steady_device", "solve_unsteady_device"]


def _contribution_plan(inv, ncells, nloc2, nnz, pattern=None):
    """CSR-of-contributions for a FixedPattern: entry k sums the element
    values ``f_src[f_ptr[k]:f_ptr[k+1]]``; element storage is component-major
    (``ab * ncells + cell``), contributions in ascending cell order (the order
    ``numpy.bincount`` adds them on the host)."""
    from . import _host
    inv = np.asarray(inv).ravel()
    if pattern is not None and not _host.use_numpy():
        ptr, order = pattern.members()     # grouped once, with the pattern
    elif _host.use_numpy():
        order = np.argsort(inv, kind="stable")
        ptr = np.zeros(nnz + 1, dtype=np.int64)
        np.cumsum(np.bincount(inv, minlength=nnz), out=ptr[1:])
    else:
        g = _host.group_pairs(inv, None, nnz)      # stable counting sort
        gptr, order = g.members()
        ptr = gptr[g.indptr]                       # (slots without members: empty)
        g.release()
    assert order.size < 2 ** 31
    return ptr.astype(np.int32), _host.contribution_src(order, nloc2, ncells)


def _expand_by_rows(P, row_of):
    """For every item t with ``row_of[t] = i``: all entries of row ``i`` of
    the CSR matrix ``P``.  Returns (item index, entry index) pairs."""
    plen = np.diff(P.indptr)
    cnt = plen[row_of]
    rep = np.repeat(np.arange(row_of.size, dtype=np.int64), cnt)
    start = np.cumsum(cnt) - cnt
    off = np.arange(rep.size, dtype=np.int64) - np.repeat(start, cnt)
    return rep, P.indptr[row_of[rep]].astype(np.int64) + off


def _group(rows, cols, nrows, ncols):
    """Distinct (row, col) pairs in CSR order + CSR-of-members (stable:
    members keep input order).  Returns (urows, ucols, ptr, order)."""
    from . import _host
    if _host.use_numpy():
        ukey, inv = np.unique(rows * ncols + cols, return_inverse=True)
        inv = inv.ravel()
        order = np.argsort(inv, kind="stable")
        ptr = np.zeros(ukey.size + 1, dtype=np.int64)
        np.cumsum(np.bincount(inv, minlength=ukey.size), out=ptr[1:])
        return ukey // ncols, ukey % ncols, ptr, order
    g = _host.group_pairs(rows, cols, nrows)
    ptr, order = g.members()
    urows = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(g.indptr))
    ucols = g.ucols
    g.release()
    return urows, ucols, ptr, order


def block_positions(A, indptr_f, indices_f, d):
    """Where the ``d x d`` blocks of the scalar pattern ``(indptr_f,
    indices_f)`` sit in the values of the CSR matrix ``A`` (interleaved dofs,
    sorted indices): ``pos[i * d + j, k]`` = position of entry ``(d r_k + i,
    d c_k + j)``."""
    A = sp.csr_matrix(A)
    if not A.has_sorted_indices:
        raise ValueError("block_positions: the operator's indices must be "
                         "sorted")
    n = A.shape[1]
    rows_a = np.repeat(np.arange(A.shape[0], dtype=np.int64),
                       np.diff(A.indptr))
    keys = rows_a * n + A.indices
    rows = np.repeat(np.arange(indptr_f.size - 1, dtype=np.int64),
                     np.diff(indptr_f))
    cols = np.asarray(indices_f, dtype=np.int64)
    pos = np.empty((d * d, cols.size), dtype=np.int64)
    for i in range(d):
        for j in range(d):
            key = (d * rows + i) * n + d * cols + j
            p = np.searchsorted(keys, key)
            if p.max(initial=0) >= keys.size or np.any(keys[p] != key):
                raise ValueError("block_positions: the operator lacks entries "
                                 "of the coupled pattern")
            pos[i * d + j] = p
    return pos


def coupled_pattern(indptr_f, indices_f, n, d):
    """CSR pattern ``(pattern of F) x ones(d, d)`` (values 1)."""
    ones = sp.csr_matrix((np.ones(len(indices_f)), indices_f, indptr_f),
                         shape=(n, n))
    K = sp.kron(ones, np.ones((d, d)), format="csr")
    K.sort_indices()
    return K


#: refuse Galerkin plans beyond this many (entry, weight) pairs: the lists are
#: built with numpy temporaries several times their size
MAX_GALERKIN_PAIRS = int(os.environ.get("FENAPACK_AMD_MAX_GALERKIN_PAIRS",
                                        "1000000000"))


MAX_GALERKIN_PAIRS_NUMPY = 400000000


def galerkin_plan(rows_f, cols_f, P):
    """Fixed-pattern numeric ``P^T F P``: two weighted-gather stages.

    ``rows_f, cols_f``: CSR-ordered pattern of the fine scalar operator;
    ``P``: scalar prolongation (fine nodes x coarse nodes).  Returns
    ``(b_ptr, b_src, b_w, c_ptr, c_src, c_w, indptr_c, indices_c)``."""
    from . import _host
    P = sp.csr_matrix(P)
    P.sort_indices()
    ncoarse = P.shape[1]
    rows_f = np.asarray(rows_f, dtype=np.int64)
    cols_f = np.asarray(cols_f, dtype=np.int64)
    est = float(np.diff(P.indptr).mean()) * rows_f.size
    if not _host.use_numpy() and est <= MAX_GALERKIN_PAIRS:
        # native builder (libpcd_host, threads over the rows): the same plans,
        # term for term, without the numpy temporaries - config 5's first
        # algebraic level (440 M terms) in seconds instead of 42
        nf = P.shape[0]
        indptr_f = np.zeros(nf + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows_f, minlength=nf), out=indptr_f[1:])
        Fpat = sp.csr_matrix((np.ones(cols_f.size), cols_f.astype(np.int32),
                              indptr_f.astype(np.int32)), shape=(nf, nf))
        Fpat.has_sorted_indices = True
        brp, bcol, b_ptr, b_src, b_w = _host.product_plan(Fpat, P, 0)
        Bpat = sp.csr_matrix((np.ones(bcol.size), bcol, brp.astype(np.int32)),
                             shape=(nf, ncoarse))
        Bpat.has_sorted_indices = True
        crp, ccol, c_ptr, c_src, c_w = _host.product_plan(
            _host.transpose(P), Bpat, 1)
        if c_src.size > MAX_GALERKIN_PAIRS:
            raise ValueError("Galerkin plan of %d pairs exceeds the limit; "
                             "use -pc_mg_galerkin none" % c_src.size)
        return (b_ptr, b_src, b_w, c_ptr, c_src, c_w, crp.astype(np.int32),
                ccol)
    # (the numpy builder makes temporaries several times the list size: its
    # cap stays at 4e8 pairs whatever the native builder is allowed)
    if est > min(MAX_GALERKIN_PAIRS, MAX_GALERKIN_PAIRS_NUMPY):
        raise ValueError("Galerkin plan of ~%.3g pairs exceeds the limit; use "
                         "-pc_mg_galerkin none" % est)
    # B = F P: entry (i, J) collects F[i, j] * P[j, J]
    k_rep, pidx = _expand_by_rows(P, cols_f)
    rows_b, cols_b, b_ptr, order = _group(
        rows_f[k_rep], P.indices[pidx].astype(np.int64), P.shape[0], ncoarse)
    b_src, b_w = k_rep[order], P.data[pidx][order]
    del k_rep, pidx, order
    # F_c = P^T B: entry (I, J) collects P[i, I] * B[i, J]
    e_rep, pidx = _expand_by_rows(P, rows_b)
    if e_rep.size > min(MAX_GALERKIN_PAIRS, MAX_GALERKIN_PAIRS_NUMPY):
        raise ValueError("Galerkin plan of %d pairs exceeds the limit; use "
                         "-pc_mg_galerkin none" % e_rep.size)
    rows_c, cols_c, c_ptr, order = _group(
        P.indices[pidx].astype(np.int64), cols_b[e_rep], ncoarse, ncoarse)
    c_src, c_w = e_rep[order], P.data[pidx][order]
    indptr = np.zeros(ncoarse + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows_c, minlength=ncoarse), out=indptr[1:])
    assert b_src.size < 2 ** 31 and cols_c.size < 2 ** 31
    return (b_ptr, b_src.astype(np.int32), b_w, c_ptr, c_src.astype(np.int32),
            c_w, indptr.astype(np.int32), cols_c.astype(np.int32))


class DeviceProducer(object):
    """Plans for ``problem`` (a :class:`fenapack_amd.fem.FlowProblem`) and the
    engine behind ``ksp`` (a set-up :class:`PCDKSP`)."""

    def __init__(self, problem, ksp):
        pb, V = problem, problem.space
        self.newton = pb.nls == "newton"
        self.pb, self.V, self.ksp = pb, V, ksp
        # 0: no communicator attached; >= 1: the engine's rows are partitioned
        # (1 = a one-rank RCCL communicator, PCD_FORCE_COMM=1)
        self.ranks = int(ksp.engine.info(c.INFO_RANKS))
        self.eng = eng = ksp.engine
        if not eng.L.hip:
            raise c.EngineError("device producer needs the HIP engine")
        d = V.dim
        ksp0, ksp1 = ksp.pc.getFieldSplitSubKSP()
        self.ksp0 = ksp0
        self.mg = ksp0.pc.type == "mg"
        if not self.mg:
            raise ValueError("device producer: the velocity solve must be "
                             "-fieldsplit_u_pc_type mg (its smoother bounds "
                             "are re-estimated on the device)")
        # -pc_type gamg: the aggregation - hence every prolongator and every
        # coarse PATTERN - is fixed across the nonlinear iterations (built
        # once from the first operator, fenapack_amd/amg.py); only the values
        # of the Galerkin products change.  The finest level is assembled from
        # the mesh like any other, the levels below it are the weighted-gather
        # products of pcd_fe_set_level_galerkin with the smoothed-aggregation
        # prolongators in place of the nested meshes' ones: what the reference
        # re-does through hypre's set-up every outer iteration
        # (demo_navier-stokes-pcd.py:153-160) happens in HBM.
        self.algebraic = bool(ksp0.pc.mg_algebraic)
        if self.algebraic and self.newton:
            # --nls newton on an algebraic hierarchy (the reference's bench
            # sweeps nls x ls: test/bench/test_pcd_scaling.py:194-223): the
            # chain must prolongate every component alike, P = P_s (x) I_d
            # (amg.smoothed_aggregation_chain coupled="scalar": aggregates of
            # the mean diagonal block) - then P^T (F (x) I + N) P is the scalar
            # triple product of every block, as on nested meshes
            from .petsc import _scalar_of
            for P in ksp0.pc.mg_data["chain"][1:]:
                f = _scalar_of(P)
                if f is None or f[1] != V.dim:
                    raise ValueError(
                        "device producer: --nls newton on an algebraic "
                        "hierarchy needs -fieldsplit_u_pc_gamg_coupled scalar "
                        "(P = P_s (x) I_d)")
        # (several ranks with the GLOBAL hand-over: the aggregation runs on
        # every rank alike - amg.smoothed_aggregation_chain is deterministic -
        # so the prolongators and the coarse patterns are replicated like the
        # nested meshes' ones and the Galerkin plans below stay replicated;
        # the engine takes its rows of every level.  The PARTITIONED host
        # producer's hierarchy (amg.PartitionedSA: aggregates per rank, levels
        # cut where they fall) has no device producer yet.)
        if self.algebraic and self.ranks and \
                getattr(ksp0.pc, "_mg_psa", None) is not None:
            raise ValueError("device producer: the algebraic hierarchy of "
                             "the partitioned host producer "
                             "(amg.PartitionedSA) is refreshed on the host")
        self.galerkin = bool(ksp0.pc.mg_galerkin) or self.algebraic
        self.supg = bool(pb.stabilize)
        if self.supg and self.galerkin:
            raise ValueError("device producer: the SUPG preconditioner "
                             "matrix goes with -fieldsplit_u_pc_mg_galerkin "
                             "none (every level re-discretised with its own "
                             "stabilisation parameter)")
        nlev = len(ksp0.pc.mg_data["ops"]) if self.mg else 1
        self.nlev = nlev
        # Several ranks: plans CUT BY ROWS (pcd_fe_set_rows) - every rank hands
        # the engine the cells, contribution lists and constants of the node
        # rows it owns of each partitioned level only, so that the element
        # work and the plan memory on a GPU are 1 / R of the level's (vectors
        # stay replicated).  Picard block, re-discretised coarse levels;
        # FENAPACK_AMD_FE_ROWS=0 keeps the replicated plans.
        self.rows = bool(self.ranks) and not self.newton and not self.galerkin \
            and os.environ.get("FENAPACK_AMD_FE_ROWS", "1") != "0"
        self._cut = [None] * nlev      # per level: (a0, a1, e0, e1) or None
        self.plan_entries = [0] * nlev  # what this rank holds (cells, entries)
        self.plan_terms = [0] * nlev    # per-term lists of a Galerkin level ("plans" mode)
        self.refresh_bytes = [0] * nlev  # HBM a Galerkin level's refresh holds
        top_h = len(pb.hierarchy.meshes) - 1
        self.levels = []                       # host problems, coarsest first
        for l in range(nlev):
            lh = top_h - (nlev - 1) + l
            if l == nlev - 1:
                self.levels.append(pb)
            elif self.galerkin:
                self.levels.append(None)       # no mesh data on these levels
            else:
                if not hasattr(pb, "_coarse_problems"):
                    pb._coarse_problems = {}
                if lh not in pb._coarse_problems:
                    pb._coarse_problems[lh] = pb._same_problem_on_level(lh)
                self.levels.append(pb._coarse_problems[lh])
        dphi = self._dphi(V)
        qw = V.wq[0] / V.area[0]
        # seconds of this set-up by phase (bench.py's picard_step reports it)
        self.init_timing = {}
        _t = [time.perf_counter()]

        def lap(what):
            now = time.perf_counter()
            self.init_timing[what] = self.init_timing.get(what, 0.0) + now - _t[0]
            _t[0] = now
        eng.fe_begin(d, nlev, qw, V.phi, dphi, V.psi)
        chain = pb.interpolations().velocity \
            if nlev > 1 and not self.algebraic else None
        self._pat = [None] * nlev       # (indptr, indices, n) of scalar F_l
        for l in range(nlev - 1, -1, -1):
            pl = self.levels[l]
            lh = top_h - (nlev - 1) + l
            if l == nlev - 1:
                self._set_level(l, pl, None)
                lap("finest level: contribution lists, constants")
            elif self.algebraic:
                # (mg_data["chain"][l + 1]: level l -> l + 1 of the hierarchy
                # the engine holds, after the choice of the coarsest level)
                self._set_level_galerkin(l, ksp0.pc.mg_data["chain"][l + 1])
                lap("Galerkin levels: symbolic products, hand-over")
            elif self.galerkin:
                self._set_level_galerkin(l, chain[lh + 1])
                lap("Galerkin levels: symbolic products, hand-over")
            else:
                self._set_level(l, pl, injection_map(chain[lh + 1], d))
                lap("re-discretised coarse levels")
        if self.newton:
            self._set_newton(ksp0)
        if self.ranks:
            # replicated producer, partitioned solve: the engine needs the
            # layout of every level's global F x I_d (pcd_fe_bind_pattern)
            for l in range(nlev):
                indptr, indices, _ = self._pat[l]
                if self._cut[l] is not None:
                    a0, a1, e0, e1 = self._cut[l]
                    # (GLOBAL entry offsets of the owned rows, their columns)
                    eng.fe_bind_pattern(l, indptr[a0:a1 + 1], indices[e0:e1])
                else:
                    eng.fe_bind_pattern(l, indptr, indices)
        lap("newton positions / patterns of the ranks")
        self._bind_system()
        lap("positions in the system values")
        self._bind_kp(ksp1)
        lap("Kp plan")
        a, b, cc, dd = ksp0.pc.mg_esteig
        eng.fe_bind_mg(c.KSP_A00, b, dd, 12)
        # the coarsest level's dense inverse: on the device as well, unless
        # asked otherwise (host LAPACK; kept as a cross-check)
        n0 = self._pat[0][2] * (V.dim if self.newton else 1)
        self.device_inverse = os.environ.get("PCD_FE_HOST_INVERSE") != "1" \
            and n0 <= 8192
        if self.newton and not self.device_inverse:
            raise ValueError("device producer: with the Newton linearisation "
                             "the coarsest level (%d rows) is inverted on the "
                             "device, limit 8192 rows" % n0)
        if self.newton:
            eng.fe_bind_coarse_inverse(self._K0.indptr, self._K0.indices)
        elif self.device_inverse:
            eng.fe_bind_coarse_inverse(self._pat[0][0], self._pat[0][1])
        # the residual and the Picard loop on the device as well - unless a
        # host-side coarse inverse needs the iterate on the host every
        # iteration
        self.device_loop = self.device_inverse \
            and os.environ.get("PCD_FE_HOST_LOOP") != "1"
        if self.ranks and not self.device_loop:
            raise ValueError("device producer: several ranks run the device "
                             "loop (coarsest level <= 8192 rows, inverted on "
                             "every rank)")
        if self.device_loop:
            mass = None
            if pb.idt:
                pat = V._patterns(False)["SS"]
                mass = pat.sum_entries(V.p2_mass_cells())
                cut = self._cut[nlev - 1]
                if cut is not None:
                    mass = mass[cut[2]:cut[3]]
            eng.fe_bind_residual(sp.csr_matrix(pb._A01_raw),
                                 sp.csr_matrix(pb._A10_raw), pb.bc_u_idx,
                                 pb._bc_mult[pb.bc_u_idx], mass, pb.idt)
        lap("coarse inverse, residual blocks")
        # constant host pieces of the residual
        self._bc_idx = pb.bc_u_idx
        self.timing = {"update": 0.0, "coarse_inverse": 0.0, "host": 0.0}

    # ------------------------------------------------------------------ plans
    @staticmethod
    def _dphi(V):
        from .fem.taylor_hood import _p2_basis
        _, dphi = _p2_basis(V.psi, V.local_edges)
        return dphi

    def _set_level(self, l, pl, inject):
        V, d = pl.space, pl.space.dim
        nc, na = V.mesh.num_cells, V.na
        pat = V._patterns(False)["SS"]
        ptr, src = _contribution_plan(pat.inv, nc, na * na, pat.nnz, pat)
        S0 = pl.nu * V.p2_stiffness_cells()
        if pl.idt:
            S0 = S0 + pl.idt * V.p2_mass_cells()
        f_const = pat.sum_entries(S0)
        nodes = np.unique(pl.bc_u_idx // d)
        assert nodes.size * d == pl.bc_u_idx.size, \
            "device producer: Dirichlet data must constrain whole nodes"
        flag = np.zeros(V.nn, dtype=bool)
        flag[nodes] = True
        rows, cols = pat.rows, pat.indices
        keep = ~(flag[rows] | flag[cols])
        diag_pos = np.nonzero((rows == cols) & flag[rows])[0]
        diag_val = pl._bc_mult[d * rows[diag_pos]]
        g = V.gradlam                                  # (nc, d+1, d)
        dofs2, gl, area, cell_h = V.cell_dofs2, g.reshape(nc, -1), V.area, \
            (V.cell_h if self.supg else None)
        cut = self._rows_of_level(l, V)
        if cut is not None:
            # this rank's node rows [a0, a1): their entries [e0, e1) of the
            # pattern, the cells that touch them (renumbered 0 .. in the
            # element storage), everything else of the level left out
            a0, a1 = cut
            e0, e1 = int(pat.indptr[a0]), int(pat.indptr[a1])
            own = np.zeros(V.nn, dtype=bool)
            own[a0:a1] = True
            sel_c = own[dofs2].any(axis=1)
            if l == self.nlev - 1:
                # the finest level's cells also serve Kp: the cells that touch
                # this rank's PRESSURE rows as well (the pressure cut falls
                # elsewhere than the velocity cut)
                q0, q1 = self.eng.row_range(V.n_p, velocity=False)
                ownp = np.zeros(V.n_p, dtype=bool)
                ownp[q0:q1] = True
                sel_c |= ownp[V.cell_dofs1].any(axis=1)
                self._p_rows = (q0, q1)
            cells = np.nonzero(sel_c)[0]
            newid = np.full(nc, -1, dtype=np.int64)
            newid[cells] = np.arange(cells.size)
            if l == self.nlev - 1:
                self._top_cells, self._top_newid = cells, newid
            p0, p1 = int(ptr[e0]), int(ptr[e1])
            ab, cell = np.divmod(src[p0:p1].astype(np.int64), nc)
            assert newid[cell].min(initial=0) >= 0
            src = (ab * cells.size + newid[cell]).astype(np.int32)
            ptr = (ptr[e0:e1 + 1] - ptr[e0]).astype(np.int32)
            f_const, keep = f_const[e0:e1], keep[e0:e1]
            sel = (diag_pos >= e0) & (diag_pos < e1)
            diag_pos, diag_val = diag_pos[sel] - e0, diag_val[sel]
            dofs2, gl, area = dofs2[cells], gl[cells], area[cells]
            if cell_h is not None:
                cell_h = cell_h[cells]
            self._cut[l] = (a0, a1, e0, e1)
        self.plan_entries[l] = (int(dofs2.shape[0]), int(ptr.size - 1))
        self.eng.fe_set_level(
            l, dofs2.T, gl.T, area, ptr, src, f_const,
            keep, diag_pos, diag_val, inject, V.nn)
        if cut is not None:
            self.eng.fe_set_rows(l, a0, a1 - a0)
        self._pat[l] = (pat.indptr, pat.indices, V.nn)
        if self.supg:
            from .fem.taylor_hood import _p2_basis
            lam = np.full((1, V.nvl), 1.0 / V.nvl)
            phi_mid, _ = _p2_basis(lam, V.local_edges)
            self.eng.fe_set_supg(l, cell_h, pl.nu, phi_mid[0], V.qw_s,
                                 V.phi_s, V.dphi_s)

    def _rows_of_level(self, l, V):
        """Node rows ``(a0, a1)`` of level ``l`` this rank's plan is cut to, or
        ``None`` (one rank, replicated plans, or a level the engine replicates:
        at most PCD_REPLICATE_BELOW rows, and the coarsest one)."""
        if not self.rows or l == 0:
            return None
        limit = int(os.environ.get("PCD_REPLICATE_BELOW", "60000"))
        if l < self.nlev - 1 and V.n_u <= limit:
            return None
        r0, r1 = self.eng.row_range(V.n_u, velocity=True)
        d = V.dim
        assert r0 % d == 0 and r1 % d == 0
        return r0 // d, r1 // d

    def _set_level_galerkin(self, l, P):
        """Level ``l`` = P^T (level l+1) P with the scalar part of the
        velocity prolongation ``P`` (= P_s x I_d)."""
        from . import _host
        d = self.V.dim
        Ps = sp.csr_matrix(P)[::d, ::d]
        Ps.sort_indices()
        indptr_f, indices_f, n_f = self._pat[l + 1]
        # The product itself: a numeric sparse product on fixed patterns
        # (k_spgemm_fixed; the symbolic phase here, once) - or, on request
        # (FENAPACK_AMD_GALERKIN=plans: the route of rounds 3-5, kept as the
        # cross-check of the kernel), two weighted gathers over per-TERM lists
        self.galerkin_mode = os.environ.get("FENAPACK_AMD_GALERKIN", "product")
        if self.galerkin_mode not in ("product", "plans"):
            raise ValueError("FENAPACK_AMD_GALERKIN = product | plans")
        if self.galerkin_mode == "plans" or _host.use_numpy():
            rows_f = np.repeat(np.arange(n_f), np.diff(indptr_f))
            plan = galerkin_plan(rows_f, indices_f, Ps)
            indptr_c, indices_c = plan[6], plan[7]
        else:
            plan = None
            PT = _host.transpose(Ps)
            b_ip, b_ix = _host.product_pattern(indptr_f, indices_f, Ps.indptr,
                                               Ps.indices, Ps.shape[1])
            indptr_c, indices_c = _host.product_pattern(
                PT.indptr, PT.indices, b_ip, b_ix, Ps.shape[1])
        # scipy's SpGEMM drops entries that cancel to an exact zero, so the
        # host's Galerkin pattern depends on the values it was built from;
        # the device product needs the structural pattern: re-create the
        # engine's level with it (values arrive with the first update)
        if 0 < l < self.nlev - 1 and not self.newton:
            ones = sp.csr_matrix((np.ones(indices_c.size), indices_c,
                                  indptr_c), shape=(Ps.shape[1],) * 2)
            K = sp.kron(ones, sp.identity(d), format="csr")
            K.sort_indices()
            mg = self.ksp0.pc.mg_data
            self.eng.mg_set_level(c.KSP_A00, l, K, mg["chain"][l],
                                  *mg["bounds"][l])
        if plan is not None:
            self.eng.fe_set_level_galerkin(l, *plan[:6])
            self.plan_terms[l] = int(plan[1].size + plan[4].size)
        else:
            self.eng.fe_set_level_product(l, Ps, PT, indptr_f, indices_f,
                                          b_ip, b_ix, indptr_c, indices_c)
        # bytes held in HBM for this level's refresh
        self.refresh_bytes[l] = (
            8 * (plan[0].size + plan[3].size) + 12 * self.plan_terms[l]
            if plan is not None else
            2 * (12 * Ps.nnz + 4 * (sum(Ps.shape) + 2))
            + 4 * (len(indices_f) + b_ix.size + indices_c.size)
            + 4 * (2 * n_f + Ps.shape[1] + 3))
        self._pat[l] = (indptr_c, indices_c, Ps.shape[1])

    def _set_newton(self, ksp0):
        """Positions of the ``d x d`` blocks in every level's operator.  The
        engine's intermediate levels are re-created on the structural pattern
        ``(pattern of F_l) x ones(d, d)`` (host products drop entries that are
        numerically zero at the first iterate); level 0 is the CSR the device
        inverts; the finest level is the engine's A00."""
        d, eng, mg = self.V.dim, self.eng, ksp0.pc.mg_data
        top = self.nlev - 1
        for l in range(self.nlev):
            indptr, indices, n = self._pat[l]
            if l == top:
                patA = self.V._patterns(True)["A00"]
                K = patA.matrix(np.ones(patA.nnz))
            else:
                K = coupled_pattern(indptr, indices, n, d)
                if l > 0:
                    eng.mg_set_level(c.KSP_A00, l, K, mg["chain"][l],
                                     *mg["bounds"][l])
            if l == 0:
                self._K0 = K     # (a one-level hierarchy: the same pattern)
                # the coupled coarsest operator is inverted WHOLE on the
                # device: level 0 holds the full (d n0)^2 inverse (the host's
                # first hierarchy, built at w = 0 where the block is still
                # F (x) I_d, handed over inv(F) once: re-create it)
                N0 = n * d
                eng.mg_set_level(c.KSP_A00, 0, dense_csr(np.zeros((N0, N0))))
            eng.fe_set_newton(l, block_positions(K, indptr, indices, d))

    def _bind_system(self):
        # Where entry (k, component[s]) of the finest scalar F sits in the
        # values of the monolithic matrix the engine was handed: a bisection
        # per entry in that matrix's own rows (libpcd_host pcdh_locate).
        # (Rounds 3-5 went through the inverse of the assembly permutation -
        # an argsort of config 5's 1.1e9 entries: 10.6 of the 14.3 s of this
        # set-up, and 9 GB of host memory per index array.)
        from . import _host
        V, d = self.V, self.V.dim
        patS = V._patterns(False)["SS"]
        M = self.ksp.getOperators()[0].A
        cut = self._cut[self.nlev - 1]
        e0, e1 = (cut[2], cut[3]) if cut is not None else (0, patS.nnz)
        pairs = [(i, j) for i in range(d) for j in range(d)] if self.newton \
            else [(k, k) for k in range(d)]
        pos = _host.locate_blocks(M, patS.rows[e0:e1], patS.indices[e0:e1], d,
                                  pairs, V.is_u)
        self.eng.fe_bind_system(pos)

    def _bind_kp(self, ksp1):
        pb, V = self.pb, self.V
        nvl, nc = V.nvl, V.mesh.num_cells
        pat = V._patterns(False)["PP"]
        ptr, src = _contribution_plan(pat.inv, nc, nvl * nvl, pat.nnz, pat)
        cst = None
        idt = 0.0 if pb.pcdr else pb.idt
        if idt:
            M = V.area[:, None, None] * V._ref()["P"][None] * (idt / pb.nu)
            cst = pat.sum_entries(M)
        cutp = None
        if self._cut[self.nlev - 1] is not None:
            # this rank's pressure rows of Kp, sources renumbered to the
            # finest level's local cells
            q0, q1 = self._p_rows
            e0, e1 = int(pat.indptr[q0]), int(pat.indptr[q1])
            p0, p1 = int(ptr[e0]), int(ptr[e1])
            ab, cell = np.divmod(src[p0:p1].astype(np.int64), nc)
            assert self._top_newid[cell].min(initial=0) >= 0
            src = (ab * self._top_cells.size
                   + self._top_newid[cell]).astype(np.int32)
            ptr = (ptr[e0:e1 + 1] - ptr[e0]).astype(np.int32)
            if cst is not None:
                cst = cst[e0:e1]
            cutp = (e0, e1)
        self.eng.fe_bind_kp(ptr, src, cst, 1.0 / pb.nu)
        if cutp is not None:
            self.eng.fe_set_kp_rows(cutp[0], pat.nnz)
        self._kp_cut = cutp
        self.nnz_kp = pat.nnz
        # BRM2: - (1/nu) int_inflow (w.n) p q ds depends on the iterate too:
        # a few boundary edges, local 2 x 2 matrices, gathered into the
        # entries of Kp they touch (pcd_fe_bind_robin)
        if pb.variant == "BRM2" and len(pb.robin_edges) > 0:
            pl = V.robin_plan(pb.robin_edges)
            nb = pl["length"].size
            pd = pl["pdofs"]
            k = pd.shape[1]             # pressure dofs per facet: 2 (edge), 3 (face)
            rows = np.repeat(pd[:, :, None], k, axis=2).ravel()
            cols = np.repeat(pd[:, None, :], k, axis=1).ravel()
            where = pat.locate(rows, cols)           # (e, i, j) -> Kp entry
            e_idx, ij = np.divmod(np.arange(where.size), k * k)
            if cutp is not None:
                # (rows of other ranks: their owners add the term)
                mine = (where >= cutp[0]) & (where < cutp[1])
                where, e_idx, ij = where[mine], e_idx[mine], ij[mine]
            aff_pos, _, aff_ptr, order = _group(
                where, np.zeros_like(where), pat.nnz, 1)
            self.eng.fe_bind_robin(pl["nodes"].T, pl["normal"].T,
                                   pl["length"], aff_pos, aff_ptr,
                                   (ij * nb + e_idx)[order],
                                   np.full(where.size, -1.0 / pb.nu))

    # ------------------------------------------------------------ device loop
    def set_time_level(self):
        """Boundary values at ``problem.t`` and the previous velocity for the
        device residual."""
        pb = self.pb
        self.eng.fe_set_bc_values(pb.bc_u_values(pb.t))
        if pb.idt:
            self.eng.fe_set_previous(pb.u0)

    def residual(self, x):
        """Device residual at the mixed vector ``x`` (operators refreshed)."""
        b = np.empty_like(x)
        self.eng.fe_residual(np.ascontiguousarray(x), b)
        return b

    # ----------------------------------------------------------------- update
    def update(self, xu, xp):
        """Refresh every iterate-dependent operator of the engine at
        ``(xu, xp)`` and return the nonlinear residual in the mixed
        numbering (same definition as ``FlowProblem.linearise``)."""
        pb, V = self.pb, self.V
        t0 = time.perf_counter()
        g = pb.bc_u_values(pb.t)
        dd = np.zeros(V.n_u)
        dd[self._bc_idx] = xu[self._bc_idx] - g
        v = xu - dd
        ru = np.empty(V.n_u)
        t1 = time.perf_counter()
        self.eng.fe_update(np.ascontiguousarray(xu), v, ru)
        t2 = time.perf_counter()
        if not self.device_inverse:
            self._refresh_coarsest()
        t3 = time.perf_counter()
        # (Newton: ru already carries - N_unc d, see pcd_fe_update)
        Fu = ru + pb._A01_raw @ xp
        if pb.idt:
            Fu -= pb.idt * (pb._Mmass @ pb.u0)
        Fp = pb._A10_raw @ v
        Fu[self._bc_idx] = pb._bc_mult[self._bc_idx] * dd[self._bc_idx]
        b = V.to_mixed(Fu, Fp)
        t4 = time.perf_counter()
        self.timing["update"] += t2 - t1
        self.timing["coarse_inverse"] += t3 - t2
        self.timing["host"] += (t1 - t0) + (t4 - t3)
        return b

    def _refresh_coarsest(self):
        """inv(F x I) = inv(F) x I: invert the scalar coarsest operator on the
        host and hand the dense inverse to the multigrid's level 0."""
        d = self.V.dim
        F0 = self._scalar(0).toarray()
        Finv = np.linalg.inv(F0)
        # (the layout fem.multigrid.coarse_inverse hands over: inv(F) (x) I_d
        # without the entries that couple no components)
        from . import _host
        C = _host.kron_expand(dense_csr(Finv), d)
        self.eng.mg_update_values(c.KSP_A00, 0, C.data)

    def _scalar(self, l):
        indptr, indices, n = self._pat[l]
        if self._cut[l] is not None:
            # this rank's rows only (global shape, the other rows empty)
            a0, a1, e0, e1 = self._cut[l]
            vals = self.eng.fe_level_values(l, e1 - e0)
            ip = np.zeros(n + 1, dtype=np.int64)
            ip[a0 + 1:a1 + 1] = np.asarray(indptr[a0 + 1:a1 + 1]) - e0
            ip[a1 + 1:] = e1 - e0
            return sp.csr_matrix((vals, indices[e0:e1], ip), shape=(n, n))
        vals = self.eng.fe_level_values(l, indices.size)
        return sp.csr_matrix((vals, indices, indptr), shape=(n, n))

    def level_matrix(self, l):
        """Velocity operator of FE level ``l`` as assembled on the device
        (scipy CSR, F x I_d [+ N]) - for tests and diagnostics."""
        d = self.V.dim
        F = self._scalar(l)
        K = sp.kron(F, sp.identity(d), format="csr")
        if self.newton:
            indptr, indices, n = self._pat[l]
            Nv = self.eng.fe_newton_values(l, indices.size, d)
            for i in range(d):
                for j in range(d):
                    E = np.zeros((d, d))
                    E[i, j] = 1.0
                    Nij = sp.csr_matrix((Nv[i * d + j], indices, indptr),
                                        shape=(n, n))
                    K = K + sp.kron(Nij, E, format="csr")
            K = sp.csr_matrix(K)
        K.sort_indices()
        return K

    def kp_matrix(self):
        pat = self.V._patterns(False)["PP"]
        cut = getattr(self, "_kp_cut", None)
        if cut is not None:                  # this rank's pressure rows only
            vals = np.zeros(pat.nnz)
            vals[cut[0]:cut[1]] = self.eng.fe_kp_values(cut[1] - cut[0])
            return pat.matrix(vals)
        return pat.matrix(self.eng.fe_kp_values(pat.nnz))


class DevicePicardSolver(object):
    """Picard (or, with ``problem.nls == "newton"``, Newton) iteration with
    the device producer.  The first nonlinear step of
    the first solve runs through the reference-shaped stack
    (``PCDNewtonSolver`` -> ``init_pcd`` -> host producer) and sets everything
    up; from then on operators never leave HBM.  Stopping rules, relaxation and
    bookkeeping are those of :class:`fenapack_amd.PCDNewtonSolver`."""

    def __init__(self, problem, **kw):
        from .driver import make_solver
        self.max_newton = kw.pop("max_newton", 25)
        self.problem = problem
        self.w, self.nls, self.nlp = make_solver(problem, max_newton=1, **kw)
        self.nls.parameters["error_on_nonconvergence"] = False
        self.producer = None
        self.time_plan = 0.0
        self.time_gmres = 0.0
        self.time_device_loop = 0.0
        self.krylov_history, self.residual_history = [], []

    def krylov_iterations(self):
        return int(sum(self.krylov_history))

    def solve(self):
        """One nonlinear solve from the current ``w``; returns
        ``(iterations, converged)``."""
        pb, V, w, nls = self.problem, self.problem.space, self.w, self.nls
        prm = nls.parameters
        x = w.vector()
        if self.producer is None:
            # (the residual at the new iterate is evaluated by the producer)
            it, converged = nls.solve(self.nlp, x, on_update=w.touch,
                                      final_residual=False)
            self.krylov_history = list(nls.krylov_history)
            self.residual_history = list(nls.residual_history)
            r0 = self.residual_history[0]
            if len(self.residual_history) == it:       # final one skipped
                self.residual_history.append(float("nan"))
            t0 = time.time()
            # (a partitioned problem gets the rank-local producer)
            from .device_producer_rows import make_device_producer
            self.producer = make_device_producer(pb,
                                                 nls.linear_solver().ksp())
            self.time_plan = time.time() - t0
            # (no hipGraph replay here: every update changes the smoother
            # bounds baked into the captured launches, and re-capturing costs
            # ~30 ms - more than the 31 eager applies of one solve)
            first = True
        else:
            it, r0, first, converged = 0, 0.0, False, False
            self.krylov_history, self.residual_history = [], []
        solver = nls.linear_solver()
        ksp = solver.ksp()
        if self.producer.device_loop:
            # the rest of the iteration is one call: residual, GMRES and
            # update stay on the device (pcd_fe_picard_solve)
            t1 = time.perf_counter()
            self.producer.set_time_level()
            k, converged, lin, res = ksp.engine.fe_picard_solve(
                x, c.MEM_HOST, r0, prm["relative_tolerance"],
                prm["absolute_tolerance"], max(self.max_newton - it, 0),
                prm["relaxation_parameter"],
                solver.parameters["relative_tolerance"],
                solver.parameters["absolute_tolerance"], ksp.restart,
                solver.parameters["maximum_iterations"])
            self.time_device_loop += time.perf_counter() - t1
            self.krylov_history += lin
            if first:
                self.residual_history[-1] = res[0]
                self.residual_history += res[1:]
            else:
                self.residual_history = res
            w.touch()
            return it + k, converged
        b = self.producer.update(x[V.is_u], x[V.is_p])
        r = float(np.linalg.norm(b))
        if first:
            self.residual_history[-1] = r
        else:
            r0 = r
            self.residual_history = [r0]
            converged = r0 < prm["absolute_tolerance"]
        dx = np.zeros_like(x)
        while not converged and it < self.max_newton:
            dx[:] = 0.0
            t1 = time.perf_counter()
            its, _ = ksp.engine.gmres_solve(
                b, dx, c.MEM_HOST, solver.parameters["relative_tolerance"],
                solver.parameters["absolute_tolerance"], ksp.restart,
                solver.parameters["maximum_iterations"])
            self.time_gmres += time.perf_counter() - t1
            self.krylov_history.append(its)
            x -= prm["relaxation_parameter"] * dx
            w.touch()
            it += 1
            b = self.producer.update(x[V.is_u], x[V.is_p])
            r = float(np.linalg.norm(b))
            self.residual_history.append(r)
            converged = (r < prm["absolute_tolerance"]
                         or r / r0 < prm["relative_tolerance"])
        return it, converged


#: the loop is the same for both linearisations
DeviceNonlinearSolver = DevicePicardSolver


def solve_steady_device(problem, **kw):
    """Steady solve; same stats dict as
    :func:`fenapack_amd.driver.solve_steady` plus a time breakdown."""
    t0 = time.time()
    s = DevicePicardSolver(problem, **kw)
    it, converged = s.solve()
    t_total = time.time() - t0
    return {"w": s.w, "newton_its": it, "converged": converged,
            "krylov_its": s.krylov_iterations(),
            "krylov_per_step": list(s.krylov_history),
            "residuals": list(s.residual_history), "time": t_total,
            "time_plan": s.time_plan, "time_gmres": s.time_gmres,
            "time_device_loop": s.time_device_loop,
            "producer_timing": dict(s.producer.timing),
            "solver": s.nls, "producer": s.producer}


def solve_unsteady_device(problem, dt, t_end, **kw):
    """Backward-Euler loop of the unsteady demo (:188-208) with the device
    producer; same stats dict as :func:`fenapack_amd.driver.solve_unsteady`."""
    s = DevicePicardSolver(problem, **kw)
    V, w = problem.space, s.w
    t, steps, krylov, newton = 0.0, 0, 0, 0
    per_step, newton_per_step, residuals = [], [], []
    t0 = time.time()
    while t < t_end - 0.1 * dt:
        t += dt
        steps += 1
        problem.t = t                           # inflow.t = t
        w.touch()
        n_it, _ = s.solve()
        krylov += s.krylov_iterations()
        newton += n_it
        per_step.append(s.krylov_iterations())
        newton_per_step.append(list(s.krylov_history))
        residuals.append(list(s.residual_history))
        problem.u0 = w.split()[0].copy()        # w0.assign(w)
        w.touch()
    return {"w": w, "steps": steps, "krylov_its": krylov,
            "krylov_per_step": per_step, "newton_its": newton,
            "krylov_per_newton": newton_per_step, "residuals": residuals,
            "time": time.time() - t0, "ndof": V.ndof,
            "producer_timing": dict(s.producer.timing),
            "producer": s.producer,
            "time_gmres": s.time_gmres,
            "time_device_loop": s.time_device_loop}
