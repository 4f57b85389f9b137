"""Device operator producer of a PARTITIONED problem: every rank feeds the
engine from its own slab.

In the reference every rank re-assembles ``fp`` / ``kp`` for ITS rows each outer
iteration (``fenapack/assembling.py:98-106``;
``fenapack/field_split_backend.py:79-83, 285-291``; owned rows only:
``fenapack/SubfieldBC.h:136-155``) and re-runs the AMG set-up on them
(``demo_navier-stokes-pcd.py:153-160``).  :class:`DeviceProducer` does that in
HBM for one GPU and - with ranks - from a WHOLE host problem replicated on
every rank (its plans cut by rows).  This class is the rank-local form: the
host side is ``fem/partition.py``'s slab (the cells that touch this rank's
rows: 1 / R of the problem), the hand-over is the rank-local one
(``pcd_set_system_local`` / ``pcd_mg_set_level_local``), and the algebraic
hierarchy is ``amg.PartitionedSA``'s - aggregates per rank, coarse levels cut
where they fall - refreshed ON THE DEVICE:

* finest level: element kernels + contribution lists of the slab's cells, node
  numbers of the plans mapped to the global ones (the winds stay whole:
  vectors of the nonlinear side are replicated, operators are not);
* every coarse level: the numeric sparse products ``B = F_rows P_ext`` and
  ``T = P_own^T B`` by rows (``k_spgemm_fixed``); a rank's terms of other ranks'
  coarse rows travel over one all-reduced wire buffer and are added in rank
  order (``pcd_fe_set_level_product_rows`` - what ``HostComm.sum_rows`` does on
  the host); the first level below the replication limit is gathered the
  same way and the levels under it are replicated products;
* smoother bounds (power iteration with all-reduced norms), the coarsest
  level's inverse, ``K_p``, the residual (by rows, completed by all-reduces)
  and the Picard loop stay on the device as on one GPU.

Nothing here is on the hot path of a PCApply; it is what makes a nonlinear
step of a partitioned run cost a solve instead of a host refresh.
"""
import os

import numpy as np
import scipy.sparse as sp

from . import _cabi as c
from . import _host
from .device_producer import DeviceProducer, _contribution_plan

__all__ = ["PartitionedDeviceProducer", "make_device_producer"]


def make_device_producer(problem, ksp):
    """The producer that goes with ``problem``: the rank-local one for a
    ``fem.partition.PartitionedProblem``, :class:`DeviceProducer` otherwise."""
    if getattr(problem, "partitioned", False):
        return PartitionedDeviceProducer(problem, ksp)
    return DeviceProducer(problem, ksp)


def _union_rows(pieces, nrows, ncols):
    """Union pattern of CSR row blocks ``(indptr, indices)`` over the same
    ``nrows`` rows, and for every piece the position of its entries in it."""
    keys = []
    for ip, ix in pieces:
        rows = np.repeat(np.arange(nrows, dtype=np.int64), np.diff(ip))
        keys.append(rows * ncols + np.asarray(ix, dtype=np.int64))
    allk = np.unique(np.concatenate(keys)) if keys else np.zeros(0, np.int64)
    indptr = np.zeros(nrows + 1, dtype=np.int64)
    np.cumsum(np.bincount(allk // ncols, minlength=nrows), out=indptr[1:])
    pos = [np.searchsorted(allk, k).astype(np.int32) for k in keys]
    return indptr.astype(np.int32), (allk % ncols).astype(np.int32), pos


class PartitionedDeviceProducer(DeviceProducer):
    """Plans for a ``PartitionedProblem`` and the engine behind ``ksp`` (a
    set-up :class:`PCDKSP` with the rank-local hand-over)."""

    def __init__(self, problem, ksp):
        pp, V = problem, problem.space
        if pp.nls != "picard":
            raise ValueError("partitioned device producer: the Picard block "
                             "(--nls picard)")
        if pp.stabilize or pp.coarse_stabilize:
            raise ValueError("partitioned device producer: no SUPG (its "
                             "hierarchy is re-discretised, not Galerkin)")
        self.newton, self.supg = False, False
        self.pb, self.V, self.ksp = pp, V, ksp
        self.eng = eng = ksp.engine
        self.ranks = int(eng.info(c.INFO_RANKS))
        if not eng.L.hip or not self.ranks or not eng.local_handover:
            raise c.EngineError("partitioned device producer: the HIP engine "
                                "with a communicator and the rank-local "
                                "hand-over (a PartitionedProblem's solver)")
        self.host, self.R, self.me = pp.host, pp.size, pp.rank
        lev = pp.fine
        self.sub, self.loc = lev.sub, lev.loc
        Vl, d = self.sub.V, V.dim
        ksp0, ksp1 = ksp.pc.getFieldSplitSubKSP()
        self.ksp0 = ksp0
        psa = getattr(ksp0.pc, "_mg_psa", None)
        if ksp0.pc.type != "mg" or not ksp0.pc.mg_algebraic or psa is None:
            raise ValueError("partitioned device producer: the velocity solve "
                             "must be -fieldsplit_u_pc_type gamg on the "
                             "partitioned producer (amg.PartitionedSA)")
        self.mg, self.algebraic, self.galerkin = True, True, True
        self.galerkin_mode = "product"
        mg = ksp0.pc.mg_data
        self.nlev = nlev = len(mg["ops"])
        self.psa, self.lvl_off = psa, psa.nlevels - nlev
        npart = len(psa.part)
        if nlev < npart + 1:
            raise ValueError("partitioned device producer: the engine's "
                             "hierarchy is cut above the gathered level")
        self.rows = True
        self._cut = [None] * nlev
        self._own = [None] * nlev       # per level: (ip_own, idx_global, n) of my rows
        self._pat = [None] * nlev
        self.plan_entries = [0] * nlev
        self.plan_terms = [0] * nlev
        self.refresh_bytes = [0] * nlev
        self.wire_doubles = [0] * nlev
        self.levels = [None] * (nlev - 1) + [self.loc]
        dphi = self._dphi(Vl)
        qw = Vl.wq[0] / Vl.area[0]
        eng.fe_begin(d, nlev, qw, Vl.phi, dphi, Vl.psi)
        top = nlev - 1
        self._set_top_level(top)
        slot = c.KSP_A00
        for k in range(npart):
            self._set_level_rows(top - 1 - k, k, mg, slot)
        for l in range(top - 1 - npart, -1, -1):
            # replicated tail: the ordinary product on every rank alike
            P = psa.tail[l + self.lvl_off + 1]
            self._set_level_product(l, sp.csr_matrix(P), mg)
        for l in range(nlev):
            ip, ix, _ = self._pat[l]
            eng.fe_bind_pattern(l, ip, ix)
        self._bind_system_rows()
        self._bind_kp_rows()
        a, b, cc, dd = ksp0.pc.mg_esteig
        eng.fe_bind_mg(c.KSP_A00, b, dd, 12)
        n0 = self._pat[0][2]
        self.device_inverse = n0 <= 8192
        if not self.device_inverse:
            raise ValueError("partitioned device producer: the coarsest level "
                             "(%d rows) is inverted on every rank, limit 8192"
                             % n0)
        eng.fe_bind_coarse_inverse(self._pat[0][0], self._pat[0][1])
        self.device_loop = True
        mass = None
        if pp.idt:
            a0, a1, e0, e1 = self._top_cut
            pat = Vl._patterns(False)["SS"]
            mass = pat.sum_entries(Vl.p2_mass_cells())[e0:e1]
        eng.fe_bind_residual(
            sp.csr_matrix(pp._up(lev, self.loc._A01_raw, "A01raw")),
            sp.csr_matrix(pp._pu(lev, self.loc._A10_raw, "A10raw")),
            pp.bc_u_idx, pp.G._bc_mult[pp.bc_u_idx], mass, pp.idt)
        eng.fe_set_residual_rows(True)
        self._bc_idx = pp.bc_u_idx
        self.timing = {"update": 0.0, "coarse_inverse": 0.0, "host": 0.0}

    # ------------------------------------------------------------ finest level
    def _set_top_level(self, l):
        """The finest level from the slab: the base class's row-cut plan with
        the local space in place of the global one - patterns, contribution
        lists and constants in the slab's numbering, the plan's NODE numbers
        (what indexes the replicated wind) mapped to the global ones."""
        pp, sub, loc = self.pb, self.sub, self.loc
        Vl, Vg, d = sub.V, self.V, self.V.dim
        lev = pp.fine
        nc, na = Vl.mesh.num_cells, Vl.na
        pat = Vl._patterns(False)["SS"]
        ptr, src = _contribution_plan(pat.inv, nc, na * na, pat.nnz, pat)
        S0 = loc.nu * Vl.p2_stiffness_cells()
        if loc.idt:
            S0 = S0 + loc.idt * Vl.p2_mass_cells()
        f_const = pat.sum_entries(S0)
        nodes = np.unique(loc.bc_u_idx // d)
        flag = np.zeros(Vl.nn, dtype=bool)
        flag[nodes] = True
        rows, cols = pat.rows, pat.indices
        keep = ~(flag[rows] | flag[cols])
        diag_pos = np.nonzero((rows == cols) & flag[rows])[0]
        diag_val = loc._bc_mult[d * rows[diag_pos]]
        # my node rows [a0, a1) in the slab's numbering, their entries
        a0, a1 = (int(v) for v in np.searchsorted(sub.nodes_g, lev.own_nodes))
        if a1 - a0 != lev.own_nodes[1] - lev.own_nodes[0]:
            raise ValueError("partitioned device producer: an owned node is "
                             "missing from the slab")
        e0, e1 = int(pat.indptr[a0]), int(pat.indptr[a1])
        p0, p1 = int(ptr[e0]), int(ptr[e1])
        # (every cell of the slab touches an owned velocity node or an owned
        # pressure dof - partition._Level.owned_cells - so all of them stay)
        self._top_cells = np.arange(nc)
        self._top_newid = np.arange(nc)
        src = src[p0:p1]
        ptr = (ptr[e0:e1 + 1] - ptr[e0]).astype(np.int32)
        f_const, keep = f_const[e0:e1], keep[e0:e1]
        sel = (diag_pos >= e0) & (diag_pos < e1)
        diag_pos, diag_val = diag_pos[sel] - e0, diag_val[sel]
        self._top_cut = (a0, a1, e0, e1)
        self.plan_entries[l] = (int(nc), int(ptr.size - 1))
        self.eng.fe_set_level(
            l, sub.nodes_g[Vl.cell_dofs2].T, Vl.gradlam.reshape(nc, -1).T,
            Vl.area, ptr, src, f_const, keep, diag_pos, diag_val, None, Vg.nn)
        self.eng.fe_set_rows(l, lev.own_nodes[0], a1 - a0)
        ip_own = (pat.indptr[a0:a1 + 1] - e0).astype(np.int32)
        idx_g = sub.nodes_g[pat.indices[e0:e1]].astype(np.int32)
        self._own[l] = (ip_own, idx_g, Vg.nn, lev.own_nodes[0])
        self._pat[l] = (ip_own, idx_g, Vg.nn)

    # ----------------------------------------------------- partitioned levels
    def _set_level_rows(self, l, k, mg, slot):
        """Engine level ``l`` = coarsening ``k`` of the partitioned hierarchy
        applied to level ``l + 1``: products by rows, the terms of other
        ranks' coarse rows over the wire, gathered when the level is the
        first replicated one."""
        eng, host, R, me, d = self.eng, self.host, self.R, self.me, self.V.dim
        psa = self.psa
        lev = psa.part[k]
        gathered = k == len(psa.part) - 1
        Pe = sp.csr_matrix(lev["Pext"] if psa.smooth == "global"
                           else lev["Pown"])
        Pe.sort_indices()
        Pg = sp.csr_matrix(lev["Pg"])
        Pg.sort_indices()
        PT = _host.transpose(Pg)                      # nc x n_own
        nc = int(lev["nc"])
        cuts = np.asarray(lev["cuts"], dtype=np.int64)
        c0, c1 = int(cuts[me]), int(cuts[me + 1])
        f_ip, f_ix, n_f, _ = self._own[l + 1]
        if Pe.shape[0] != n_f or Pg.shape[0] != f_ip.size - 1:
            raise ValueError("partitioned device producer: level %d of the "
                             "hierarchy does not match the producer's rows"
                             % (l + 1))
        b_ip, b_ix = _host.product_pattern(f_ip, f_ix, Pe.indptr, Pe.indices,
                                           nc)
        t_ip, t_ix = _host.product_pattern(PT.indptr, PT.indices, b_ip, b_ix,
                                           nc)
        # who sends what: the patterns of my terms of every other rank's rows
        t64 = t_ip.astype(np.int64)
        out = {}
        for q in range(R):
            if q == me:
                continue
            lo, hi = int(t64[cuts[q]]), int(t64[cuts[q + 1]])
            if hi > lo:
                out[q] = (t_ip[cuts[q]:cuts[q + 1] + 1] - lo, t_ix[lo:hi])
        got = host.allgather(out)
        lens = np.zeros((R, R), dtype=np.int64)
        for q in range(R):
            for r, (ip_, ix_) in got[q].items():
                lens[q, r] = ix_.size
        off = np.zeros((R, R), dtype=np.int64)
        off.ravel()[1:] = np.cumsum(lens.ravel())[:-1]
        wire = int(lens.sum())
        # my rows: the sources in RANK order (HostComm.sum_rows)
        pieces, src_of = [], []
        for q in range(R):
            if q == me:
                lo = int(t64[c0])
                pieces.append((t_ip[c0:c1 + 1] - lo, t_ix[lo:int(t64[c1])]))
                src_of.append((0, lo))
            elif me in got[q]:
                pieces.append(got[q][me])
                src_of.append((1, int(off[q, me])))
        o_ip, o_ix, pos = _union_rows(pieces, c1 - c0, nc)
        sends = [(int(t64[cuts[r]]), int(lens[me, r]), int(off[me, r]))
                 for r in range(R) if r != me and lens[me, r]]
        adds, poff = [], 0
        for (kind, so), p in zip(src_of, pos):
            adds.append((kind, so, p.size, poff))
            poff += p.size
        pos_all = np.concatenate(pos) if pos else np.zeros(0, np.int32)
        n_out = int(o_ix.size)
        goff = gtot = 0
        if gathered:
            sizes = host.allgather(n_out)
            goff, gtot = int(sum(sizes[:me])), int(sum(sizes))
        eng.fe_set_level_product_rows(
            l, Pe, PT, f_ip, f_ix, b_ip, b_ix, t_ip, t_ix, n_out, wire,
            sends, adds, pos_all, goff, gtot, c0, c1 - c0)
        self.wire_doubles[l] = wire
        self.refresh_bytes[l] = (12 * (Pe.nnz + PT.nnz) + 4 * (
            f_ix.size + b_ix.size + t_ix.size + pos_all.size)
            + 8 * (b_ix.size + t_ix.size + wire))
        chain = mg["chain"]
        bounds = mg["bounds"]
        if gathered:
            # the whole operator, rank after rank (coarse rows are numbered
            # rank by rank): replicated from here down
            parts = host.allgather((o_ip, o_ix))
            w_ix = np.concatenate([p[1] for p in parts]).astype(np.int32)
            w_ip = np.zeros(nc + 1, dtype=np.int64)
            np.cumsum(np.concatenate([np.diff(p[0]) for p in parts]),
                      out=w_ip[1:])
            w_ip = w_ip.astype(np.int32)
            self._pat[l] = (w_ip, w_ix, nc)
            self._own[l] = None
            if l > 0:
                ones = sp.csr_matrix((np.ones(w_ix.size), w_ix, w_ip),
                                     shape=(nc, nc))
                K = sp.kron(ones, sp.identity(d), format="csr")
                K.sort_indices()
                eng.mg_set_level(slot, l, K, chain[l], *bounds[l])
            return
        # a partitioned level: my rows (structural pattern) replace the
        # host-built ones of the engine's level
        self._pat[l] = (o_ip, o_ix, nc)
        self._own[l] = (o_ip, o_ix, nc, c0)
        ones = sp.csr_matrix((np.ones(o_ix.size), o_ix, o_ip),
                             shape=(c1 - c0, nc))
        K_rows = sp.kron(ones, sp.identity(d), format="csr")
        K_rows.sort_indices()
        P = sp.csr_matrix(chain[l])
        r0, r1 = c0 * d, c1 * d
        below_part = k + 1 < len(psa.part) - 1
        R_rows = None
        if below_part:
            cb = np.asarray(psa.part[k + 1]["cuts"], dtype=np.int64) * d
            R_rows = _host.transpose(P)[int(cb[me]):int(cb[me + 1])]
        eng.mg_set_level_local(slot, l, nc * d, K_rows, P[r0:r1], R_rows,
                               *bounds[l])

    def _set_level_product(self, l, Ps, mg):
        """A replicated level below the gathered one: the ordinary numeric
        product (``pcd_fe_set_level_product``), the same on every rank."""
        d = self.V.dim
        Ps = sp.csr_matrix(Ps)
        if Ps.shape[0] != self._pat[l + 1][2]:
            # (the chain of the solver stack holds P (x) I_d)
            Ps = Ps[::d, ::d]
        Ps.sort_indices()
        indptr_f, indices_f, n_f = self._pat[l + 1]
        PT = _host.transpose(Ps)
        b_ip, b_ix = _host.product_pattern(indptr_f, indices_f, Ps.indptr,
                                           Ps.indices, Ps.shape[1])
        c_ip, c_ix = _host.product_pattern(PT.indptr, PT.indices, b_ip, b_ix,
                                           Ps.shape[1])
        if l > 0:
            ones = sp.csr_matrix((np.ones(c_ix.size), c_ix, c_ip),
                                 shape=(Ps.shape[1],) * 2)
            K = sp.kron(ones, sp.identity(d), format="csr")
            K.sort_indices()
            self.eng.mg_set_level(c.KSP_A00, l, K, mg["chain"][l],
                                  *mg["bounds"][l])
        self.eng.fe_set_level_product(l, Ps, PT, indptr_f, indices_f, b_ip,
                                      b_ix, c_ip, c_ix)
        self.refresh_bytes[l] = 2 * 12 * Ps.nnz + 4 * (
            len(indices_f) + b_ix.size + c_ix.size)
        self._pat[l] = (c_ip, c_ix, Ps.shape[1])

    # ------------------------------------------------------- system, Kp
    def _bind_system_rows(self):
        """Where entry ``(k, comp)`` of my rows of the finest ``F`` sits in
        the system values this rank handed over (``pcd_set_system_local``:
        its velocity rows, then its pressure rows, of the monolithic matrix
        - each row's entries in the matrix's own column order)."""
        V, d, sub = self.V, self.V.dim, self.sub
        ksp = self.ksp
        A = ksp.getOperators()[0]
        rows, _ = ksp._local_rows(A, *ksp._is)
        M = A.A
        ip = M.indptr.astype(np.int64)
        ln = ip[rows + 1] - ip[rows]
        start = np.concatenate([[0], np.cumsum(ln)])       # local row -> first local entry
        a0, a1, e0, e1 = self._top_cut
        pat = sub.V._patterns(False)["SS"]
        n_own = a1 - a0
        g_rows = sub.nodes_g[pat.rows[e0:e1]].astype(np.int64)
        g_cols = sub.nodes_g[pat.indices[e0:e1]].astype(np.int64)
        u0 = d * self.pb.fine.own_nodes[0]                 # first owned velocity dof
        pos = np.empty((d, e1 - e0), dtype=np.int64)
        is_u = np.asarray(V.is_u, dtype=np.int64)
        for k in range(d):
            lrow = d * g_rows + k - u0                     # local system row
            if lrow.min(initial=0) < 0 or lrow.max(initial=0) >= d * n_own:
                raise ValueError("partitioned device producer: a row of the "
                                 "plan is not among the rows handed over")
            target = is_u[d * g_cols + k]
            # per row: its (sorted) columns in the monolithic matrix
            lo = ip[rows[lrow]]
            p = _host.locate(M, rows[lrow], target)
            pos[k] = start[lrow] + (p - lo)
        self.eng.fe_bind_system(pos)

    def _bind_kp_rows(self):
        pp, sub, loc = self.pb, self.sub, self.loc
        Vl = sub.V
        nvl, nc = Vl.nvl, Vl.mesh.num_cells
        pat = Vl._patterns(False)["PP"]
        ptr, src = _contribution_plan(pat.inv, nc, nvl * nvl, pat.nnz, pat)
        cst = None
        idt = 0.0 if loc.pcdr else loc.idt
        if idt:
            M = Vl.area[:, None, None] * Vl._ref()["P"][None] * (idt / loc.nu)
            cst = pat.sum_entries(M)
        q0, q1 = (int(v) for v in np.searchsorted(sub.p_g, pp.fine.own_p))
        e0, e1 = int(pat.indptr[q0]), int(pat.indptr[q1])
        p0, p1 = int(ptr[e0]), int(ptr[e1])
        src = src[p0:p1]
        ptr = (ptr[e0:e1 + 1] - ptr[e0]).astype(np.int32)
        if cst is not None:
            cst = cst[e0:e1]
        self.eng.fe_bind_kp(ptr, src, cst, 1.0 / loc.nu)
        # (the operator took this rank's rows: its value updates carry them)
        self.eng.fe_set_kp_rows(0, e1 - e0)
        self._kp_cut = (e0, e1)
        self._kp_rows = (q0, q1)
        self.nnz_kp = e1 - e0
        # BRM2: - (1/nu) int_inflow (w.n) p q ds (demo_navier-stokes-pcd.py:
        # 131-135) over the inflow facets of the SLAB - every facet that
        # touches an owned pressure row lies in it (its cell touches that row);
        # entries of other ranks' rows are left to their owners, the plan's
        # node numbers index the replicated wind (global)
        if pp.variant == "BRM2" and len(loc.robin_edges) > 0:
            from .device_producer import _group
            pl = Vl.robin_plan(loc.robin_edges)
            nb = pl["length"].size
            pd = pl["pdofs"]
            k = pd.shape[1]
            rows = np.repeat(pd[:, :, None], k, axis=2).ravel()
            cols = np.repeat(pd[:, None, :], k, axis=1).ravel()
            where = pat.locate(rows, cols)
            e_idx, ij = np.divmod(np.arange(where.size), k * k)
            mine = (where >= e0) & (where < e1)
            where, e_idx, ij = where[mine] - e0, e_idx[mine], ij[mine]
            if where.size:
                aff_pos, _, aff_ptr, order = _group(
                    where, np.zeros_like(where), e1 - e0, 1)
                self.eng.fe_bind_robin(
                    sub.nodes_g[pl["nodes"]].T, pl["normal"].T, pl["length"],
                    aff_pos, aff_ptr, (ij * nb + e_idx)[order],
                    np.full(where.size, -1.0 / loc.nu))

    # ------------------------------------------------------------ diagnostics
    def _scalar(self, l):
        """My rows of level ``l``'s scalar operator (global shape, the other
        rows empty), or the whole one where the level is replicated."""
        ip, ix, n = self._pat[l]
        own = self._own[l]
        if own is None:
            vals = self.eng.fe_level_values(l, ix.size)
            return sp.csr_matrix((vals, ix, ip), shape=(n, n))
        o_ip, o_ix, n, r0 = own
        vals = self.eng.fe_level_values(l, o_ix.size)
        full = np.zeros(n + 1, dtype=np.int64)
        nr = o_ip.size - 1
        full[r0 + 1:r0 + nr + 1] = o_ip[1:]
        full[r0 + nr + 1:] = o_ip[-1]
        return sp.csr_matrix((vals, o_ix, full), shape=(n, n))

    def kp_matrix(self):
        """My pressure rows of ``K_p`` (global shape, the other rows empty)."""
        sub, V = self.sub, self.V
        pat = sub.V._patterns(False)["PP"]
        e0, e1 = self._kp_cut
        q0, q1 = self._kp_rows
        vals = self.eng.fe_kp_values(e1 - e0)
        own = self.pb.fine.own_p
        ip = np.zeros(V.n_p + 1, dtype=np.int64)
        ip[own[0] + 1:own[1] + 1] = pat.indptr[q0 + 1:q1 + 1] - e0
        ip[own[1] + 1:] = e1 - e0
        return sp.csr_matrix((vals, sub.p_g[pat.indices[e0:e1]], ip),
                             shape=(V.n_p, V.n_p))

    def update(self, xu, xp):
        raise NotImplementedError(
            "partitioned device producer: the residual is evaluated on the "
            "device (residual() / pcd_fe_picard_solve)")
