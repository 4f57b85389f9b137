// The C ABI and host side of the device operator producer (pcd_fe_*; kernels: pcd_fe.hpp); included by
// pcd_producer.hip (its own translation unit).  One FE level per multigrid level of the velocity
// block (coarsest first); the finest level writes the caller's system values,
// the intermediate ones the multigrid operators, the coarsest one is handed
// back to the host (its explicit inverse is a host computation).
#pragma once

struct FeLevel {
  int64_t nc = 0, nn2 = 0, nnzf = 0, ndiag = 0;
  DBuf<int> dofs2, f_ptr, f_src, diag_pos, inject;
  DBuf<double> gradlam, measure, f_const, diag_val;
  DBuf<unsigned char> f_keep;
  DBuf<double> U, cells, F, ev;
  DBuf<double> cell_h, cells_s, Fa;    // SUPG: cell sizes, its element storage,
                                       // the unstabilised operator (finest level)
  bool set = false, ev_init = false;
  double ev_lam = 0.0;                 // last eigenvalue estimate (warm starts)
  // Galerkin level: F = P^T F_finer P by two weighted gathers (no mesh data)
  bool galerkin = false;
  int64_t nnzb = 0;
  DBuf<int64_t> b_ptr, c_ptr;
  DBuf<int> b_src, c_src;
  DBuf<double> b_w, c_w, B;
  // ... or by the numeric sparse product on fixed patterns (k_spgemm_fixed,
  // pcd_fe_set_level_product): P and P^T with their values, the patterns of
  // the finer level's F, of B = F P and of this level's F = P^T B
  bool product = false;
  int64_t pr_nf = 0, pr_nc = 0;       // fine / coarse scalar rows
  int pr_g1 = 8, pr_g2 = 32;          // lanes per row of the two stages
  DBuf<int> p_rp, p_col, pt_rp, pt_col, ff_rp, ff_col, bb_rp, bb_col, cc_rp, cc_col;
  DBuf<double> p_val, pt_val;
  // ... BY ROWS (pcd_fe_set_level_product_rows: the hierarchy of a partitioned
  // producer): this rank multiplies ITS rows of the finer operator, its TERMS
  // of the coarse rows of other ranks travel to their owners over one
  // all-reduced wire buffer, the terms of its own coarse rows are added in
  // rank order, and (a level the engine replicates) the rows are gathered
  bool prows = false;
  int64_t pr_nown = 0, pr_nterm = 0, pr_nout = 0, pr_wire = 0, pr_goff = 0, pr_gtot = 0;
  std::vector<int64_t> pr_sends;      // (first term, count, wire offset) per destination
  std::vector<int64_t> pr_adds;       // (source 0 terms / 1 wire, source offset, count, offset into pr_pos)
  DBuf<int> pr_pos;                   // where a source's entries land in this rank's rows
  DBuf<double> T, wire, Fown;
  // Newton linearisation: the d*d scalar matrices N_ij (masked), their element
  // storage, the Galerkin intermediate, and where block entry (i, j, k) sits in
  // the values of this level's operator
  DBuf<double> N, cellsN, BN;
  DBuf<int> npos;
  // several ranks: where entry (k, c) of F x I_d sits in the GLOBAL sorted CSR
  // of the level (pcd_fe_bind_pattern), and the values expanded that way
  DBuf<int> kpos;
  DBuf<double> gvals;
  // ... with the plan CUT BY ROWS (pcd_fe_set_rows): cells, contribution lists
  // and F cover the node rows [row0, row0 + nrows) this rank owns of the level
  // only - the element work and the plan memory of a rank are 1 / R of the
  // level's; the winds stay whole (vectors are replicated, operators are not)
  bool rows = false;
  int64_t row0 = 0, nrows = 0;
  void release() {
    b_ptr.release(); c_ptr.release(); b_src.release(); c_src.release();
    b_w.release(); c_w.release(); B.release();
    product = false; prows = false; pr_sends.clear(); pr_adds.clear();
    pr_pos.release(); T.release(); wire.release(); Fown.release();
    p_rp.release(); p_col.release(); pt_rp.release(); pt_col.release(); ff_rp.release(); ff_col.release();
    bb_rp.release(); bb_col.release(); cc_rp.release(); cc_col.release(); p_val.release(); pt_val.release();
    N.release(); cellsN.release(); BN.release(); npos.release();
    kpos.release(); gvals.release();
    cell_h.release(); cells_s.release(); Fa.release();
    dofs2.release(); f_ptr.release(); f_src.release(); diag_pos.release();
    inject.release(); gradlam.release(); measure.release(); f_const.release();
    diag_val.release(); f_keep.release(); U.release(); cells.release();
    F.release(); ev.release();
  }
};

struct FeState {
  int dim = 0, nlev = 0, nq = 0;
  DBuf<double> qw, phi, dphi, psi, phic, qw_s, phi_s, dphi_s;
  int nq_s = 0;
  bool supg = false; double nu = 0.0;
  std::vector<FeLevel> lev;
  DBuf<double> Func;                 // unconstrained finest values (residual)
  bool newton = false;               // coupled block F x I + N (pcd_fe_set_newton)
  DBuf<double> Nunc, Jv, inv_vals, dvec, y2;
  // several ranks: the nonlinear side is replicated (global vectors and
  // operators on every rank), the linear solve is partitioned
  DCsr Ku;                           // global F x I_d of the finest level
  DCsr Ju;                           // ... and the global coupled block (Newton)
  DBuf<int> gperm;                   // split position -> caller's index, all rows
  DBuf<double> sys_tmp;              // row-cut plans: scratch system values (residual)
  DBuf<double> bloc, dxloc;          // this rank's slices for GMRES
  DBuf<int64_t> sys_pos; bool sys_bound = false; int sys_blocks = 0;
  int64_t nnz_kp = 0; double kp_scale = 1.0; bool kp_bound = false;
  // row-cut plan of Kp (pcd_fe_set_kp_rows): the bound entries are this rank's
  // pressure rows, `kp_off` entries into the operator's `kp_glob` values
  int64_t kp_off = 0, kp_glob = 0;
  DBuf<int> kp_ptr, kp_src;
  DBuf<double> kp_const, kp_cells, kp_vals;
  // coarsest level inverted on the device (pattern of its scalar operator)
  int64_t inv_n = 0; bool inv_bound = false;
  DBuf<int> inv_rowptr, inv_col, inv_flag, inv_piv;
  DBuf<double> inv_W, inv_col0, inv_col1;
  // BRM2 boundary term of Kp (pcd_fe_bind_robin)
  bool robin_bound = false; int64_t rb_nb = 0, rb_naff = 0;
  DBuf<int> rb_nodes, rb_pos, rb_src;
  DBuf<int64_t> rb_ptr;
  DBuf<double> rb_normal, rb_length, rb_w, rb_loc, rb_tmp;
  // nonlinear residual on the device (pcd_fe_bind_residual)
  bool res_bound = false, have_mu0 = false;
  DCsr A01raw, A10raw;                 // unconstrained constant blocks
  bool res_rows = false;               // ... of which this rank holds ITS ROWS only (pcd_fe_set_residual_rows)
  DBuf<double> rtmp;
  int64_t n_bc = 0; double idt = 0.0;
  DBuf<int> bc_idx;
  DBuf<double> bc_mult, bc_g, mass, mu0, u0;
  DBuf<double> xd, bd, dxd, xs, bs, vv;
  int mg_slot = -1, est_iters = 12;
  double emin_f = 0.1, emax_f = 1.1;
  DBuf<double> xu, v, ru, y, parts, slot;
  void release() {
    for (auto& l : lev) l.release();
    lev.clear();
    qw.release(); phi.release(); dphi.release(); psi.release(); Func.release();
    Nunc.release(); Jv.release(); inv_vals.release(); dvec.release(); y2.release();
    Ku.release(); Ju.release(); gperm.release(); bloc.release(); dxloc.release();
    sys_tmp.release();
    phic.release(); qw_s.release(); phi_s.release(); dphi_s.release();
    sys_pos.release(); kp_ptr.release(); kp_src.release(); kp_const.release();
    kp_cells.release(); kp_vals.release(); xu.release(); v.release();
    ru.release(); y.release(); parts.release(); slot.release();
    inv_rowptr.release(); inv_col.release(); inv_flag.release(); inv_W.release();
    inv_piv.release(); inv_col0.release(); inv_col1.release();
    rb_nodes.release(); rb_pos.release(); rb_src.release(); rb_ptr.release();
    rb_normal.release(); rb_length.release(); rb_w.release(); rb_loc.release(); rb_tmp.release();
    A01raw.release(); A10raw.release(); rtmp.release(); bc_idx.release(); bc_mult.release();
    bc_g.release(); mass.release(); mu0.release(); u0.release(); xd.release();
    bd.release(); dxd.release(); xs.release(); bs.release(); vv.release();
  }
};

void fe_release(Engine* h) {
  if (!h->fe) return;
  h->fe->release();
  delete h->fe;
  h->fe = nullptr;
}

template <class T>
static int fe_upload(DBuf<T>& b, const T* src, size_t n) {
  CHK(b.ensure(n));
  if (n) HIPCHK(hipMemcpy(b.p, src, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

static FeTables fe_tables(const FeState& fe) {
  return FeTables{fe.nq, fe.qw.p, fe.phi.p, fe.dphi.p, fe.psi.p, fe.phic.p,
                  fe.nq_s, fe.qw_s.p, fe.phi_s.p, fe.dphi_s.p};
}

// one numeric product on fixed patterns, G lanes per row of C
static int fe_spgemm(Engine* h, int G, int64_t nrows, const int* arp, const int* ac, const double* av,
                     const int* brp, const int* bc, const double* bv, const int* crp, const int* cc,
                     double* cv) {
  if (nrows < 1) return 0;
  const int ng = kBlock / G;
  const int grid = (int)std::min<int64_t>((nrows + ng - 1) / ng, (int64_t)g_num_cus * 16);
#define PCD_SPGEMM(GG)                                                                                     \
  hipLaunchKernelGGL(k_spgemm_fixed<GG>, dim3(grid), dim3(kBlock), 0, h->stream, (int)nrows, arp, ac, av, \
                     brp, bc, bv, crp, cc, cv)
  switch (G) {
    case 8: PCD_SPGEMM(8); break;
    case 16: PCD_SPGEMM(16); break;
    case 32: PCD_SPGEMM(32); break;
    default: PCD_SPGEMM(64); break;
  }
#undef PCD_SPGEMM
  return 0;
}

// lanes per row of C for a product whose B rows hold `avg` entries on average
static int fe_spgemm_group(double avg) { return avg <= 8.0 ? 8 : avg <= 16.0 ? 16 : avg <= 40.0 ? 32 : 64; }

// coarse operator as the Galerkin product of the next finer level's one
static int fe_galerkin_level(Engine* h, FeLevel& L, const FeLevel& finer, bool newton, int d2) {
  if (L.prows) {
    // by rows: B = (my rows of F_finer) P_ext, T = (my rows of P)^T B - my
    // TERMS of every coarse row my fine rows reach (MatPtAP on an MPI matrix;
    // amg.PartitionedSA._galerkin on the host)
    if (newton) return fail(PCD_ERR_STATE, "fe: the by-rows product goes with the Picard block");
    CHK(fe_spgemm(h, L.pr_g1, L.pr_nown, L.ff_rp.p, L.ff_col.p, finer.F.p, L.p_rp.p, L.p_col.p, L.p_val.p,
                  L.bb_rp.p, L.bb_col.p, L.B.p));
    CHK(fe_spgemm(h, L.pr_g2, L.pr_nc, L.pt_rp.p, L.pt_col.p, L.pt_val.p, L.bb_rp.p, L.bb_col.p, L.B.p,
                  L.cc_rp.p, L.cc_col.p, L.T.p));
    // the terms of other ranks' coarse rows: contiguous slices of T (rows of a
    // rank are consecutive), each at its place of a wire buffer every rank
    // holds alike - one all-reduce delivers all of them (sums with zeros)
    if (L.pr_wire) {
      HIPCHK(hipMemsetAsync(L.wire.p, 0, (size_t)L.pr_wire * sizeof(double), h->stream));
      for (size_t q = 0; q + 2 < L.pr_sends.size(); q += 3)
        if (L.pr_sends[q + 1])
          HIPCHK(hipMemcpyAsync(L.wire.p + L.pr_sends[q + 2], L.T.p + L.pr_sends[q],
                                (size_t)L.pr_sends[q + 1] * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      if (!h->comm) return fail(PCD_ERR_STATE, "fe: the by-rows product needs the communicator");
      if (h->comm->allreduce(L.wire.p, (size_t)L.pr_wire, h->stream))
        return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
    }
    // my coarse rows: the sources' entries in RANK order (HostComm.sum_rows),
    // each a pattern of its own - positions distinct within one source
    double* own = L.pr_gtot ? L.Fown.p : L.F.p;
    if (L.pr_nout) HIPCHK(hipMemsetAsync(own, 0, (size_t)L.pr_nout * sizeof(double), h->stream));
    for (size_t q = 0; q + 3 < L.pr_adds.size(); q += 4) {
      const int64_t cnt = L.pr_adds[q + 2];
      if (!cnt) continue;
      const double* src = (L.pr_adds[q] ? L.wire.p : L.T.p) + L.pr_adds[q + 1];
      hipLaunchKernelGGL(k_fe_add_at, dim3(grid1d(cnt)), dim3(kBlock), 0, h->stream, (int)cnt,
                         L.pr_pos.p + L.pr_adds[q + 3], src, own);
    }
    if (L.pr_gtot) {
      // a level the engine replicates: every rank's rows, in rank order, are
      // the whole operator (coarse rows are numbered rank by rank)
      HIPCHK(hipMemsetAsync(L.F.p, 0, (size_t)L.pr_gtot * sizeof(double), h->stream));
      if (L.pr_nout)
        HIPCHK(hipMemcpyAsync(L.F.p + L.pr_goff, own, (size_t)L.pr_nout * sizeof(double),
                              hipMemcpyDeviceToDevice, h->stream));
      if (h->comm->allreduce(L.F.p, (size_t)L.pr_gtot, h->stream))
        return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
    }
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (L.product) {
    // B = F_finer P, F = P^T B: two numeric products on the patterns bound by
    // pcd_fe_set_level_product (Newton: the same for each of the d*d blocks)
    CHK(fe_spgemm(h, L.pr_g1, L.pr_nf, L.ff_rp.p, L.ff_col.p, finer.F.p, L.p_rp.p, L.p_col.p, L.p_val.p,
                  L.bb_rp.p, L.bb_col.p, L.B.p));
    CHK(fe_spgemm(h, L.pr_g2, L.pr_nc, L.pt_rp.p, L.pt_col.p, L.pt_val.p, L.bb_rp.p, L.bb_col.p, L.B.p,
                  L.cc_rp.p, L.cc_col.p, L.F.p));
    if (newton)
      for (int m = 0; m < d2; ++m) {
        CHK(fe_spgemm(h, L.pr_g1, L.pr_nf, L.ff_rp.p, L.ff_col.p, finer.N.p + (int64_t)m * finer.nnzf,
                      L.p_rp.p, L.p_col.p, L.p_val.p, L.bb_rp.p, L.bb_col.p, L.BN.p + (int64_t)m * L.nnzb));
        CHK(fe_spgemm(h, L.pr_g2, L.pr_nc, L.pt_rp.p, L.pt_col.p, L.pt_val.p, L.bb_rp.p, L.bb_col.p,
                      L.BN.p + (int64_t)m * L.nnzb, L.cc_rp.p, L.cc_col.p, L.N.p + (int64_t)m * L.nnzf));
      }
    HIPCHK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_fe_wgather, dim3(grid1d(L.nnzb, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                     L.nnzb, L.b_ptr.p, L.b_src.p, L.b_w.p, finer.F.p, L.B.p);
  hipLaunchKernelGGL(k_fe_wgather, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                     L.nnzf, L.c_ptr.p, L.c_src.p, L.c_w.p, L.B.p, L.F.p);
  if (newton) {
    // P = P_s x I_d: every block of P^T (F x I + N) P is the scalar triple
    // product of that block
    for (int m = 0; m < d2; ++m) {
      hipLaunchKernelGGL(k_fe_wgather, dim3(grid1d(L.nnzb, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                         L.nnzb, L.b_ptr.p, L.b_src.p, L.b_w.p, finer.N.p + (int64_t)m * finer.nnzf,
                         L.BN.p + (int64_t)m * L.nnzb);
      hipLaunchKernelGGL(k_fe_wgather, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                         L.nnzf, L.c_ptr.p, L.c_src.p, L.c_w.p, L.BN.p + (int64_t)m * L.nnzb,
                         L.N.p + (int64_t)m * L.nnzf);
    }
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// assemble the scalar velocity operator of one level from the wind `U`.
// Without SUPG: L.F.  With SUPG: L.F = the stabilised operator (what the
// preconditioner and its multigrid use); on the finest level (`top`) also
// L.Fa = the unstabilised one (the system matrix).
static int fe_assemble_level(Engine* h, FeState& fe, FeLevel& L, const double* U,
                             double* unc, bool top) {
  const int na = fe.dim == 2 ? 6 : 10;
  const int64_t nt = (int64_t)na * L.nc;
  const int g = (int)((nt + kBlock - 1) / kBlock);
  const bool supg = fe.supg;
  if (supg && !L.cell_h.p) return fail(PCD_ERR_STATE, "fe: SUPG is on but a level has no cell sizes");
  double* cs = supg ? L.cells_s.p : nullptr;
  if (fe.dim == 2)
    hipLaunchKernelGGL(k_fe_convection_p2<2>, dim3(g), dim3(kBlock), 0, h->stream, (int)L.nc, L.dofs2.p,
                       L.gradlam.p, L.measure.p, fe_tables(fe), U, L.cells.p, L.cell_h.p, fe.nu, cs);
  else
    hipLaunchKernelGGL(k_fe_convection_p2<3>, dim3(g), dim3(kBlock), 0, h->stream, (int)L.nc, L.dofs2.p,
                       L.gradlam.p, L.measure.p, fe_tables(fe), U, L.cells.p, L.cell_h.p, fe.nu, cs);
  double* plain = supg ? (top ? L.Fa.p : nullptr) : L.F.p;
  hipLaunchKernelGGL(k_fe_gather, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                     L.nnzf, L.f_ptr.p, L.f_src.p, L.cells.p, L.f_const.p, L.f_keep.p, unc, plain,
                     cs, supg ? L.F.p : nullptr);
  if (L.ndiag) {
    hipLaunchKernelGGL(k_fe_set, dim3(grid1d(L.ndiag)), dim3(kBlock), 0, h->stream,
                       (int)L.ndiag, L.diag_pos.p, L.diag_val.p, L.F.p);
    if (supg && top)
      hipLaunchKernelGGL(k_fe_set, dim3(grid1d(L.ndiag)), dim3(kBlock), 0, h->stream,
                         (int)L.ndiag, L.diag_pos.p, L.diag_val.p, L.Fa.p);
  }
  if (fe.newton) {
    if (fe.dim == 2)
      hipLaunchKernelGGL(k_fe_newton_p2<2>, dim3(g), dim3(kBlock), 0, h->stream, (int)L.nc, L.dofs2.p,
                         L.gradlam.p, L.measure.p, fe_tables(fe), U, L.cellsN.p);
    else
      hipLaunchKernelGGL(k_fe_newton_p2<3>, dim3(g), dim3(kBlock), 0, h->stream, (int)L.nc, L.dofs2.p,
                         L.gradlam.p, L.measure.p, fe_tables(fe), U, L.cellsN.p);
    hipLaunchKernelGGL(k_fe_gather_blocks, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                       L.nnzf, fe.dim * fe.dim, (int64_t)na * na * L.nc, L.f_ptr.p, L.f_src.p,
                       L.cellsN.p, L.f_keep.p, (top && unc) ? fe.Nunc.p : (double*)nullptr, L.N.p);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// largest eigenvalue (modulus) of D^-1 A by power iteration, warm-started
static int fe_estimate_emax(Engine* h, FeState& fe, FeLevel& L, const DCsr& A,
                            double* lam_out) {
  const int64_t n = A.nrows;
  CHK(L.ev.ensure(n)); CHK(fe.y.ensure(n));
  CHK(fe.parts.ensure(1024)); CHK(fe.slot.ensure(1));
  const int g = grid1d(n, 4, 1024);
  int iters = fe.est_iters;
  double lam = 1.0;
  auto sqnorm_of_y = [&](const double* dinv, double* out) -> int {
    hipLaunchKernelGGL(k_fe_scale_sqnorm, dim3(g), dim3(kBlock), 0, h->stream, n, dinv, fe.y.p, fe.parts.p);
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(kBlock), 0, h->stream, fe.parts.p, g, 0, fe.slot.p);
    if (h->comm && !A.replicated && h->comm->allreduce(fe.slot.p, 1, h->stream))
      return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
    double s = 0.0;
    HIPCHK(hipMemcpyAsync(&s, fe.slot.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = std::sqrt(s);
    return 0;
  };
  if (!L.ev_init) {
    hipLaunchKernelGGL(k_fe_seed, dim3(g), dim3(kBlock), 0, h->stream, n, fe.y.p);
    double nrm = 0.0;
    CHK(sqnorm_of_y(nullptr, &nrm));
    if (!(nrm > 0.0)) return fail(PCD_ERR_STATE, "fe: zero start vector");
    hipLaunchKernelGGL(k_axpby, dim3(grid1d(n, 4)), dim3(kBlock), 0, h->stream, (int)n, 1.0 / nrm, fe.y.p, 0.0, L.ev.p);
    L.ev_init = true;
  } else {
    iters = std::max(3, iters / 4);
  }
  const bool cold = iters == fe.est_iters;
  double best = 0.0;
  for (int it = 0; it < iters; ++it) {
    CHK(spmv(h, A, L.ev.p, fe.y.p));
    CHK(sqnorm_of_y(A.dinv.p, &lam));
    // envelope over the iterations (non-normal operators oscillate); the
    // first two steps of a cold start only shake off the random start
    if (!(cold && it < 2)) best = std::max(best, lam);
    if (!(lam > 0.0) || !std::isfinite(lam))
      return fail(PCD_ERR_STATE, "fe: eigenvalue estimate broke down (%g)", lam);
    hipLaunchKernelGGL(k_axpby, dim3(grid1d(n, 4)), dim3(kBlock), 0, h->stream, (int)n, 1.0 / lam, fe.y.p, 0.0, L.ev.p);
    // a warm start is trusted only while the estimate stays near the last one
    if (it == iters - 1 && iters < fe.est_iters && std::fabs(lam - L.ev_lam) > 0.1 * lam)
      iters = fe.est_iters;
  }
  lam = std::max(best, lam);
  L.ev_lam = lam;
  *lam_out = lam;
  return 0;
}

// level 0 of the multigrid = dense inverse of the coarsest operator F_0 x I_d
static int fe_invert_coarsest(Engine* h, FeState& fe, MgLevel& M0) {
  // Newton: the coarsest operator is a coupled (d n0)^2 matrix, inverted whole
  const int n = (int)fe.inv_n, d = fe.newton ? 1 : fe.dim;
  const int64_t N = (int64_t)n * d;
  // (stored whole, or - Picard, d > 1 - as inv(F) (x) I_d without the entries
  // that couple no components: n per row)
  const bool compact = d > 1 && M0.A.nnz == N * n;
  if (!M0.A.set || M0.A.nrows != N || (M0.A.nnz != N * N && !compact))
    return fail(PCD_ERR_STATE, "fe: multigrid level 0 is not the dense %lld x %lld inverse",
                (long long)N, (long long)N);
  HIPCHK(hipMemsetAsync(fe.inv_flag.p, 0, sizeof(int), h->stream));
  const double* vals0 = fe.lev[0].F.p;
  if (fe.newton) {
    FeLevel& L0 = fe.lev[0];
    hipLaunchKernelGGL(k_fe_scatter_blocks<int>, dim3(grid1d(L0.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                       h->stream, L0.nnzf, fe.dim, L0.npos.p, L0.F.p, L0.N.p, fe.inv_vals.p);
    vals0 = fe.inv_vals.p;
  }
  hipLaunchKernelGGL(k_gj_init, dim3(n), dim3(kBlock), 0, h->stream, n, fe.inv_rowptr.p,
                     fe.inv_col.p, vals0, fe.inv_W.p, fe.inv_col0.p);
  double* cc[2] = {fe.inv_col0.p, fe.inv_col1.p};
  for (int k = 0; k < n; ++k)
    hipLaunchKernelGGL(k_gj_step, dim3(n), dim3(kBlock), 0, h->stream, n, k, fe.inv_W.p,
                       cc[k & 1], cc[(k + 1) & 1], fe.inv_piv.p, fe.inv_flag.p);
  if (compact)
    hipLaunchKernelGGL(k_gj_store_compact, dim3(grid1d(N * n, 4, 1 << 16)), dim3(kBlock), 0, h->stream, n, d,
                       fe.inv_W.p, fe.inv_piv.p, M0.A.val.p);
  else
    hipLaunchKernelGGL(k_gj_store, dim3(grid1d(N * N, 4, 1 << 16)), dim3(kBlock), 0, h->stream, n, d,
                       fe.inv_W.p, fe.inv_piv.p, M0.A.val.p);
  HIPCHK(hipGetLastError());
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, fe.inv_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (flag) return fail(PCD_ERR_STATE, "fe: the coarsest operator is singular");
  CHK(refresh_dinv(h, M0.A));
  return 0;
}

extern "C" {

// invert the coarsest level on the device after every update: rowptr/col =
// CSR pattern of its scalar operator (n0 rows, the order of
// pcd_fe_get_level_values(h, 0, .)); needs pcd_fe_bind_mg
int pcd_fe_bind_coarse_inverse(pcd_handle h, int64_t n0, const int32_t* rowptr,
                               const int32_t* colidx) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_bind_coarse_inverse: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  const int64_t want = fe.newton ? (int64_t)fe.dim * fe.dim * fe.lev[0].nnzf : fe.lev[0].nnzf;
  if (n0 < 1 || n0 > 8192 || !rowptr || !colidx || !fe.lev[0].set || rowptr[n0] != want)
    return fail(PCD_ERR_ARG, "fe_bind_coarse_inverse: bad pattern (n0 <= 8192)");
  if (fe.newton) {
    if (!fe.lev[0].npos.p) return fail(PCD_ERR_STATE, "fe_bind_coarse_inverse: level 0 has no Newton positions");
    CHK(fe.inv_vals.ensure((size_t)want));
  }
  HIPCHK(hipSetDevice(h->device));
  fe.inv_n = n0;
  CHK(fe_upload(fe.inv_rowptr, rowptr, (size_t)n0 + 1));
  CHK(fe_upload(fe.inv_col, colidx, (size_t)rowptr[n0]));
  CHK(fe.inv_W.ensure((size_t)2 * n0 * n0));
  CHK(fe.inv_flag.ensure(1)); CHK(fe.inv_piv.ensure(n0));
  CHK(fe.inv_col0.ensure(n0)); CHK(fe.inv_col1.ensure(n0));
  fe.inv_bound = true;
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_coarse_inverse)

int pcd_fe_begin(pcd_handle h, int dim, int nlevels, int nq, const double* qw,
                 const double* phi, const double* dphi, const double* psi) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (dim != 2 && dim != 3) return fail(PCD_ERR_ARG, "fe_begin: dim must be 2 or 3");
  // the producer's plans address operator entries in the numbering its caller
  // handed over: it works on engines that kept that numbering
  // (a rank-local hand-over - pcd_set_system_local - goes with plans whose
  // positions address THIS RANK'S value arrays: pcd_fe_bind_system checks every
  // position against the values the engine holds; device_producer_rows.py)
  if (h->ru.active() || h->rp.active())
    return fail(PCD_ERR_STATE, "fe_begin: the engine renumbered the dofs at pcd_set_system (the "
                               "caller's numbering was not local); the device producer needs "
                               "PCD_REORDER=none or a local numbering");
  if (nlevels < 1 || nlevels > 32 || nq < 1 || !qw || !phi || !dphi || !psi)
    return fail(PCD_ERR_ARG, "fe_begin: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  fe_release(h);
  h->fe = new FeState();
  FeState& fe = *h->fe;
  fe.dim = dim; fe.nlev = nlevels; fe.nq = nq;
  const int na = dim == 2 ? 6 : 10, nv = dim + 1;
  CHK(fe_upload(fe.qw, qw, nq));
  CHK(fe_upload(fe.phi, phi, (size_t)nq * na));
  CHK(fe_upload(fe.dphi, dphi, (size_t)nq * na * nv));
  CHK(fe_upload(fe.psi, psi, (size_t)nq * nv));
  fe.lev.resize(nlevels);
  return 0;
} PCD_ABI_CATCH(pcd_fe_begin)

int pcd_fe_set_level(pcd_handle h, int level, int64_t ncells, int64_t nn2,
                     const int32_t* dofs2, const double* gradlam,
                     const double* measure, int64_t nnz_f, const int32_t* f_ptr,
                     const int32_t* f_src, const double* f_const,
                     const unsigned char* f_keep, int64_t n_diag,
                     const int32_t* diag_pos, const double* diag_val,
                     const int32_t* inject) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_level: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev) return fail(PCD_ERR_ARG, "fe_set_level: bad level %d", level);
  if (ncells < 1 || nn2 < 1 || nnz_f < 1 || !dofs2 || !gradlam || !measure || !f_ptr || !f_src ||
      !f_const || !f_keep || n_diag < 0 || (n_diag && (!diag_pos || !diag_val)))
    return fail(PCD_ERR_ARG, "fe_set_level: bad arguments");
  if ((level == fe.nlev - 1) != (inject == nullptr))
    return fail(PCD_ERR_ARG, "fe_set_level: every level but the finest needs an injection map");
  const int na = fe.dim == 2 ? 6 : 10, nv = fe.dim + 1;
  if ((int64_t)na * na * ncells >= INT32_MAX)
    return fail(PCD_ERR_ARG, "fe_set_level: element storage exceeds int32 indexing");
  HIPCHK(hipSetDevice(h->device));
  FeLevel& L = fe.lev[level];
  L.galerkin = false;
  L.nc = ncells; L.nn2 = nn2; L.nnzf = nnz_f; L.ndiag = n_diag;
  CHK(fe_upload(L.dofs2, dofs2, (size_t)na * ncells));
  CHK(fe_upload(L.gradlam, gradlam, (size_t)nv * fe.dim * ncells));
  CHK(fe_upload(L.measure, measure, (size_t)ncells));
  CHK(fe_upload(L.f_ptr, f_ptr, (size_t)nnz_f + 1));
  CHK(fe_upload(L.f_src, f_src, (size_t)f_ptr[nnz_f]));
  CHK(fe_upload(L.f_const, f_const, (size_t)nnz_f));
  CHK(fe_upload(L.f_keep, f_keep, (size_t)nnz_f));
  CHK(fe_upload(L.diag_pos, diag_pos, (size_t)n_diag));
  CHK(fe_upload(L.diag_val, diag_val, (size_t)n_diag));
  if (inject) CHK(fe_upload(L.inject, inject, (size_t)nn2));
  CHK(L.cells.ensure((size_t)na * na * ncells));
  CHK(L.F.ensure(nnz_f));
  CHK(L.U.ensure((size_t)fe.dim * nn2));
  L.set = true; L.ev_init = false;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_level)

// Several ranks: the plan of `level` handed to pcd_fe_set_level covers the node
// rows [node_row0, node_row0 + n_node_rows) of the level's scalar operator only
// (this rank's rows, pcd_row_range: the cells that touch them, the contribution
// lists and constants of their entries; dofs2 and the injection map keep GLOBAL
// node numbers, nn2 stays the global node count).  A rank then assembles and
// stores 1 / R of a partitioned level; vectors - the iterate, the winds, the
// residual - stay replicated.  Picard block, re-discretised levels
// (-pc_mg_galerkin none).  Afterwards pcd_fe_bind_pattern / pcd_fe_bind_system /
// the mass values of pcd_fe_bind_residual take these rows' entries.
int pcd_fe_set_rows(pcd_handle h, int level, int64_t node_row0, int64_t n_node_rows) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_rows: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 1 || level >= fe.nlev || !fe.lev[level].set || fe.lev[level].galerkin)
    return fail(PCD_ERR_ARG, "fe_set_rows: level %d is not a re-discretised level above the coarsest one", level);
  if (!h->comm) return fail(PCD_ERR_STATE, "fe_set_rows: no communicator attached (pcd_comm_init first)");
  FeLevel& L = fe.lev[level];
  if (node_row0 < 0 || n_node_rows < 0 || node_row0 + n_node_rows > L.nn2)
    return fail(PCD_ERR_ARG, "fe_set_rows: rows [%lld, %lld) outside the level's %lld nodes",
                (long long)node_row0, (long long)(node_row0 + n_node_rows), (long long)L.nn2);
  L.rows = true; L.row0 = node_row0; L.nrows = n_node_rows;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_rows)

// A coarse level whose operator is the Galerkin product of the next finer one
// (-pc_mg_galerkin both): B = F_finer P (nnz_b entries, entry e = sum
// b_w[t] * F_finer[b_src[t]]), F = P^T B (nnz_f entries, entry k = sum
// c_w[t] * B[c_src[t]]).
int pcd_fe_set_level_galerkin(pcd_handle h, int level, int64_t nnz_f, int64_t nnz_b,
                              const int64_t* b_ptr, const int32_t* b_src,
                              const double* b_w, const int64_t* c_ptr,
                              const int32_t* c_src, const double* c_w) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_level_galerkin: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev - 1)
    return fail(PCD_ERR_ARG, "fe_set_level_galerkin: level %d is not a coarse level", level);
  if (nnz_f < 1 || nnz_b < 1 || !b_ptr || !b_src || !b_w || !c_ptr || !c_src || !c_w)
    return fail(PCD_ERR_ARG, "fe_set_level_galerkin: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  FeLevel& L = fe.lev[level];
  L.release();
  L.galerkin = true; L.nnzf = nnz_f; L.nnzb = nnz_b;
  L.rows = false; L.row0 = 0; L.nrows = 0;
  CHK(fe_upload(L.b_ptr, b_ptr, (size_t)nnz_b + 1));
  CHK(fe_upload(L.b_src, b_src, (size_t)b_ptr[nnz_b]));
  CHK(fe_upload(L.b_w, b_w, (size_t)b_ptr[nnz_b]));
  CHK(fe_upload(L.c_ptr, c_ptr, (size_t)nnz_f + 1));
  CHK(fe_upload(L.c_src, c_src, (size_t)c_ptr[nnz_f]));
  CHK(fe_upload(L.c_w, c_w, (size_t)c_ptr[nnz_f]));
  CHK(L.B.ensure(nnz_b));
  CHK(L.F.ensure(nnz_f));
  L.set = true; L.ev_init = false;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_level_galerkin)

// The same level by a NUMERIC SPARSE PRODUCT on fixed patterns (k_spgemm_fixed):
// the symbolic phase ran on the host once (the aggregation, hence every
// pattern, is kept across the nonlinear iterations), the device keeps P, P^T
// and the three patterns and recomputes values - what the reference's
// transposeMatMult(..., result=) does (fenapack/field_split_backend.py:160-166)
// and what hypre's / GAMG's set-up re-does every outer iteration
// (demo_navier-stokes-pcd.py:153-160).  No per-term lists: the memory held
// for the refresh is the patterns (12 B per entry of P, 4 B per entry of the
// three patterns) instead of ~12 B per TERM of both products.
int pcd_fe_set_level_product(pcd_handle h, int level, int64_t n_fine, int64_t n_coarse,
                             const int32_t* p_rowptr, const int32_t* p_col, const double* p_val,
                             const int32_t* pt_rowptr, const int32_t* pt_col, const double* pt_val,
                             const int32_t* f_rowptr, const int32_t* f_col,
                             const int32_t* b_rowptr, const int32_t* b_col,
                             const int32_t* c_rowptr, const int32_t* c_col) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_level_product: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev - 1)
    return fail(PCD_ERR_ARG, "fe_set_level_product: level %d is not a coarse level", level);
  if (n_fine < 1 || n_coarse < 1 || n_fine > INT32_MAX || !p_rowptr || !p_col || !p_val || !pt_rowptr ||
      !pt_col || !pt_val || !f_rowptr || !f_col || !b_rowptr || !b_col || !c_rowptr || !c_col)
    return fail(PCD_ERR_ARG, "fe_set_level_product: bad arguments");
  const int64_t nnz_p = p_rowptr[n_fine], nnz_f = f_rowptr[n_fine], nnz_b = b_rowptr[n_fine],
                nnz_c = c_rowptr[n_coarse];
  if (pt_rowptr[n_coarse] != nnz_p || nnz_f < 1 || nnz_b < 1 || nnz_c < 1)
    return fail(PCD_ERR_ARG, "fe_set_level_product: P^T does not have P's entries, or an empty pattern");
  HIPCHK(hipSetDevice(h->device));
  FeLevel& L = fe.lev[level];
  L.release();
  L.galerkin = true; L.product = true; L.nnzf = nnz_c; L.nnzb = nnz_b;
  L.rows = false; L.row0 = 0; L.nrows = 0;      // (a whole level: a re-bound one may have been a row block)
  L.pr_nf = n_fine; L.pr_nc = n_coarse;
  CHK(fe_upload(L.p_rp, p_rowptr, (size_t)n_fine + 1));
  CHK(fe_upload(L.p_col, p_col, (size_t)nnz_p));
  CHK(fe_upload(L.p_val, p_val, (size_t)nnz_p));
  CHK(fe_upload(L.pt_rp, pt_rowptr, (size_t)n_coarse + 1));
  CHK(fe_upload(L.pt_col, pt_col, (size_t)nnz_p));
  CHK(fe_upload(L.pt_val, pt_val, (size_t)nnz_p));
  CHK(fe_upload(L.ff_rp, f_rowptr, (size_t)n_fine + 1));
  CHK(fe_upload(L.ff_col, f_col, (size_t)nnz_f));
  CHK(fe_upload(L.bb_rp, b_rowptr, (size_t)n_fine + 1));
  CHK(fe_upload(L.bb_col, b_col, (size_t)nnz_b));
  CHK(fe_upload(L.cc_rp, c_rowptr, (size_t)n_coarse + 1));
  CHK(fe_upload(L.cc_col, c_col, (size_t)nnz_c));
  CHK(L.B.ensure(nnz_b));
  CHK(L.F.ensure(nnz_c));
  // lanes per row of C: along the rows of the product's SECOND factor
  L.pr_g1 = fe_spgemm_group((double)nnz_p / (double)n_fine);
  L.pr_g2 = fe_spgemm_group((double)nnz_b / (double)n_fine);
  // (PCD_SPGEMM_GROUP=8|16|32|64 forces the group size of both stages - the
  // tests use it to send ordinary rows through the several-pass path)
  if (const char* e = getenv("PCD_SPGEMM_GROUP")) {
    const int g = atoi(e);
    if (g == 8 || g == 16 || g == 32 || g == 64) L.pr_g1 = L.pr_g2 = g;
  }
  L.set = true; L.ev_init = false;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_level_product)

// The product BY ROWS - a hierarchy whose levels are partitioned (what every
// rank of the reference does for its rows of fp / kp each outer iteration:
// fenapack/assembling.py:98-106, field_split_backend.py:79-83, 285-291; the
// owned-rows-only rule of SubfieldBC.h:136-155; hypre / GAMG re-forming the
// coarse operators, demo_navier-stokes-pcd.py:153-160).  This rank holds
// n_own node rows of the finer level's scalar F (global columns: f_*), the
// prolongation rows of its own AND its halo nodes (P_ext: n_fine rows, global
// shape, the others empty), the transpose of its own rows (n_coarse x n_own),
// the patterns of B = F_rows P_ext (n_own rows) and of its TERMS T = P_own^T B
// (n_coarse rows).  sends: (first term, count, wire offset) per destination -
// the terms of another rank's coarse rows are consecutive; wire_len doubles
// every rank holds alike, delivered by ONE all-reduce.  adds: (source 0 = T /
// 1 = wire, source offset, count, offset into pos), applied in order, land in
// this rank's n_out entries (pos).  gather_total > 0: the level is replicated -
// the rows of all ranks, this rank's at gather_off, are the level's F.
int pcd_fe_set_level_product_rows(pcd_handle h, int level, int64_t n_own, int64_t n_fine, int64_t n_coarse,
                                  const int32_t* pe_rowptr, const int32_t* pe_col, const double* pe_val,
                                  const int32_t* pt_rowptr, const int32_t* pt_col, const double* pt_val,
                                  const int32_t* f_rowptr, const int32_t* f_col,
                                  const int32_t* b_rowptr, const int32_t* b_col,
                                  const int32_t* t_rowptr, const int32_t* t_col,
                                  int64_t n_out, int64_t wire_len, int n_sends, const int64_t* sends,
                                  int n_adds, const int64_t* adds, const int32_t* pos,
                                  int64_t gather_off, int64_t gather_total,
                                  int64_t node_row0, int64_t n_node_rows) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_level_product_rows: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev - 1)
    return fail(PCD_ERR_ARG, "fe_set_level_product_rows: level %d is not a coarse level", level);
  if (!h->comm) return fail(PCD_ERR_STATE, "fe_set_level_product_rows: no communicator attached");
  if (n_own < 0 || n_fine < 1 || n_coarse < 1 || n_fine > INT32_MAX || !pe_rowptr || !pt_rowptr || !f_rowptr ||
      !b_rowptr || !t_rowptr || n_out < 0 || wire_len < 0 || n_sends < 0 || n_adds < 0 ||
      (n_sends && !sends) || (n_adds && (!adds || !pos)) || gather_off < 0 || gather_total < 0 ||
      (gather_total && gather_off + n_out > gather_total))
    return fail(PCD_ERR_ARG, "fe_set_level_product_rows: bad arguments");
  const int64_t nnz_pe = pe_rowptr[n_fine], nnz_pt = pt_rowptr[n_coarse], nnz_f = f_rowptr[n_own],
                nnz_b = b_rowptr[n_own], nnz_t = t_rowptr[n_coarse];
  if ((nnz_pe && (!pe_col || !pe_val)) || (nnz_pt && (!pt_col || !pt_val)) || (nnz_f && !f_col) ||
      (nnz_b && !b_col) || (nnz_t && !t_col))
    return fail(PCD_ERR_ARG, "fe_set_level_product_rows: null index / value arrays");
  int64_t npos = 0;
  for (int q = 0; q < n_sends; ++q)
    if (sends[3 * q] < 0 || sends[3 * q + 1] < 0 || sends[3 * q] + sends[3 * q + 1] > nnz_t ||
        sends[3 * q + 2] < 0 || sends[3 * q + 2] + sends[3 * q + 1] > wire_len)
      return fail(PCD_ERR_ARG, "fe_set_level_product_rows: send %d outside the terms / the wire", q);
  for (int q = 0; q < n_adds; ++q) {
    const int64_t src = adds[4 * q], off = adds[4 * q + 1], cnt = adds[4 * q + 2], po = adds[4 * q + 3];
    if ((src != 0 && src != 1) || off < 0 || cnt < 0 || po < 0 || off + cnt > (src ? wire_len : nnz_t))
      return fail(PCD_ERR_ARG, "fe_set_level_product_rows: add %d outside its source", q);
    npos = std::max(npos, po + cnt);
  }
  for (int64_t k = 0; k < npos; ++k)
    if (pos[k] < 0 || pos[k] >= n_out) return fail(PCD_ERR_ARG, "fe_set_level_product_rows: a position outside the rows");
  HIPCHK(hipSetDevice(h->device));
  FeLevel& L = fe.lev[level];
  L.release();
  L.galerkin = true; L.prows = true;
  L.nnzb = nnz_b; L.pr_nown = n_own; L.pr_nf = n_fine; L.pr_nc = n_coarse; L.pr_nterm = nnz_t;
  L.pr_nout = n_out; L.pr_wire = wire_len; L.pr_goff = gather_off; L.pr_gtot = gather_total;
  L.nnzf = gather_total ? gather_total : n_out;
  L.pr_sends.assign(sends, sends + 3 * (size_t)n_sends);
  L.pr_adds.assign(adds, adds + 4 * (size_t)n_adds);
  CHK(fe_upload(L.p_rp, pe_rowptr, (size_t)n_fine + 1));
  CHK(fe_upload(L.p_col, pe_col, (size_t)nnz_pe));
  CHK(fe_upload(L.p_val, pe_val, (size_t)nnz_pe));
  CHK(fe_upload(L.pt_rp, pt_rowptr, (size_t)n_coarse + 1));
  CHK(fe_upload(L.pt_col, pt_col, (size_t)nnz_pt));
  CHK(fe_upload(L.pt_val, pt_val, (size_t)nnz_pt));
  CHK(fe_upload(L.ff_rp, f_rowptr, (size_t)n_own + 1));
  CHK(fe_upload(L.ff_col, f_col, (size_t)nnz_f));
  CHK(fe_upload(L.bb_rp, b_rowptr, (size_t)n_own + 1));
  CHK(fe_upload(L.bb_col, b_col, (size_t)nnz_b));
  CHK(fe_upload(L.cc_rp, t_rowptr, (size_t)n_coarse + 1));
  CHK(fe_upload(L.cc_col, t_col, (size_t)nnz_t));
  CHK(fe_upload(L.pr_pos, pos, (size_t)npos));
  CHK(L.B.ensure(std::max<int64_t>(nnz_b, 1)));
  CHK(L.T.ensure(std::max<int64_t>(nnz_t, 1)));
  CHK(L.F.ensure(std::max<int64_t>(L.nnzf, 1)));
  if (wire_len) CHK(L.wire.ensure(wire_len));
  if (gather_total) CHK(L.Fown.ensure(std::max<int64_t>(n_out, 1)));
  L.pr_g1 = fe_spgemm_group(n_own ? (double)nnz_pe / (double)std::max<int64_t>(1, n_own) : 8.0);
  L.pr_g2 = fe_spgemm_group(n_own ? (double)nnz_b / (double)n_own : 8.0);
  if (const char* e = getenv("PCD_SPGEMM_GROUP")) {
    const int g = atoi(e);
    if (g == 8 || g == 16 || g == 32 || g == 64) L.pr_g1 = L.pr_g2 = g;
  }
  // a partitioned level keeps this rank's node rows only (like a row-cut plan)
  L.rows = false; L.row0 = 0; L.nrows = 0;
  if (!gather_total) { L.rows = true; L.row0 = node_row0; L.nrows = n_node_rows; }
  L.nn2 = n_coarse;
  L.set = true; L.ev_init = false;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_level_product_rows)

// SUPG-stabilised preconditioner matrix (fenapack/stabilization.py:39-68 and
// its use at demo_navier-stokes-pcd.py:122-127): per re-discretised level the
// cell sizes h (DOLFIN Cell::h()); nu, the P2 basis at the cell midpoint and
// the quadrature tables of the streamline-diffusion term (degree 6 with a P2
// wind: its own, higher rule) are shared.  From then on the multigrid and A00/A01 are built from the
// stabilised operator, the system matrix from the unstabilised one.
int pcd_fe_set_supg(pcd_handle h, int level, const double* cell_h, double nu,
                    const double* phi_mid, int nq_s, const double* qw_s,
                    const double* phi_s, const double* dphi_s) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_supg: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev || !fe.lev[level].set || fe.lev[level].galerkin)
    return fail(PCD_ERR_ARG, "fe_set_supg: level %d is not a re-discretised level", level);
  if (!cell_h || !phi_mid || !(nu > 0.0) || nq_s < 1 || !qw_s || !phi_s || !dphi_s)
    return fail(PCD_ERR_ARG, "fe_set_supg: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  FeLevel& L = fe.lev[level];
  const int na = fe.dim == 2 ? 6 : 10;
  CHK(fe_upload(L.cell_h, cell_h, (size_t)L.nc));
  CHK(L.cells_s.ensure((size_t)na * na * L.nc));
  if (level == fe.nlev - 1) CHK(L.Fa.ensure(L.nnzf));
  CHK(fe_upload(fe.phic, phi_mid, (size_t)na));
  CHK(fe_upload(fe.qw_s, qw_s, (size_t)nq_s));
  CHK(fe_upload(fe.phi_s, phi_s, (size_t)nq_s * na));
  CHK(fe_upload(fe.dphi_s, dphi_s, (size_t)nq_s * na * (fe.dim + 1)));
  fe.nq_s = nq_s;
  fe.nu = nu; fe.supg = true;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_supg)

// Several ranks (pcd_comm_init before pcd_fe_begin): the scalar CSR pattern of
// a level (nn2 + 1 row pointers, nnz_f sorted column indices - the order of
// pcd_fe_get_level_values).  The producer itself is REPLICATED - every rank
// assembles every level from the replicated iterate, a few ms per nonlinear
// step - and the engine cuts its rows out of the expanded values; this call
// gives it the layout of the global F x I_d (sorted CSR, interleaved dofs).  On
// the finest level it also creates the replicated operator the residual
// applies.  Not needed on one GPU.
int pcd_fe_bind_pattern(pcd_handle h, int level, int64_t nn2, const int32_t* rowptr,
                        const int32_t* colidx) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_bind_pattern: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev || !fe.lev[level].set || !rowptr || !colidx || nn2 < 1)
    return fail(PCD_ERR_ARG, "fe_bind_pattern: level %d is not set / bad arguments", level);
  FeLevel& L = fe.lev[level];
  if (L.rows) {
    // rows [row0, row0 + nrows) only: rowptr holds the GLOBAL entry offsets of
    // these rows (rowptr[0] = entries of the rows before them), colidx their
    // column indices
    const int d = fe.dim;
    if (nn2 != L.nrows || rowptr[nn2] - rowptr[0] != L.nnzf)
      return fail(PCD_ERR_ARG, "fe_bind_pattern: row-cut level %d holds %lld rows / %lld entries, got %lld / %lld",
                  level, (long long)L.nrows, (long long)L.nnzf, (long long)nn2, (long long)(rowptr[nn2] - rowptr[0]));
    if (fe.newton) return fail(PCD_ERR_STATE, "fe_bind_pattern: row-cut plans go with the Picard block");
    HIPCHK(hipSetDevice(h->device));
    std::vector<int32_t> kpos((size_t)d * L.nnzf);
    for (int64_t a = 0; a < nn2; ++a) {
      const int64_t b0 = rowptr[a], len = rowptr[a + 1] - b0;
      if ((b0 + len) * d >= INT32_MAX) return fail(PCD_ERR_ARG, "fe_bind_pattern: values exceed int32 indexing");
      for (int64_t k = b0; k < b0 + len; ++k)
        for (int c = 0; c < d; ++c)
          kpos[(size_t)c * L.nnzf + (k - rowptr[0])] = (int32_t)(d * b0 + c * len + (k - b0));
    }
    CHK(fe_upload(L.kpos, kpos.data(), kpos.size()));
    return 0;                            // (gvals is sized with the level's operator, fe_refresh)
  }
  if (rowptr[nn2] != L.nnzf) return fail(PCD_ERR_ARG, "fe_bind_pattern: %lld entries, the level has %lld",
                                        (long long)rowptr[nn2], (long long)L.nnzf);
  const int d = fe.dim;
  if ((int64_t)d * L.nnzf >= INT32_MAX) return fail(PCD_ERR_ARG, "fe_bind_pattern: values exceed int32 indexing");
  HIPCHK(hipSetDevice(h->device));
  std::vector<int32_t> kpos((size_t)d * L.nnzf);
  for (int64_t a = 0; a < nn2; ++a) {
    const int32_t b0 = rowptr[a], len = rowptr[a + 1] - b0;
    for (int32_t k = b0; k < b0 + len; ++k)
      for (int c = 0; c < d; ++c)
        kpos[(size_t)c * L.nnzf + k] = d * b0 + c * len + (k - b0);
  }
  CHK(fe_upload(L.kpos, kpos.data(), kpos.size()));
  CHK(L.gvals.ensure((size_t)d * L.nnzf));
  if (level == fe.nlev - 1) {
    std::vector<int32_t> rp((size_t)d * nn2 + 1), cc((size_t)d * L.nnzf);
    rp[0] = 0;
    for (int64_t a = 0; a < nn2; ++a) {
      const int32_t b0 = rowptr[a], len = rowptr[a + 1] - b0;
      for (int c = 0; c < d; ++c) {
        rp[d * a + c + 1] = rp[d * a + c] + len;
        for (int32_t k = 0; k < len; ++k) cc[(size_t)d * b0 + c * len + k] = d * colidx[b0 + k] + c;
      }
    }
    CHK(upload_csr(h, fe.Ku, d * nn2, d * nn2, rp.data(), cc.data(), nullptr, nullptr));
    fe.Ku.replicated = true;
    HIPCHK(hipMemsetAsync(fe.Ku.val.p, 0, (size_t)fe.Ku.nnz * sizeof(double), h->stream));
    if (fe.newton) {
      // (pattern of F) x ones(d, d), sorted: row d a + i holds, for every
      // entry b of row a, the columns d b .. d b + d - 1
      if ((int64_t)d * d * L.nnzf >= INT32_MAX) return fail(PCD_ERR_ARG, "fe_bind_pattern: coupled block exceeds int32 indexing");
      std::vector<int32_t> jp((size_t)d * nn2 + 1), jc((size_t)d * d * L.nnzf);
      jp[0] = 0;
      for (int64_t a = 0; a < nn2; ++a) {
        const int32_t b0 = rowptr[a], len = rowptr[a + 1] - b0;
        for (int i = 0; i < d; ++i) {
          const int64_t base = (int64_t)d * d * b0 + (int64_t)i * d * len;
          jp[d * a + i + 1] = (int32_t)(base + (int64_t)d * len);
          for (int32_t k = 0; k < len; ++k)
            for (int j = 0; j < d; ++j) jc[base + (int64_t)k * d + j] = d * colidx[b0 + k] + j;
        }
      }
      CHK(upload_csr(h, fe.Ju, d * nn2, d * nn2, jp.data(), jc.data(), nullptr, nullptr));
      fe.Ju.replicated = true;
      HIPCHK(hipMemsetAsync(fe.Ju.val.p, 0, (size_t)fe.Ju.nnz * sizeof(double), h->stream));
    }
  }
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_pattern)

// Newton linearisation (`--nls newton`, demo_navier-stokes-pcd.py:42,113-116):
// the velocity block becomes F x I_d + N(w), N_ij = (phi_b d_j w_i, phi_a), on
// the pattern (pattern of F) x ones(d, d).  Call once per level after
// pcd_fe_set_level / _galerkin and before pcd_fe_bind_system /
// pcd_fe_bind_coarse_inverse: pos[(i*d+j) * nnz_f + k] = where block entry
// (i, j) of scalar entry k sits in the values of that level's operator - the
// multigrid level's CSR (0 < level < finest), the engine's A00 (finest), the
// CSR handed to pcd_fe_bind_coarse_inverse (level 0, then (d n0) rows).
int pcd_fe_set_newton(pcd_handle h, int level, const int32_t* pos) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_set_newton: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev || !fe.lev[level].set || !pos)
    return fail(PCD_ERR_ARG, "fe_set_newton: level %d is not set / null positions", level);
  FeLevel& L = fe.lev[level];
  const int d2 = fe.dim * fe.dim;
  const int64_t total = (int64_t)d2 * L.nnzf;
  if (total >= INT32_MAX) return fail(PCD_ERR_ARG, "fe_set_newton: block values exceed int32 indexing");
  for (int64_t i = 0; i < total; ++i)
    if (pos[i] < 0 || pos[i] >= total) return fail(PCD_ERR_ARG, "fe_set_newton: position outside the block values");
  HIPCHK(hipSetDevice(h->device));
  CHK(fe_upload(L.npos, pos, (size_t)total));
  CHK(L.N.ensure((size_t)total));
  if (L.galerkin) CHK(L.BN.ensure((size_t)d2 * L.nnzb));
  else {
    const int na = fe.dim == 2 ? 6 : 10;
    CHK(L.cellsN.ensure((size_t)d2 * na * na * L.nc));
  }
  fe.newton = true;
  fe.sys_bound = false;                  // positions of d blocks are not enough any more
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_newton)

// sys_pos[c * nnz_f + k]: where entry k of the finest scalar operator sits, for
// component c, in the caller's system values (pcd_set_system's array).  After
// pcd_fe_set_newton: sys_pos[(i * d + j) * nnz_f + k] for every block (i, j).
int pcd_fe_bind_system(pcd_handle h, const int64_t* sys_pos) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_bind_system: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  FeLevel& L = fe.lev[fe.nlev - 1];
  if (!L.set || !sys_pos) return fail(PCD_ERR_ARG, "fe_bind_system: finest level not set / null map");
  if (!h->mat[PCD_MAT_A].set) return fail(PCD_ERR_STATE, "fe_bind_system: no system set");
  const int nb = fe.newton ? fe.dim * fe.dim : fe.dim;
  for (int64_t i = 0; i < nb * L.nnzf; ++i)
    if (sys_pos[i] < 0 || sys_pos[i] >= h->sys_nnz)
      return fail(PCD_ERR_ARG, "fe_bind_system: position outside the system values");
  HIPCHK(hipSetDevice(h->device));
  CHK(fe_upload(fe.sys_pos, sys_pos, (size_t)nb * L.nnzf));
  fe.sys_bound = true; fe.sys_blocks = nb;
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_system)

int pcd_fe_bind_kp(pcd_handle h, int64_t nnz_kp, const int32_t* kp_ptr,
                   const int32_t* kp_src, const double* kp_const, double scale) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_bind_kp: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  FeLevel& L = fe.lev[fe.nlev - 1];
  if (!L.set || nnz_kp < 1 || !kp_ptr || !kp_src)
    return fail(PCD_ERR_ARG, "fe_bind_kp: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  const int nv = fe.dim + 1;
  fe.nnz_kp = nnz_kp; fe.kp_scale = scale;
  CHK(fe_upload(fe.kp_ptr, kp_ptr, (size_t)nnz_kp + 1));
  CHK(fe_upload(fe.kp_src, kp_src, (size_t)kp_ptr[nnz_kp]));
  if (kp_const) CHK(fe_upload(fe.kp_const, kp_const, (size_t)nnz_kp));
  else fe.kp_const.release();
  CHK(fe.kp_cells.ensure((size_t)nv * nv * L.nc));
  CHK(fe.kp_vals.ensure(nnz_kp));
  fe.kp_off = 0; fe.kp_glob = nnz_kp;
  fe.kp_bound = true;
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_kp)

// Several ranks, plans cut by rows: the entries bound by pcd_fe_bind_kp are this
// rank's pressure rows - `entry_offset` entries into the operator's
// `nnz_global` values (the cells of the finest level's plan must then cover the
// cells that touch these rows as well).
int pcd_fe_set_kp_rows(pcd_handle h, int64_t entry_offset, int64_t nnz_global) try {
  if (!h || !h->fe || !h->fe->kp_bound) return fail(PCD_ERR_STATE, "fe_set_kp_rows: bind Kp first");
  FeState& fe = *h->fe;
  if (!h->comm) return fail(PCD_ERR_STATE, "fe_set_kp_rows: no communicator attached");
  if (entry_offset < 0 || entry_offset + fe.nnz_kp > nnz_global)
    return fail(PCD_ERR_ARG, "fe_set_kp_rows: entries [%lld, %lld) outside the operator's %lld",
                (long long)entry_offset, (long long)(entry_offset + fe.nnz_kp), (long long)nnz_global);
  HIPCHK(hipSetDevice(h->device));
  fe.kp_off = entry_offset; fe.kp_glob = nnz_global;
  CHK(fe.kp_vals.ensure(nnz_global));
  HIPCHK(hipMemset(fe.kp_vals.p, 0, nnz_global * sizeof(double)));
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_kp_rows)

// new constant part of Kp (terms the host keeps assembling, e.g. the BRM2
// boundary integral); NULL = none
int pcd_fe_set_kp_const(pcd_handle h, const double* kp_const) try {
  if (!h || !h->fe || !h->fe->kp_bound) return fail(PCD_ERR_STATE, "fe_set_kp_const: Kp is not bound");
  HIPCHK(hipSetDevice(h->device));
  FeState& fe = *h->fe;
  if (!kp_const) { fe.kp_const.release(); return 0; }
  CHK(fe.kp_const.ensure(fe.nnz_kp));
  HIPCHK(hipMemcpyAsync(fe.kp_const.p, kp_const, fe.nnz_kp * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_kp_const)

// BRM2 boundary term of Kp on the device: per inflow edge (2-D) the P2 nodes
// [3][nb] (start, end, midpoint), outward normals [2][nb], lengths - per inflow
// face (3-D) [6][nb] nodes (vertices, then midpoints of the edges 01, 02, 12),
// [3][nb] normals, areas; the
// affected entries of Kp: aff_pos[n_aff] (distinct positions), each the sum of
// aff_w[t] * loc[aff_src[t]] over aff_ptr (loc = local 2 x 2 matrices stored
// [(i*2+j)][edge]; the weights carry the factor -1/nu).
int pcd_fe_bind_robin(pcd_handle h, int64_t nb, const int32_t* nodes,
                      const double* normals, const double* lengths, int64_t n_aff,
                      const int32_t* aff_pos, const int64_t* aff_ptr,
                      const int32_t* aff_src, const double* aff_w) try {
  if (!h || !h->fe || !h->fe->kp_bound) return fail(PCD_ERR_STATE, "fe_bind_robin: bind Kp first");
  FeState& fe = *h->fe;
  if (nb < 0 || n_aff < 0 || (nb && (!nodes || !normals || !lengths || !aff_pos || !aff_ptr || !aff_src || !aff_w)))
    return fail(PCD_ERR_ARG, "fe_bind_robin: bad arguments");
  for (int64_t i = 0; i < n_aff; ++i)
    if (aff_pos[i] < 0 || aff_pos[i] >= fe.kp_glob) return fail(PCD_ERR_ARG, "fe_bind_robin: position outside Kp");
  HIPCHK(hipSetDevice(h->device));
  fe.rb_nb = nb; fe.rb_naff = n_aff;
  // (3-D: six P2 nodes per boundary face, three normal components, `lengths`
  // = areas, local 3 x 3 matrices)
  const size_t nnode = fe.dim == 2 ? 3 : 6, nloc = fe.dim == 2 ? 4 : 9;
  CHK(fe_upload(fe.rb_nodes, nodes, nnode * nb));
  CHK(fe_upload(fe.rb_normal, normals, (size_t)fe.dim * nb));
  CHK(fe_upload(fe.rb_length, lengths, (size_t)nb));
  CHK(fe_upload(fe.rb_pos, aff_pos, (size_t)n_aff));
  CHK(fe_upload(fe.rb_ptr, aff_ptr, (size_t)n_aff + 1));
  CHK(fe_upload(fe.rb_src, aff_src, (size_t)(n_aff ? aff_ptr[n_aff] : 0)));
  CHK(fe_upload(fe.rb_w, aff_w, (size_t)(n_aff ? aff_ptr[n_aff] : 0)));
  CHK(fe.rb_loc.ensure(nloc * nb)); CHK(fe.rb_tmp.ensure((size_t)n_aff));
  fe.robin_bound = true;
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_robin)

// multigrid hierarchy of inner solve `slot` follows the FE levels; smoother
// bounds after every update: [emin_factor, emax_factor] * lambda_max(D^-1 A)
int pcd_fe_bind_mg(pcd_handle h, int slot, double emin_factor,
                   double emax_factor, int iters) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_bind_mg: call pcd_fe_begin first");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "fe_bind_mg: bad slot %d", slot);
  FeState& fe = *h->fe;
  if ((int)h->inner[slot].mg.size() != fe.nlev)
    return fail(PCD_ERR_STATE, "fe_bind_mg: %d multigrid levels, %d FE levels",
                (int)h->inner[slot].mg.size(), fe.nlev);
  if (!(emax_factor > emin_factor && emin_factor > 0.0) || iters < 1)
    return fail(PCD_ERR_ARG, "fe_bind_mg: bad smoother factors");
  fe.mg_slot = slot; fe.emin_f = emin_factor; fe.emax_f = emax_factor; fe.est_iters = iters;
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_mg)

// Re-assemble everything that depends on the iterate `xu` (velocity dofs,
// fieldsplit-local numbering) and refresh the engine's operators in place.
// Optional: ru = (unconstrained velocity operator) * v, the matrix-dependent
// part of the nonlinear residual (nonlinear_solvers.py:85-112 `F`).
}  // extern "C" (helpers below are internal)

// Everything that depends on the iterate, refreshed in place from the device
// vector dxu; `want_unc`: also keep the unmasked finest operator (residual).
static int fe_refresh(Engine* h, FeState& fe, const double* dxu, bool want_unc) {
  for (auto& L : fe.lev) if (!L.set) return fail(PCD_ERR_STATE, "fe_update: a level is not set");
  for (int l = 1; l < fe.nlev; ++l)
    if (fe.lev[l].galerkin && !fe.lev[l - 1].galerkin)
      return fail(PCD_ERR_STATE, "fe_update: level %d is re-discretised below the Galerkin level %d", l - 1, l);
  const int top = fe.nlev - 1;
  FeLevel& Lt = fe.lev[top];
  const int64_t nu = fe.dim * Lt.nn2;
  // winds: finest = iterate, coarser levels by injection
  HIPCHK(hipMemcpyAsync(Lt.U.p, dxu, nu * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  for (int l = top - 1; l >= 0; --l) {
    FeLevel& L = fe.lev[l];
    if (L.galerkin) break;               // Galerkin levels (and below) need no wind
    hipLaunchKernelGGL(k_fe_inject, dim3(grid1d(L.nn2, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                       L.nn2, fe.dim, L.inject.p, fe.lev[l + 1].U.p, L.U.p);
  }
  if (want_unc) CHK(fe.Func.ensure(Lt.nnzf));
  const int d2 = fe.dim * fe.dim;
  if (fe.newton) {
    for (auto& L : fe.lev)
      if (!L.npos.p) return fail(PCD_ERR_STATE, "fe_update: pcd_fe_set_newton was not called for every level");
    if (fe.sys_bound && fe.sys_blocks != d2)
      return fail(PCD_ERR_STATE, "fe_update: bind the system after pcd_fe_set_newton (d*d blocks)");
    if (want_unc) CHK(fe.Nunc.ensure((size_t)d2 * Lt.nnzf));
  }
  for (int l = top; l >= 0; --l) {
    if (fe.lev[l].galerkin) CHK(fe_galerkin_level(h, fe.lev[l], fe.lev[l + 1], fe.newton, fe.dim * fe.dim));
    else CHK(fe_assemble_level(h, fe, fe.lev[l], fe.lev[l].U.p, (l == top && want_unc) ? fe.Func.p : nullptr, l == top));
  }

  // finest level -> the caller's system values -> A, A00, A01 (+ diagonal)
  if (fe.sys_bound) {
    if (!h->sysvals.p || (int64_t)h->sysvals.n < h->sys_nnz)
      return fail(PCD_ERR_STATE, "fe_update: system values were never staged from the host");
    if (fe.supg) {
      // operator A from the unstabilised values, preconditioner blocks from
      // the stabilised ones (nonlinear_solvers.py:75-76: P != A)
      if (!h->psysvals.p || (int64_t)h->psysvals.n < h->sys_nnz)
        return fail(PCD_ERR_STATE, "fe_update: SUPG needs the preconditioner values staged (pvals of pcd_set_system)");
      if (fe.newton) {
        hipLaunchKernelGGL(k_fe_scatter_blocks<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                           h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, Lt.Fa.p, Lt.N.p, h->sysvals.p);
        hipLaunchKernelGGL(k_fe_scatter_blocks<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                           h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, Lt.F.p, Lt.N.p, h->psysvals.p);
      } else {
        hipLaunchKernelGGL(k_fe_scatter<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                           h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, Lt.Fa.p, h->sysvals.p);
        hipLaunchKernelGGL(k_fe_scatter<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                           h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, Lt.F.p, h->psysvals.p);
      }
      CHK(pcd_update_system(h, h->sysvals.p, h->psysvals.p, PCD_MEM_DEVICE));
    } else {
      if (h->psysvals.p)
        return fail(PCD_ERR_STATE, "fe_update: a separate preconditioner matrix needs pcd_fe_set_supg");
      if (fe.newton)
        hipLaunchKernelGGL(k_fe_scatter_blocks<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                           h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, Lt.F.p, Lt.N.p, h->sysvals.p);
      else
        hipLaunchKernelGGL(k_fe_scatter<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                           h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, Lt.F.p, h->sysvals.p);
      CHK(pcd_update_system(h, h->sysvals.p, nullptr, PCD_MEM_DEVICE));
    }
  }
  // intermediate multigrid levels + smoother bounds
  if (fe.mg_slot >= 0) {
    Inner& s = h->inner[fe.mg_slot];
    if ((int)s.mg.size() != fe.nlev) return fail(PCD_ERR_STATE, "fe_update: multigrid hierarchy changed");
    for (int l = 1; l <= top; ++l) {
      FeLevel& L = fe.lev[l];
      MgLevel& M = s.mg[l];
      M.fused = false;                   // composed from the previous iterate
      const DCsr* A = &h->mat[kSlotMat[fe.mg_slot]];
      if (l < top) {
        if (h->comm) {
          // the level's operator is this rank's slice (or a replica) of the
          // global F x I_d: expand the replicated scalar values into the
          // global CSR order and let the ordinary value refresh cut it
          if (!L.kpos.p) return fail(PCD_ERR_STATE, "fe_update: several ranks need pcd_fe_bind_pattern on every level");
          const int nb = fe.newton ? d2 : fe.dim;
          if (!M.A.set || (!L.rows && M.A.gnnz != (int64_t)nb * L.nnzf) || (L.rows && M.A.replicated))
            return fail(PCD_ERR_STATE, "fe_update: multigrid level %d does not have the FE pattern", l);
          // (row-cut plan: only this rank's rows of the global value array are
          // written - the only ones its operator gathers)
          CHK(L.gvals.ensure((size_t)(L.rows ? M.A.gnnz : nb * L.nnzf)));
          if (fe.newton)        // npos = positions in the global coupled CSR
            hipLaunchKernelGGL(k_fe_scatter_blocks<int>, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                               h->stream, L.nnzf, fe.dim, L.npos.p, L.F.p, L.N.p, L.gvals.p);
          else
            hipLaunchKernelGGL(k_fe_scatter<int>, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                               h->stream, L.nnzf, fe.dim, L.kpos.p, L.F.p, L.gvals.p);
          CHK(refresh_values(h, M.A, L.gvals.p, PCD_MEM_DEVICE));
        } else if (fe.newton) {
          if (!M.A.set || M.A.nnz != (int64_t)d2 * L.nnzf)
            return fail(PCD_ERR_STATE, "fe_update: multigrid level %d is not the coupled block on the FE pattern", l);
          hipLaunchKernelGGL(k_fe_scatter_blocks<int>, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                             h->stream, L.nnzf, fe.dim, L.npos.p, L.F.p, L.N.p, M.A.val.p);
        } else {
          if (!M.A.set || M.A.kron != fe.dim || M.A.nnz2 != L.nnzf)
            return fail(PCD_ERR_STATE, "fe_update: multigrid level %d is not F x I_%d on the FE pattern", l, fe.dim);
          hipLaunchKernelGGL(k_fe_scatter<int>, dim3(grid1d(L.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                             h->stream, L.nnzf, fe.dim, M.A.kron_pos.p, L.F.p, M.A.val.p);
        }
        CHK(refresh_dinv(h, M.A));
        A = &M.A;
      }
      double lam = 0.0;
      CHK(fe_estimate_emax(h, fe, L, *A, &lam));
      M.emin = fe.emin_f * lam; M.emax = fe.emax_f * lam;
    }
    if (fe.inv_bound) CHK(fe_invert_coarsest(h, fe, s.mg[0]));
    ++h->gen;                            // Chebyshev coefficients are baked in
  }
  // Kp
  if (fe.kp_bound) {
    const int g = (int)((Lt.nc + kBlock - 1) / kBlock);
    if (fe.dim == 2)
      hipLaunchKernelGGL(k_fe_convection_p1<2>, dim3(g), dim3(kBlock), 0, h->stream, (int)Lt.nc,
                         Lt.dofs2.p, Lt.gradlam.p, Lt.measure.p, fe_tables(fe), Lt.U.p, fe.kp_scale, fe.kp_cells.p);
    else
      hipLaunchKernelGGL(k_fe_convection_p1<3>, dim3(g), dim3(kBlock), 0, h->stream, (int)Lt.nc,
                         Lt.dofs2.p, Lt.gradlam.p, Lt.measure.p, fe_tables(fe), Lt.U.p, fe.kp_scale, fe.kp_cells.p);
    hipLaunchKernelGGL(k_fe_gather, dim3(grid1d(fe.nnz_kp, 1, 1 << 20)), dim3(kBlock), 0, h->stream,
                       fe.nnz_kp, fe.kp_ptr.p, fe.kp_src.p, fe.kp_cells.p, fe.kp_const.p,
                       (const unsigned char*)nullptr, (double*)nullptr, fe.kp_vals.p + fe.kp_off,
                       (const double*)nullptr, (double*)nullptr);
    if (fe.robin_bound && fe.rb_nb) {
      if (fe.dim == 2)
        hipLaunchKernelGGL(k_fe_robin_edges, dim3(grid1d(fe.rb_nb)), dim3(kBlock), 0, h->stream, (int)fe.rb_nb,
                           fe.rb_nodes.p, fe.rb_normal.p, fe.rb_length.p, Lt.U.p, fe.rb_loc.p);
      else
        hipLaunchKernelGGL(k_fe_robin_faces, dim3(grid1d(fe.rb_nb)), dim3(kBlock), 0, h->stream, (int)fe.rb_nb,
                           fe.rb_nodes.p, fe.rb_normal.p, fe.rb_length.p, Lt.U.p, fe.rb_loc.p);
      hipLaunchKernelGGL(k_fe_wgather, dim3(grid1d(fe.rb_naff)), dim3(kBlock), 0, h->stream, fe.rb_naff,
                         fe.rb_ptr.p, fe.rb_src.p, fe.rb_w.p, fe.rb_loc.p, fe.rb_tmp.p);
      hipLaunchKernelGGL(k_fe_add_at, dim3(grid1d(fe.rb_naff)), dim3(kBlock), 0, h->stream, (int)fe.rb_naff,
                         fe.rb_pos.p, fe.rb_tmp.p, fe.kp_vals.p);
    }
    HIPCHK(hipGetLastError());
    if (h->mat[PCD_MAT_KP].set) {
      if ((h->comm ? h->mat[PCD_MAT_KP].gnnz : h->mat[PCD_MAT_KP].nnz) != fe.kp_glob)
        return fail(PCD_ERR_STATE, "fe_update: Kp pattern differs from the FE pattern");
      CHK(pcd_update_values(h, PCD_MAT_KP, fe.kp_vals.p, PCD_MEM_DEVICE));
    }
  }
  return 0;
}

// y = (F_unconstrained x I_d) v on the pattern of A00 (needs fe_refresh with
// want_unc)
// y = (S x I_d + [Nb]) v on the pattern of the coupled A00 (Newton): the values
// are scattered into a scratch array that stands in for A00's own
static int fe_apply_blocks(Engine* h, FeState& fe, const double* S, const double* Nb,
                           const double* dv, double* dy) {
  FeLevel& Lt = fe.lev[fe.nlev - 1];
  if (h->comm) {                         // replicated global operator
    if (!fe.Ju.set) return fail(PCD_ERR_STATE, "fe: several ranks need pcd_fe_bind_pattern after pcd_fe_set_newton");
    hipLaunchKernelGGL(k_fe_scatter_blocks<int>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                       h->stream, Lt.nnzf, fe.dim, Lt.npos.p, S, Nb, fe.Ju.val.p);
    return spmv(h, fe.Ju, dv, dy);
  }
  DCsr& A = h->mat[PCD_MAT_A00];
  if (!A.set || A.kron || A.nnz != (int64_t)fe.dim * fe.dim * Lt.nnzf)
    return fail(PCD_ERR_STATE, "fe_update: A00 is not the coupled block on the FE pattern");
  CHK(fe.Jv.ensure((size_t)A.nnz));
  hipLaunchKernelGGL(k_fe_scatter_blocks<int>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                     h->stream, Lt.nnzf, fe.dim, Lt.npos.p, S, Nb, fe.Jv.p);
  std::swap(A.val.p, fe.Jv.p);
  const int rc = spmv(h, A, dv, dy);
  std::swap(A.val.p, fe.Jv.p);
  return rc;
}

// Newton: y -= N_unc (xu - v), the part of J d (d = boundary defect of the
// iterate) that the Picard operator does not carry (FlowProblem.linearise:
// F_u = A00_picard x_u - A00_newton d)
static int fe_subtract_newton_defect(Engine* h, FeState& fe, const double* dxu, const double* dv, double* dy) {
  const int64_t nu = fe.dim * fe.lev[fe.nlev - 1].nn2;
  CHK(fe.dvec.ensure(nu)); CHK(fe.y2.ensure(nu));
  HIPCHK(hipMemcpyAsync(fe.dvec.p, dxu, nu * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  hipLaunchKernelGGL(k_axpby, dim3(grid1d(nu, 4)), dim3(kBlock), 0, h->stream, (int)nu, -1.0, dv, 1.0, fe.dvec.p);
  CHK(fe_apply_blocks(h, fe, nullptr, fe.Nunc.p, fe.dvec.p, fe.y2.p));
  hipLaunchKernelGGL(k_axpby, dim3(grid1d(nu, 4)), dim3(kBlock), 0, h->stream, (int)nu, -1.0, fe.y2.p, 1.0, dy);
  return 0;
}

// several ranks, row-cut plans: y = (S x I_d) v by rows.  S holds this rank's
// rows of the scalar operator; they take the place of A00's values for one
// SpMV (scattered into a scratch copy of the system values at A00's positions,
// gathered through the operator's provenance like any value refresh), the
// engine's partitioned A00 - halo exchange and all - applies them to this
// rank's slice of the replicated vector, the slices are summed into a replica,
// and A00 gets its own values back.
static int fe_apply_rows(Engine* h, FeState& fe, const double* S, const double* dv, double* dy) {
  FeLevel& Lt = fe.lev[fe.nlev - 1];
  DCsr& A = h->mat[PCD_MAT_A00];
  if (!fe.sys_bound || !A.set || !A.has_src)
    return fail(PCD_ERR_STATE, "fe: row-cut plans need the system bound (pcd_fe_bind_system)");
  const int64_t nu = fe.dim * Lt.nn2, r0 = h->sp_u.bounds[0][h->rank];
  CHK(fe.sys_tmp.ensure(h->sys_nnz));
  hipLaunchKernelGGL(k_fe_scatter<int64_t>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                     h->stream, Lt.nnzf, fe.dim, fe.sys_pos.p, S, fe.sys_tmp.p);
  CHK(gather_block_values(h, A, fe.sys_tmp.p));
  CHK(refresh_kron(h, A));
  HIPCHK(hipMemsetAsync(dy, 0, nu * sizeof(double), h->stream));
  int rc = spmv(h, A, dv + r0, dy + r0);
  if (!rc && h->comm->allreduce(dy, nu, h->stream)) rc = fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  // A00's own values back (the preconditioner's: P may differ from A)
  const double* own = (h->p_is_a || !h->psysvals.p) ? h->sysvals.p : h->psysvals.p;
  CHK(gather_block_values(h, A, own));
  CHK(refresh_kron(h, A));
  return rc;
}

// several ranks: y = (S x I_d) v with the replicated global operator Ku
static int fe_apply_global(Engine* h, FeState& fe, const double* S, const double* dv, double* dy) {
  FeLevel& Lt = fe.lev[fe.nlev - 1];
  if (Lt.rows) return fe_apply_rows(h, fe, S, dv, dy);
  if (!fe.Ku.set || !Lt.kpos.p)
    return fail(PCD_ERR_STATE, "fe: several ranks need pcd_fe_bind_pattern on the finest level");
  hipLaunchKernelGGL(k_fe_scatter<int>, dim3(grid1d(Lt.nnzf, 1, 1 << 20)), dim3(kBlock), 0,
                     h->stream, Lt.nnzf, fe.dim, Lt.kpos.p, S, fe.Ku.val.p);
  CHK(refresh_kron(h, fe.Ku));
  return spmv(h, fe.Ku, dv, dy);
}

static int fe_apply_unconstrained(Engine* h, FeState& fe, const double* dv, double* dy) {
  DCsr& A = h->mat[PCD_MAT_A00];
  FeLevel& Lt = fe.lev[fe.nlev - 1];
  if (h->comm) return fe_apply_global(h, fe, fe.Func.p, dv, dy);
  if (fe.newton) return fe_apply_blocks(h, fe, fe.Func.p, nullptr, dv, dy);
  if (!A.set || A.kron != fe.dim || A.nnz2 != Lt.nnzf)
    return fail(PCD_ERR_STATE, "fe_update: A00 is not F x I_%d on the FE pattern", fe.dim);
  if (!kron_ok(A, dv, dy))
    return fail(PCD_ERR_ARG, "fe_update: v / ru must be 16-byte aligned for the two-component SpMV");
  return spmv_other_values(h, A, fe.Func.p, dv, dy);   // same pattern, unmasked values
}

extern "C" {

int pcd_fe_update(pcd_handle h, const double* xu, const double* v, double* ru,
                  int mem) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_update: call pcd_fe_begin first");
  if (!xu || ((v == nullptr) != (ru == nullptr))) return fail(PCD_ERR_ARG, "fe_update: bad vectors");
  FeState& fe = *h->fe;
  HIPCHK(hipSetDevice(h->device));
  const int64_t nu = fe.dim * fe.lev[fe.nlev - 1].nn2;
  const double *dxu = xu, *dv = v;
  double* dru = ru;
  if (mem == PCD_MEM_HOST) {
    CHK(fe.xu.ensure(nu));
    HIPCHK(hipMemcpyAsync(fe.xu.p, xu, nu * sizeof(double), hipMemcpyHostToDevice, h->stream));
    dxu = fe.xu.p;
    if (v) {
      CHK(fe.v.ensure(nu)); CHK(fe.ru.ensure(nu));
      HIPCHK(hipMemcpyAsync(fe.v.p, v, nu * sizeof(double), hipMemcpyHostToDevice, h->stream));
      dv = fe.v.p; dru = fe.ru.p;
    }
  }
  CHK(fe_refresh(h, fe, dxu, v != nullptr));
  if (v) {
    CHK(fe_apply_unconstrained(h, fe, dv, dru));
    if (fe.newton) CHK(fe_subtract_newton_defect(h, fe, dxu, dv, dru));
    if (mem == PCD_MEM_HOST)
      HIPCHK(hipMemcpyAsync(ru, dru, nu * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_fe_update)

}  // extern "C"

// b (caller's mixed numbering, device) = nonlinear residual at the device
// iterate x, with every iterate-dependent operator refreshed on the way:
//   F_u = (F_unc x I)(x_u - d) + B^T_raw x_p - idt M u0,  F_p = B_raw (x_u - d),
//   d = boundary defect, Dirichlet rows F_u[bc] = mult d[bc]
// (the residual of demo_navier-stokes-pcd.py:104-120 with DOLFIN's symmetric
// BC application, fenapack/assembling.py:143-155)
static int fe_residual_dev(Engine* h, FeState& fe, const double* xd, double* bd, double* norm) {
  const int64_t nu = h->n_u, np = h->n_p, n = nu + np;
  CHK(fe.xs.ensure(n)); CHK(fe.bs.ensure(n)); CHK(fe.vv.ensure(nu));
  CHK(h->gparts.ensure(2 * 512)); CHK(h->gh.ensure(4)); CHK(ensure_pinned(h, 8));
  const int g1 = grid1d(n, 1);
  // several ranks: every rank evaluates the whole residual from the replicated
  // iterate (one pass over the unconstrained operators: a fraction of a
  // GMRES iteration), so perm is the global one and the norm needs no reduction
  const int* perm = h->comm ? fe.gperm.p : h->perm.p;
  if (h->comm && !perm) return fail(PCD_ERR_STATE, "fe_residual: bind the residual after pcd_comm_init");
  hipLaunchKernelGGL(k_gather, dim3(g1), dim3(kBlock), 0, h->stream, (int)n, perm, xd, fe.xs.p);
  const double *xu = fe.xs.p, *xp = fe.xs.p + nu;
  double *Fu = fe.bs.p, *Fp = fe.bs.p + nu;
  CHK(fe_refresh(h, fe, xu, true));
  HIPCHK(hipMemcpyAsync(fe.vv.p, xu, nu * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (fe.n_bc)
    hipLaunchKernelGGL(k_fe_bc_replace, dim3(grid1d(fe.n_bc)), dim3(kBlock), 0, h->stream,
                       (int)fe.n_bc, fe.bc_idx.p, fe.bc_g.p, fe.vv.p);
  CHK(fe_apply_unconstrained(h, fe, fe.vv.p, Fu));
  if (fe.newton) CHK(fe_subtract_newton_defect(h, fe, xu, fe.vv.p, Fu));
  if (fe.res_rows) {
    // B^T by rows: this rank's rows of B^T x_p (the others are zero), summed
    // into every replica - like F_unc v above (fe_apply_rows)
    CHK(fe.rtmp.ensure(nu));
    CHK(spmv(h, fe.A01raw, xp, fe.rtmp.p));
    if (h->comm->allreduce(fe.rtmp.p, (size_t)nu, h->stream))
      return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
    hipLaunchKernelGGL(k_axpby, dim3(grid1d(nu, 4)), dim3(kBlock), 0, h->stream, (int)nu, 1.0, fe.rtmp.p, 1.0, Fu);
  } else {
    CHK(spmv(h, fe.A01raw, xp, Fu, 1, Fu));
  }
  if (fe.have_mu0)
    hipLaunchKernelGGL(k_axpby, dim3(grid1d(nu, 4)), dim3(kBlock), 0, h->stream, (int)nu, -1.0, fe.mu0.p, 1.0, Fu);
  CHK(spmv(h, fe.A10raw, fe.vv.p, Fp));
  if (fe.res_rows && h->comm->allreduce(Fp, (size_t)np, h->stream))
    return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  if (fe.n_bc)
    hipLaunchKernelGGL(k_fe_bc_rows, dim3(grid1d(fe.n_bc)), dim3(kBlock), 0, h->stream,
                       (int)fe.n_bc, fe.bc_idx.p, fe.bc_g.p, fe.bc_mult.p, xu, Fu);
  hipLaunchKernelGGL(k_scatter, dim3(g1), dim3(kBlock), 0, h->stream, (int)n, perm, fe.bs.p, bd);
  HIPCHK(hipGetLastError());
  if (!h->comm) return dev_norm(h, n, fe.bs.p, norm);
  const int G = grid1d(n, 4, 512);
  hipLaunchKernelGGL(k_mdot, dim3(G, 1), dim3(kBlock), 0, h->stream, n, fe.bs.p, (int64_t)0, 1, fe.bs.p, h->gparts.p, G);
  hipLaunchKernelGGL(k_mdot_reduce, dim3(1), dim3(kBlock), 0, h->stream, h->gparts.p, G, h->gh.p);
  HIPCHK(hipMemcpyAsync(h->pinned, h->gh.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *norm = std::sqrt(h->pinned[0]);
  return 0;
}

extern "C" {

// Constant pieces of the nonlinear residual: the unconstrained blocks
// B^T (n_u x n_p) and B (n_p x n_u), the Dirichlet dofs of the velocity
// (fieldsplit-local indices) with the diagonal values of their rows, and for
// time stepping the scalar mass values on the pattern of F with idt.
int pcd_fe_bind_residual(pcd_handle h, const int32_t* bt_rowptr, const int32_t* bt_col,
                         const double* bt_val, const int32_t* b_rowptr,
                         const int32_t* b_col, const double* b_val, int64_t n_bc,
                         const int32_t* bc_idx, const double* bc_mult,
                         const double* mass_vals, double idt) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_bind_residual: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (!fe.sys_bound) return fail(PCD_ERR_STATE, "fe_bind_residual: bind the system first");
  if (!bt_rowptr || !bt_col || !bt_val || !b_rowptr || !b_col || !b_val || n_bc < 0 ||
      (n_bc && (!bc_idx || !bc_mult)) || (idt != 0.0 && !mass_vals))
    return fail(PCD_ERR_ARG, "fe_bind_residual: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  const int64_t nu = h->n_u, np = h->n_p;
  for (int64_t i = 0; i < n_bc; ++i)
    if (bc_idx[i] < 0 || bc_idx[i] >= nu) return fail(PCD_ERR_ARG, "fe_bind_residual: bc index out of range");
  CHK(upload_csr(h, fe.A01raw, nu, np, bt_rowptr, bt_col, bt_val, nullptr));
  CHK(upload_csr(h, fe.A10raw, np, nu, b_rowptr, b_col, b_val, nullptr));
  if (h->comm) {
    fe.A01raw.replicated = fe.A10raw.replicated = true;
    if ((int64_t)h->perm_glob.size() != nu + np) return fail(PCD_ERR_STATE, "fe_bind_residual: no system set");
    CHK(fe_upload(fe.gperm, h->perm_glob.data(), h->perm_glob.size()));
  }
  fe.n_bc = n_bc; fe.idt = idt;
  CHK(fe_upload(fe.bc_idx, bc_idx, (size_t)n_bc));
  CHK(fe_upload(fe.bc_mult, bc_mult, (size_t)n_bc));
  CHK(fe.bc_g.ensure((size_t)n_bc));
  if (n_bc) HIPCHK(hipMemset(fe.bc_g.p, 0, n_bc * sizeof(double)));
  if (mass_vals) CHK(fe_upload(fe.mass, mass_vals, (size_t)fe.lev[fe.nlev - 1].nnzf));
  fe.have_mu0 = false;
  fe.res_rows = false;                   // (whole blocks unless pcd_fe_set_residual_rows says otherwise)
  fe.res_bound = true;
  return 0;
} PCD_ABI_CATCH(pcd_fe_bind_residual)

// The constant blocks handed to pcd_fe_bind_residual hold THIS RANK'S ROWS
// only (global shape, the other rows empty - what a partitioned assembly
// holds: fenapack/SubfieldBC.h:136-155, _field_split_utils.py:39-50): B^T x_p
// and B v of the residual are then completed by one all-reduce each.
int pcd_fe_set_residual_rows(pcd_handle h, int on) try {
  if (!h || !h->fe || !h->fe->res_bound) return fail(PCD_ERR_STATE, "fe_set_residual_rows: bind the residual first");
  if (on && !h->comm) return fail(PCD_ERR_STATE, "fe_set_residual_rows: no communicator attached");
  h->fe->res_rows = on != 0;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_residual_rows)

// boundary values of the Dirichlet dofs (order of bc_idx); time dependent
int pcd_fe_set_bc_values(pcd_handle h, const double* g) try {
  if (!h || !h->fe || !h->fe->res_bound) return fail(PCD_ERR_STATE, "fe_set_bc_values: bind the residual first");
  if (!g && h->fe->n_bc) return fail(PCD_ERR_ARG, "fe_set_bc_values: null values");
  HIPCHK(hipSetDevice(h->device));
  if (h->fe->n_bc)
    HIPCHK(hipMemcpyAsync(h->fe->bc_g.p, g, h->fe->n_bc * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_bc_values)

// previous time level u0 (velocity dofs): the residual subtracts idt M u0
// (demo_unsteady-navier-stokes-pcd.py:104-120); NULL drops the term
int pcd_fe_set_previous(pcd_handle h, const double* u0, int mem) try {
  if (!h || !h->fe || !h->fe->res_bound) return fail(PCD_ERR_STATE, "fe_set_previous: bind the residual first");
  FeState& fe = *h->fe;
  if (!u0) { fe.have_mu0 = false; return 0; }
  if (fe.idt == 0.0 || !fe.mass.p) return fail(PCD_ERR_STATE, "fe_set_previous: no mass term was bound");
  HIPCHK(hipSetDevice(h->device));
  const int64_t nu = h->n_u;
  CHK(fe.u0.ensure(nu)); CHK(fe.mu0.ensure(nu));
  HIPCHK(hipMemcpyAsync(fe.u0.p, u0, nu * sizeof(double),
                        mem == PCD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  DCsr& A = h->mat[PCD_MAT_A00];
  if (h->comm) {
    CHK(fe_apply_global(h, fe, fe.mass.p, fe.u0.p, fe.mu0.p));
  } else if (fe.newton) {
    CHK(fe_apply_blocks(h, fe, fe.mass.p, nullptr, fe.u0.p, fe.mu0.p));
  } else {
    if (!A.set || A.kron != fe.dim || A.nnz2 != fe.lev[fe.nlev - 1].nnzf || !kron_ok(A, fe.u0.p, fe.mu0.p))
      return fail(PCD_ERR_STATE, "fe_set_previous: A00 is not F x I_%d on the FE pattern", fe.dim);
    CHK(spmv_other_values(h, A, fe.mass.p, fe.u0.p, fe.mu0.p));   // (M x I) u0 on the pattern of F
  }
  hipLaunchKernelGGL(k_axpby, dim3(grid1d(nu, 4)), dim3(kBlock), 0, h->stream, (int)nu, fe.idt, fe.mu0.p, 0.0, fe.mu0.p);
  HIPCHK(hipStreamSynchronize(h->stream));
  fe.have_mu0 = true;
  return 0;
} PCD_ABI_CATCH(pcd_fe_set_previous)

// Refresh the operators at the iterate x (caller's mixed numbering) and
// return the nonlinear residual b (same numbering) and its 2-norm.
int pcd_fe_residual(pcd_handle h, const double* x, double* b, int mem, double* norm) try {
  if (!h || !h->fe || !h->fe->res_bound) return fail(PCD_ERR_STATE, "fe_residual: bind the residual first");
  if (!x || !b || !norm) return fail(PCD_ERR_ARG, "fe_residual: null argument");
  FeState& fe = *h->fe;
  HIPCHK(hipSetDevice(h->device));
  const int64_t n = h->n_u + h->n_p;
  const double* dx = x; double* db = b;
  if (mem == PCD_MEM_HOST) {
    CHK(fe.xd.ensure(n)); CHK(fe.bd.ensure(n));
    HIPCHK(hipMemcpyAsync(fe.xd.p, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    dx = fe.xd.p; db = fe.bd.p;
  }
  CHK(fe_residual_dev(h, fe, dx, db, norm));
  if (mem == PCD_MEM_HOST) {
    HIPCHK(hipMemcpyAsync(b, db, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
} PCD_ABI_CATCH(pcd_fe_residual)

// The whole Picard iteration on the device (the loop of
// fenapack/nonlinear_solvers.py:28-82 around dolfin::NewtonSolver [ext]):
//   repeat: b = F(x) (operators refreshed), stop on |b| < atol or |b|/r0 < rtol,
//           GMRES: J dx = b, x -= relax dx.
// r0 <= 0: the first residual of this call is the reference norm.  lin_its /
// res_hist receive the GMRES count of every step and every residual norm
// (max_it and max_it + 1 entries).  x is updated in place.
int pcd_fe_picard_solve(pcd_handle h, double* x, int mem, double r0, double rtol,
                        double atol, int max_it, double relax, double lin_rtol,
                        double lin_atol, int restart, int lin_max_it, int* n_it,
                        int* lin_its, double* res_hist, int* converged) try {
  if (!h || !h->fe || !h->fe->res_bound) return fail(PCD_ERR_STATE, "fe_picard_solve: bind the residual first");
  if (!x || !n_it || !converged || max_it < 0 || (max_it && (!lin_its || !res_hist)))
    return fail(PCD_ERR_ARG, "fe_picard_solve: bad arguments");
  FeState& fe = *h->fe;
  HIPCHK(hipSetDevice(h->device));
  const int64_t n = h->n_u + h->n_p;
  CHK(fe.bd.ensure(n)); CHK(fe.dxd.ensure(n));
  double* xd = x;
  if (mem == PCD_MEM_HOST) {
    CHK(fe.xd.ensure(n));
    HIPCHK(hipMemcpyAsync(fe.xd.p, x, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    xd = fe.xd.p;
  }
  int it = 0;
  double r = 0.0;
  CHK(fe_residual_dev(h, fe, xd, fe.bd.p, &r));
  if (res_hist) res_hist[0] = r;
  if (!(r0 > 0.0)) r0 = r;
  // a zero first residual (r0 == 0) is a solved problem, not 0/0
  bool conv = r < atol || r == 0.0 || (r0 > 0.0 && r / r0 < rtol);
  while (!conv && it < max_it) {
    HIPCHK(hipMemsetAsync(fe.dxd.p, 0, n * sizeof(double), h->stream));
    int its = 0; double rn = 0.0;
    if (h->comm) {
      // the linear solve is partitioned: hand GMRES this rank's rows of the
      // replicated right-hand side, sum the slices of dx back into a replica
      const int64_t nl = h->nu_loc + h->np_loc;
      CHK(fe.bloc.ensure(nl)); CHK(fe.dxloc.ensure(nl));
      hipLaunchKernelGGL(k_gather, dim3(grid1d(nl, 1)), dim3(kBlock), 0, h->stream, (int)nl, h->perm.p, fe.bd.p, fe.bloc.p);
      CHK(pcd_gmres_solve(h, fe.bloc.p, fe.dxloc.p, PCD_MEM_DEVICE, lin_rtol, lin_atol, restart, lin_max_it, &its, &rn));
      hipLaunchKernelGGL(k_scatter, dim3(grid1d(nl, 1)), dim3(kBlock), 0, h->stream, (int)nl, h->perm.p, fe.dxloc.p, fe.dxd.p);
      if (h->comm->allreduce(fe.dxd.p, n, h->stream))
        return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
    } else {
      CHK(pcd_gmres_solve(h, fe.bd.p, fe.dxd.p, PCD_MEM_DEVICE, lin_rtol, lin_atol, restart, lin_max_it, &its, &rn));
    }
    lin_its[it] = its;
    hipLaunchKernelGGL(k_axpby, dim3(grid1d(n, 4)), dim3(kBlock), 0, h->stream, (int)n, -relax, fe.dxd.p, 1.0, xd);
    ++it;
    CHK(fe_residual_dev(h, fe, xd, fe.bd.p, &r));
    res_hist[it] = r;
    if (!std::isfinite(r))
      return fail(PCD_ERR_BREAKDOWN, "fe_picard_solve: non-finite residual norm after step %d", it);
    conv = r < atol || r == 0.0 || r / r0 < rtol;
  }
  if (mem == PCD_MEM_HOST)
    HIPCHK(hipMemcpyAsync(x, xd, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *n_it = it; *converged = conv ? 1 : 0;
  return 0;
} PCD_ABI_CATCH(pcd_fe_picard_solve)

// scalar operator of one level as last assembled (host array of nnz_f values)
int pcd_fe_get_level_values(pcd_handle h, int level, double* out) try {
  if (!h || !h->fe) return fail(PCD_ERR_STATE, "fe_get_level_values: call pcd_fe_begin first");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev || !fe.lev[level].set || !out)
    return fail(PCD_ERR_ARG, "fe_get_level_values: bad level / null output");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(out, fe.lev[level].F.p, fe.lev[level].nnzf * sizeof(double),
                        hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_fe_get_level_values)

// the d*d scalar matrices of the Newton term on one level as last assembled
// (host array of d*d*nnz_f values, [(i*d+j)][k]; Dirichlet rows/columns zero)
int pcd_fe_get_newton_values(pcd_handle h, int level, double* out) try {
  if (!h || !h->fe || !h->fe->newton) return fail(PCD_ERR_STATE, "fe_get_newton_values: pcd_fe_set_newton was not called");
  FeState& fe = *h->fe;
  if (level < 0 || level >= fe.nlev || !fe.lev[level].N.p || !out)
    return fail(PCD_ERR_ARG, "fe_get_newton_values: bad level / null output");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(out, fe.lev[level].N.p, (size_t)fe.dim * fe.dim * fe.lev[level].nnzf * sizeof(double),
                        hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_fe_get_newton_values)

int pcd_fe_get_kp_values(pcd_handle h, double* out) try {
  if (!h || !h->fe || !h->fe->kp_bound || !out)
    return fail(PCD_ERR_STATE, "fe_get_kp_values: Kp is not bound");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(out, h->fe->kp_vals.p + h->fe->kp_off, h->fe->nnz_kp * sizeof(double),
                        hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_fe_get_kp_values)

// smoother bounds the producer installed on a multigrid level (diagnostics)
int pcd_fe_get_bounds(pcd_handle h, int level, double* emin, double* emax) try {
  if (!h || !h->fe || h->fe->mg_slot < 0) return fail(PCD_ERR_STATE, "fe_get_bounds: no multigrid bound");
  Inner& s = h->inner[h->fe->mg_slot];
  if (level < 1 || level >= (int)s.mg.size() || !emin || !emax)
    return fail(PCD_ERR_ARG, "fe_get_bounds: bad level");
  *emin = s.mg[level].emin; *emax = s.mg[level].emax;
  return 0;
} PCD_ABI_CATCH(pcd_fe_get_bounds)

}  // extern "C"
