// pcd_apply.hip - launches of the apply path, inner solvers, multigrid cycle, apply bodies
// (one of the engine's translation units; shared declarations: pcd_internal.hpp)
#include "pcd_internal.hpp"

// --------------------------------------------------------------- launches
#define LAUNCH_LPR(A, KERNEL, GRID, ...)                                        \
  do {                                                                          \
    switch ((A).lpr) {                                                          \
      case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(GRID), dim3(kBlock), 0,      \
                                 h->stream, __VA_ARGS__); break;                \
      case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(GRID), dim3(kBlock), 0,      \
                                 h->stream, __VA_ARGS__); break;                \
      case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(GRID), dim3(kBlock), 0,    \
                                  h->stream, __VA_ARGS__); break;               \
      default: hipLaunchKernelGGL((KERNEL<32>), dim3(GRID), dim3(kBlock), 0,    \
                                  h->stream, __VA_ARGS__); break;               \
    }                                                                           \
  } while (0)

#define LAUNCH_RB(A, KERNEL, GRID, ...)                                         \
  do {                                                                          \
    switch ((A).rb) {                                                           \
      case 256: hipLaunchKernelGGL((KERNEL<256>), dim3(GRID), dim3(kBlock), 0,  \
                                   h->stream, __VA_ARGS__); break;              \
      case 128: hipLaunchKernelGGL((KERNEL<128>), dim3(GRID), dim3(kBlock), 0,  \
                                   h->stream, __VA_ARGS__); break;              \
      case 64: hipLaunchKernelGGL((KERNEL<64>), dim3(GRID), dim3(kBlock), 0,    \
                                  h->stream, __VA_ARGS__); break;               \
      default: hipLaunchKernelGGL((KERNEL<32>), dim3(GRID), dim3(kBlock), 0,    \
                                  h->stream, __VA_ARGS__); break;               \
    }                                                                           \
  } while (0)

static inline XVec xvec(const DCsr& A, const double* x) {
  return XVec{x, A.ghost.p, (int)A.ncols};
}

// Neighbour halo exchange of an SpMV input vector (multi-GPU): pack the owned
// entries other ranks read, grouped send/recv into the ghost buffer.  Every
// rank calls it for every SpMV (the threaded test backend synchronises there).
void halo_collect(Engine* h, const DCsr& A, const double* x,
                         std::vector<Msg>& sends, std::vector<Msg>& recvs) {
  const HaloPlan& pl = A.plan;
  const int ns = (int)pl.send_idx.size();
  if (ns)
    hipLaunchKernelGGL(k_pack, dim3(grid1d(ns, 1)), dim3(kBlock), 0, h->stream,
                       ns, A.send_idx.p, x, A.sendbuf.p);
  for (size_t i = 0; i < pl.peers_send.size(); ++i)
    sends.push_back(Msg{pl.peers_send[i], A.sendbuf.p + pl.send_off[i],
                        (size_t)(pl.send_off[i + 1] - pl.send_off[i])});
  for (size_t i = 0; i < pl.peers_recv.size(); ++i)
    recvs.push_back(Msg{pl.peers_recv[i], A.ghost.p + pl.recv_off[i],
                        (size_t)(pl.recv_off[i + 1] - pl.recv_off[i])});
}
int halo_exchange(Engine* h, const DCsr& A, const double* x) {
  if (!h->comm || A.replicated) return 0;
  if (A.ph.ready && static_cast<PeerBackend*>(h->comm)->usable(h->stream)) {
    // one kernel: pack, remote store, signal, wait, land
    PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
    if (pb->halo(A.ph, x, h->stream)) return fail(PCD_ERR_COMM, "halo exchange: %s", pb->err.c_str());
    return 0;
  }
  h->boot_exchanges++;
  std::vector<Msg> sends, recvs;
  halo_collect(h, A, x, sends, recvs);
  if (h->comm->exchange(sends, recvs, h->stream))
    return fail(PCD_ERR_COMM, "halo exchange: %s", h->comm->err.c_str());
  return 0;
}
// SpMV overlaps the halo transfer with the rows that need no ghost column
// (what PETSc's MPIAIJ MatMult does with its diag / offd blocks
// under the reference's mpirun -np 3, test/regression/test.py:186-190).  With
// the peer protocol the exchange is one ~4 us kernel; split, it is a send
// kernel, the interior blocks, a wait-and-land kernel, the boundary blocks:
// two launches more per SpMV, the neighbours' latency hidden behind the
// interior blocks.  Only real peers can price it (two processes on one GPU
// time-share it), so it sits behind a switch: PCD_OVERLAP=1.  Blocks are
// computed exactly as without the split: bitwise the same result.
static int g_overlap = [] { const char* e = getenv("PCD_OVERLAP"); return e ? atoi(e) : 0; }();
bool overlap_ok(Engine* h, const DCsr& A) {
  return g_overlap && h->comm && !A.replicated && A.vt && A.vt_nbnd > 0 && A.ph.ready &&
         static_cast<PeerBackend*>(h->comm)->usable(h->stream);
}
int halo_send(Engine* h, const DCsr& A, const double* x) {
  PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
  if (pb->halo_send(A.ph, x, h->stream)) return fail(PCD_ERR_COMM, "halo exchange: %s", pb->err.c_str());
  return 0;
}
int halo_wait(Engine* h, const DCsr& A) {
  PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
  if (pb->halo_wait(A.ph, h->stream)) return fail(PCD_ERR_COMM, "halo exchange: %s", pb->err.c_str());
  return 0;
}
static inline VtBlocks vt_blocks_now(const Engine* h, const DCsr& A) {
  if (h->ov_phase == 1) return VtBlocks{A.vt_nint, A.vt_list.p};
  if (h->ov_phase == 2) return VtBlocks{A.vt_nbnd, A.vt_list.p + A.vt_nint};
  return VtBlocks{A.vt_blocks, nullptr};
}

int halo_exchange_group(Engine* h, std::initializer_list<HaloItem> items) {
  if (!h->comm) return 0;
  std::vector<Msg> sends, recvs;
  bool any = false;
  for (const HaloItem& it : items) {
    if (!it.A->set || it.A->replicated) continue;
    if (it.A->ph.ready && static_cast<PeerBackend*>(h->comm)->usable(h->stream)) {
      CHK(halo_exchange(h, *it.A, it.x));
      continue;
    }
    halo_collect(h, *it.A, it.x, sends, recvs);
    any = true;
  }
  if (!any) return 0;
  h->boot_exchanges++;
  if (h->comm->exchange(sends, recvs, h->stream))
    return fail(PCD_ERR_COMM, "halo exchange: %s", h->comm->err.c_str());
  return 0;
}

int reduce_global(Engine* h, double* parts, int nparts, double* slot,
                         PartsRef* out) {
  if (!h->comm) { out->p = parts; out->n = nparts; return 0; }
  hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(kBlock), 0, h->stream, parts,
                     nparts, 0, slot);
  if (h->comm->allreduce(slot, 1, h->stream))
    return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  out->p = slot; out->n = 1;
  return 0;
}

#define LAUNCH_RBC__(NC, NT, A, KERNEL, GRID, ...)                              \
  do {                                                                          \
    switch ((A).rb2) {                                                          \
      case 256: hipLaunchKernelGGL((KERNEL<256, NC, NT>), dim3(GRID), dim3(kBlock), \
                                   0, h->stream, __VA_ARGS__); break;           \
      case 128: hipLaunchKernelGGL((KERNEL<128, NC, NT>), dim3(GRID), dim3(kBlock), \
                                   0, h->stream, __VA_ARGS__); break;           \
      case 64: hipLaunchKernelGGL((KERNEL<64, NC, NT>), dim3(GRID), dim3(kBlock),   \
                                  0, h->stream, __VA_ARGS__); break;            \
      default: hipLaunchKernelGGL((KERNEL<32, NC, NT>), dim3(GRID), dim3(kBlock),   \
                                  0, h->stream, __VA_ARGS__); break;            \
    }                                                                           \
  } while (0)
#define LAUNCH_RBC_(NC, A, KERNEL, GRID, ...)                                   \
  do {                                                                          \
    if ((A).nt2) LAUNCH_RBC__(NC, true, A, KERNEL, GRID, __VA_ARGS__);          \
    else LAUNCH_RBC__(NC, false, A, KERNEL, GRID, __VA_ARGS__);                 \
  } while (0)
#define LAUNCH_RBC(A, KERNEL, GRID, ...)                                        \
  do {                                                                          \
    if ((A).kron == 2) LAUNCH_RBC_(2, A, KERNEL, GRID, __VA_ARGS__);            \
    else LAUNCH_RBC_(3, A, KERNEL, GRID, __VA_ARGS__);                          \
  } while (0)


// multi-component operator: F streamed once, all components of a node together
// (`ghost` / `ncols`: the second segment of the gathered vector and where it
// starts - the operator's own halo buffer, or the second piece of a two-piece
// input)
template <int MODE, int NC>
static void launch_spmv_kron_nc(Engine* h, const DCsr& A, const double* x,
                                const double* add, double* y,
                                const double* ghost, int64_t ncols) {
  const int nn = (int)(A.nrows / NC);
  const int nloc = (int)(ncols / NC);
  if (A.long_rows) {
    hipLaunchKernelGGL((k_spmv_longc<MODE, NC>), dim3(std::min(nn, 65535)), dim3(kBlock), 0, h->stream,
                       nn, A.rowptr2.p, A.col2.p, A.val2.p, x, ghost, nloc, add, y);
    return;
  }
  if (A.wave_rows) {
    const int gw = (int)std::min<int64_t>((nn + 3) / 4, 1 << 16);
    hipLaunchKernelGGL((k_spmv_wc<MODE, NC>), dim3(gw), dim3(kBlock), 0, h->stream,
                       nn, A.rowptr2.p, A.col2.p, A.val2.p, x, ghost, nloc, add, y);
    return;
  }
  const VtBlocks vb = vt_blocks_now(h, A);
  if (A.vt && vb.n == 0) return;
  if (A.vt && A.vt_lm) {
    hipLaunchKernelGGL((k_spmv_lm<MODE, NC>), dim3(grid_stream(vb.n, 1)), dim3(kBlock), 0, h->stream,
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.vt_val.p,
                       A.vt_loc.p, x, ghost, nloc, add, y);
    return;
  }
  if (A.vt) {
    const int gt = grid_stream(vb.n, 1);
    // (ROWS = 16: operators with long rows, 16 lanes per row - build_vec_tile)
    if (A.vt_rows == 16)
      hipLaunchKernelGGL((k_spmv_tc<MODE, NC, 16>), dim3(gt), dim3(kBlock), 0, h->stream,
                         vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.val2.p,
                         A.vt_loc.p, x, ghost, nloc, add, y);
    else
    hipLaunchKernelGGL((k_spmv_tc<MODE, NC, 64>), dim3(gt), dim3(kBlock), 0, h->stream,
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.val2.p,
                       A.vt_loc.p, x, ghost, nloc, add, y);
    return;
  }
  const int g = grid_stream(nn, A.rb2);
#define PCD_SPMV_SC(RB, NT)                                                               \
  hipLaunchKernelGGL((k_spmv_sc<RB, MODE, NC, NT>), dim3(g), dim3(kBlock), 0, h->stream, \
                     nn, A.rowptr2.p, A.col2.p, A.val2.p, x, ghost, nloc, add, y)
  if (A.nt2) {
    switch (A.rb2) {
      case 256: PCD_SPMV_SC(256, true); break;
      case 128: PCD_SPMV_SC(128, true); break;
      case 64: PCD_SPMV_SC(64, true); break;
      default: PCD_SPMV_SC(32, true); break;
    }
  } else {
    switch (A.rb2) {
      case 256: PCD_SPMV_SC(256, false); break;
      case 128: PCD_SPMV_SC(128, false); break;
      case 64: PCD_SPMV_SC(64, false); break;
      default: PCD_SPMV_SC(32, false); break;
    }
  }
#undef PCD_SPMV_SC
}

template <int MODE>
static void launch_spmv_any(Engine* h, const DCsr& A, const double* x,
                            const double* add, double* y, const double* ghost,
                            int64_t ncols, bool kron) {
  const XVec xv{x, ghost, (int)ncols};
  if (kron && A.dense2 && ghost == A.ghost.p) {
    const int nn = (int)(A.nrows / A.kron), mm = (int)(A.ncols / A.kron);
    const int g = std::min(nn, 65535);
    if (A.kron == 2) hipLaunchKernelGGL((k_dense_c<MODE, 2>), dim3(g), dim3(kBlock), 0, h->stream, nn, mm, A.val2.p, x, add, y);
    else hipLaunchKernelGGL((k_dense_c<MODE, 3>), dim3(g), dim3(kBlock), 0, h->stream, nn, mm, A.val2.p, x, add, y);
  } else if (kron) {
    if (A.kron == 2) launch_spmv_kron_nc<MODE, 2>(h, A, x, add, y, ghost, ncols);
    else launch_spmv_kron_nc<MODE, 3>(h, A, x, add, y, ghost, ncols);
  } else if (A.rk && A.rk_rb && (A.rk == 3 || (aligned16(y) && aligned16(add)))) {
    // the rows of a node share their columns: one index per node-entry
    const int nn = (int)(A.nrows / A.rk);
    const int g = grid_stream(nn, A.rk_rb);
#define PCD_SPMV_RK_(RB, NC, NT)                                                          \
    hipLaunchKernelGGL((k_spmv_rk<RB, MODE, NC, NT>), dim3(g), dim3(kBlock), 0, h->stream, \
                       nn, A.rk_rowptr.p, A.rk_col.p, A.rk_val.p, xv, add, y)
#define PCD_SPMV_RK(NC, NT)                                                               \
    switch (A.rk_rb) {                                                                    \
      case 256: PCD_SPMV_RK_(256, NC, NT); break;                                         \
      case 128: PCD_SPMV_RK_(128, NC, NT); break;                                         \
      case 64: PCD_SPMV_RK_(64, NC, NT); break;                                           \
      default: PCD_SPMV_RK_(32, NC, NT); break;                                           \
    }
    if (A.rk == 2) { if (A.rk_nt) { PCD_SPMV_RK(2, true) } else { PCD_SPMV_RK(2, false) } }
    else { if (A.rk_nt) { PCD_SPMV_RK(3, true) } else { PCD_SPMV_RK(3, false) } }
#undef PCD_SPMV_RK
#undef PCD_SPMV_RK_
  } else if (A.dense && ghost == A.ghost.p) {
    const int g = (int)std::min<int64_t>(A.nrows, 65535);
    hipLaunchKernelGGL((k_dense_c<MODE, 1>), dim3(g), dim3(kBlock), 0, h->stream,
                       (int)A.nrows, (int)A.ncols, A.val.p, x, add, y);
  } else if (A.long_rows) {
    const int g = (int)std::min<int64_t>(A.nrows, 65535);
    hipLaunchKernelGGL((k_spmv_long<MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                       (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y);
  } else if (A.wave_rows) {
    const int g = (int)std::min<int64_t>((A.nrows + 3) / 4, 1 << 16);
    hipLaunchKernelGGL((k_spmv_w<MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                       (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y);
  } else if (A.rb && A.small_tile) {
    const int g = grid_stream(A.nrows, A.rb);
    switch (A.rb) {
      case 256: hipLaunchKernelGGL((k_spmv_s<256, MODE, kTileSmall>), dim3(g), dim3(kBlock), 0, h->stream,
                                   (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      case 128: hipLaunchKernelGGL((k_spmv_s<128, MODE, kTileSmall>), dim3(g), dim3(kBlock), 0, h->stream,
                                   (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      case 64: hipLaunchKernelGGL((k_spmv_s<64, MODE, kTileSmall>), dim3(g), dim3(kBlock), 0, h->stream,
                                  (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      default: hipLaunchKernelGGL((k_spmv_s<32, MODE, kTileSmall>), dim3(g), dim3(kBlock), 0, h->stream,
                                  (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
    }
  } else if (A.rb) {
    const int g = grid_stream(A.nrows, A.rb);
    switch (A.rb) {
      case 256: hipLaunchKernelGGL((k_spmv_s<256, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                   (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      case 128: hipLaunchKernelGGL((k_spmv_s<128, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                   (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      case 64: hipLaunchKernelGGL((k_spmv_s<64, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                  (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      default: hipLaunchKernelGGL((k_spmv_s<32, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                  (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
    }
  } else {
    const int g = grid_rows(A.nrows, A.lpr);
    switch (A.lpr) {
      case 4: hipLaunchKernelGGL((k_spmv<4, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                 (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      case 8: hipLaunchKernelGGL((k_spmv<8, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                 (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      case 16: hipLaunchKernelGGL((k_spmv<16, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                  (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
      default: hipLaunchKernelGGL((k_spmv<32, MODE>), dim3(g), dim3(kBlock), 0, h->stream,
                                  (int)A.nrows, A.rowptr.p, A.col.p, A.val.p, xv, add, y); break;
    }
  }
}

// y = A x (mode 0) | add + A x (1) | add - A x (2) | -A x (3)
// `x2` (optional, operators WITHOUT a halo only): the input is the
// concatenation [x (n1 entries) | x2] - the ghost segment of the gather
// functor carries the second piece, so no extra kernel is needed.
int spmv(Engine* h, const DCsr& A, const double* x, double* y,
                int mode, const double* add,
                const double* x2, int64_t n1,
                bool halo_done) {
  if (!A.set) return fail(PCD_ERR_STATE, "spmv: operator not set");
  const double* ghost = A.ghost.p;
  int64_t ncols = A.ncols;
  if (x2) {
    if (A.plan.nghost || (h->comm && !A.replicated))
      return fail(PCD_ERR_STATE, "spmv: two-piece input on an operator with a halo");
    if (A.kron && n1 % A.kron) return fail(PCD_ERR_ARG, "spmv: piece boundary splits a node");
    ghost = x2; ncols = n1;
  }
  const bool kron = kron_ok(A, x, y, add, x2);
  auto run = [&]() {
    switch (mode) {
      case 0: launch_spmv_any<0>(h, A, x, add, y, ghost, ncols, kron); break;
      case 1: launch_spmv_any<1>(h, A, x, add, y, ghost, ncols, kron); break;
      case 2: launch_spmv_any<2>(h, A, x, add, y, ghost, ncols, kron); break;
      default: launch_spmv_any<3>(h, A, x, add, y, ghost, ncols, kron); break;
    }
  };
  if (!x2 && !halo_done && kron && overlap_ok(h, A)) {
    // interior blocks while the halo travels, boundary blocks after it landed
    CHK(halo_send(h, A, x));
    h->ov_phase = 1; run();
    h->ov_phase = 0;
    CHK(halo_wait(h, A));
    h->ov_phase = 2; run();
    h->ov_phase = 0;
  } else {
    if (!x2 && !halo_done) CHK(halo_exchange(h, A, x));
    run();
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// y = (G (x) I) x for another value array G on F's pattern (row-major, like
// val2); the lane-major copy of the tile kernels follows the values in force
int spmv_other_values(Engine* h, DCsr& A, double*& other, const double* x, double* y) {
  auto lane_major = [&]() {
    if (A.vt && A.vt_lm)
      hipLaunchKernelGGL(k_lm_values, dim3(grid1d(A.vt_slots, 4)), dim3(kBlock), 0, h->stream,
                         A.vt_slots, A.vt_pos.p, A.val2.p, A.vt_val.p);
  };
  std::swap(A.val2.p, other);
  lane_major();
  const int rc = spmv(h, A, x, y);
  std::swap(A.val2.p, other);
  lane_major();
  return rc;
}

// ---- ChebPatch: m Chebyshev-Jacobi steps of a small scalar operator in ONE
// launch (pcd_internal.hpp, k_cheb_patch).  PCD_CHEB_PATCH=0 switches it off;
// PCD_CHEB_PATCH_ROWS (default 2 000 000) is the largest operator it is built
// for; PCD_CHEB_PATCH_CLUSTER (512) rows per cluster.  Measured on the
// pressure mass matrix of the cavity (5 steps, same-box A/B, profiles/r05_u_*):
// level 6 (103 041 rows: 206 clusters, 2.0 x the rows in patches) 27 us in
// five launches -> 9.6 us, one PCApply 0.288 -> 0.273 ms; level 7 (410 881
// rows) 64 -> 26 us, 0.744 -> 0.702 ms - the matrix is read once (twice, with
// the redundancy) instead of five times.  In space the patches of five edges
// outgrow the workgroup and the step-by-step path stays.
// One rank only (a ghost column changes with every step).  Clusters: greedy
// breadth-first balls in the operator's own row order (pcd_reorder.hpp's
// cluster rule); the patch = the cluster and everything within m edges,
// ordered by distance.  Not built (the step-by-step path stays) when a patch
// exceeds the workgroup's LDS or the patches together exceed 3 x the rows.
int build_cheb_patch(Engine* h, DCsr& A, int m) {
  A.cp.release();
  // declined builds are remembered until the pattern or m changes (pcd_inner_solve
  // prepares on every call: the clustering of up to 2 M rows is not redone)
  A.cp.tried_m = m;
  // (read per build: the A/B tests of one process switch it between engines)
  const char* eo = getenv("PCD_CHEB_PATCH");
  const int on = eo ? atoi(eo) : 1;
  const char* er = getenv("PCD_CHEB_PATCH_ROWS");
  const int64_t max_rows = er ? atoll(er) : 2000000;
  const char* ec = getenv("PCD_CHEB_PATCH_CLUSTER");
  const int csize = std::max(32, ec ? atoi(ec) : 512);
  if (!on || h->comm || m < 2 || m > kChebPatchMaxM || !A.set || A.kron || A.dense || A.nrows != A.ncols ||
      A.nrows < 1 || A.nrows > max_rows || A.nnz < 1 || A.plan.nghost)
    return 0;
  const int64_t n = A.nrows;
  std::vector<int32_t> rp(n + 1), cc(A.nnz);
  HIPCHK(hipMemcpy(rp.data(), A.rowptr.p, (n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(cc.data(), A.col.p, (size_t)A.nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
  // symmetric adjacency (the operator's pattern may be one-sided at BC rows)
  std::vector<int64_t> ap(n + 1, 0);
  for (int64_t i = 0; i < n; ++i)
    for (int32_t q = rp[i]; q < rp[i + 1]; ++q)
      if (cc[q] != i) { ++ap[i + 1]; ++ap[cc[q] + 1]; }
  for (int64_t i = 0; i < n; ++i) ap[i + 1] += ap[i];
  std::vector<int32_t> adj(ap[n]);
  {
    std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int32_t q = rp[i]; q < rp[i + 1]; ++q)
        if (cc[q] != i) { adj[fill[i]++] = cc[q]; adj[fill[cc[q]]++] = (int32_t)i; }
  }
  // clusters
  std::vector<int32_t> order; order.reserve(n);
  std::vector<int64_t> cstart;
  {
    std::vector<char> seen(n, 0);
    for (int64_t s0 = 0; s0 < n; ++s0) {
      if (seen[s0]) continue;
      const size_t first = order.size();
      cstart.push_back((int64_t)first);
      order.push_back((int32_t)s0); seen[s0] = 1;
      for (size_t q = first; q < order.size() && order.size() - first < (size_t)csize; ++q) {
        const int32_t v = order[q];
        for (int64_t e = ap[v]; e < ap[v + 1] && order.size() - first < (size_t)csize; ++e)
          if (!seen[adj[e]]) { seen[adj[e]] = 1; order.push_back(adj[e]); }
      }
      std::sort(order.begin() + first, order.end());
    }
    cstart.push_back((int64_t)order.size());
  }
  const int nb = (int)cstart.size() - 1;
  // patches: breadth-first layers around every cluster
  std::vector<int32_t> node;
  std::vector<int4> desc(nb);
  std::vector<int> cnt((size_t)nb * (kChebPatchMaxM + 1), 0);
  std::vector<unsigned short> ecol;
  std::vector<int> epos;
  std::vector<int32_t> stamp(n, -1), local(n, 0);
  int64_t total = 0;
  int wmax = 0;
  for (int b = 0; b < nb; ++b) {
    const size_t off = node.size();
    for (int64_t q = cstart[b]; q < cstart[b + 1]; ++q) {
      stamp[order[q]] = b; local[order[q]] = (int32_t)(node.size() - off); node.push_back(order[q]);
    }
    int* c = &cnt[(size_t)b * (kChebPatchMaxM + 1)];
    c[0] = (int)(node.size() - off);
    size_t lo = off;
    for (int k = 1; k <= m; ++k) {
      const size_t hi = node.size();
      for (size_t q = lo; q < hi; ++q) {
        const int32_t v = node[q];
        for (int64_t e = ap[v]; e < ap[v + 1]; ++e) {
          const int32_t w = adj[e];
          if (stamp[w] != b) { stamp[w] = b; local[w] = (int32_t)(node.size() - off); node.push_back(w); }
        }
      }
      std::sort(node.begin() + hi, node.end());
      for (size_t q = hi; q < node.size(); ++q) local[node[q]] = (int32_t)(q - off);
      lo = hi;
      c[k] = (int)(node.size() - off);
    }
    for (int k = m + 1; k <= kChebPatchMaxM; ++k) c[k] = c[m];
    const int P = c[m];
    if (P > kChebPatchNodes) return 0;                        // does not fit the workgroup's LDS
    total += P;
    if (total > 3 * n + 4096) return 0;                       // too much redundant work
    // ELL rows of the nodes within m - 1 edges (their columns lie within m)
    const int R = c[m - 1], Rpad = (R + 63) / 64 * 64;
    int W = 0;
    for (int r = 0; r < R; ++r) { const int32_t g = node[off + r]; W = std::max(W, rp[g + 1] - rp[g]); }
    if (W > 16 || Rpad > 0xffff) return 0;                   // (the kernel keeps the rows in registers)
    wmax = std::max(wmax, W);
    const size_t ell = ecol.size();
    if (ell + (size_t)W * Rpad > (size_t)INT32_MAX) return 0;
    ecol.resize(ell + (size_t)W * Rpad, 0);
    epos.resize(ell + (size_t)W * Rpad, -1);
    for (int r = 0; r < R; ++r) {
      const int32_t g = node[off + r];
      int e = 0;
      for (int32_t q = rp[g]; q < rp[g + 1]; ++q, ++e) {
        if (stamp[cc[q]] != b) return fail(PCD_ERR_STATE, "cheb patch: column outside the patch");
        ecol[ell + (size_t)e * Rpad + r] = (unsigned short)local[cc[q]];
        epos[ell + (size_t)e * Rpad + r] = q;
      }
    }
    desc[b] = int4{(int)off, (int)ell, P, Rpad | (W << 16)};
  }
  if (node.size() > (size_t)INT32_MAX) return 0;
  ChebPatch& cp = A.cp;
  CHK(cp.node.ensure(node.size())); CHK(cp.desc.ensure(nb)); CHK(cp.cnt.ensure(cnt.size()));
  CHK(cp.col.ensure(ecol.size())); CHK(cp.val.ensure(ecol.size())); CHK(cp.pos.ensure(epos.size()));
  HIPCHK(hipMemcpy(cp.node.p, node.data(), node.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.desc.p, desc.data(), (size_t)nb * sizeof(int4), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.cnt.p, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.col.p, ecol.data(), ecol.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cp.pos.p, epos.data(), epos.size() * sizeof(int), hipMemcpyHostToDevice));
  cp.nslots = (int64_t)ecol.size(); cp.nblocks = nb; cp.m = m; cp.wmax = wmax;
  hipLaunchKernelGGL(k_lm_values, dim3(grid1d(cp.nslots, 4)), dim3(kBlock), 0, h->stream,
                     cp.nslots, cp.pos.p, A.val.p, cp.val.p);
  HIPCHK(hipGetLastError());
  cp.ready = true;
  if (const char* e = getenv("PCD_CHEB_PATCH_STATS")) if (e[0] == '1')
    fprintf(stderr, "[pcd cheb patch] %lld rows, m = %d: %d clusters, %.2f x the rows in patches, %lld ELL slots\n",
            (long long)n, m, nb, (double)total / (double)n, (long long)cp.nslots);
  return 0;
}

int refresh_dinv(Engine* h, DCsr& A) {
  CHK(refresh_kron(h, A));
  if (A.nrows != A.ncols) return 0;
  CHK(A.dinv.ensure(A.nrows));
  hipLaunchKernelGGL(k_dinv, dim3(grid1d(A.nrows, 1, 1 << 30)), dim3(kBlock), 0,
                     h->stream, (int)A.nrows, A.rowptr.p, A.col.p, A.val.p,
                     A.dinv.p);
  // column-scaled values for the fused zero-guess first step; with several
  // ranks the reciprocal diagonal of the ghost columns arrives like any halo
  // (collective: every rank refreshes every operator in the same order)
  // (the exchange is collective: a rank that owns no rows of a partitioned
  // level - pcd_mg_set_level_cuts accepts empty blocks - still takes part;
  // only the launches on its own entries are skipped)
  if (h->comm && !A.replicated && !A.nnz) CHK(halo_exchange(h, A, A.dinv.p));
  if (A.nnz) {
    CHK(halo_exchange(h, A, A.dinv.p));
    if (A.kron && A.nnz2 && A.vt) {
      // tile kernels scale the gathered TILE instead (k_cheb_first_tc): no
      // second copy of the values; the ghost columns' reciprocal diagonal is
      // kept aside (the ghost buffer itself is every later exchange's)
      A.val2s.release();
      if (A.plan.nghost && h->comm && !A.replicated) {
        CHK(A.dghost.ensure(A.plan.nghost));
        HIPCHK(hipMemcpyAsync(A.dghost.p, A.ghost.p, (size_t)A.plan.nghost * sizeof(double),
                              hipMemcpyDeviceToDevice, h->stream));
      }
    } else if (A.kron && A.nnz2) {
      CHK(A.val2s.ensure(A.nnz2 + 2));
      hipLaunchKernelGGL(k_scale_cols, dim3(grid1d(A.nnz2, 4)), dim3(kBlock), 0, h->stream,
                         A.nnz2, A.col2.p, A.val2.p, A.dinv.p, A.kron, A.val2s.p,
                         A.ghost.p, (int)(A.ncols / A.kron));
    }
    if (A.rb) {
      CHK(A.vals.ensure(A.nnz));
      hipLaunchKernelGGL(k_scale_cols, dim3(grid1d(A.nnz, 4)), dim3(kBlock), 0, h->stream,
                         A.nnz, A.col.p, A.val.p, A.dinv.p, 1, A.vals.p, A.ghost.p, (int)A.ncols);
    }
  }
  if (A.cp.ready)                       // the patch copies follow the values in force
    hipLaunchKernelGGL(k_lm_values, dim3(grid1d(A.cp.nslots, 4)), dim3(kBlock), 0, h->stream,
                       A.cp.nslots, A.cp.pos.p, A.val.p, A.cp.val.p);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------ inner KSPs
int inner_prepare(Engine* h, int slot) {
  Inner& s = h->inner[slot];
  const DCsr& A = h->mat[kSlotMat[slot]];
  if (!A.set) return 0;
  const size_t n = A.nrows;
  if (s.pc == PCD_PC_MG) {
    CHK(s.t0.ensure(n));
    for (size_t l = 0; l < s.mg.size(); ++l) {
      MgLevel& M = s.mg[l];
      const size_t nl = (l + 1 == s.mg.size()) ? n
                        : (M.A.set ? (size_t)M.A.nrows : 0);
      if (!nl) continue;
      CHK(M.x.ensure(nl)); CHK(M.b.ensure(nl));
      if (l > 0) { CHK(M.t0.ensure(nl)); CHK(M.t1.ensure(nl)); CHK(M.r.ensure(nl)); }
      if (l > 0 && M.fused) CHK(M.T.ensure(nl + 2 * (size_t)M.P.ncols));
    }
    return 0;
  }
  if (s.pc == PCD_PC_EXPLICIT) {
    CHK(s.t0.ensure(n)); CHK(s.t1.ensure(n));
    return 0;
  }
  switch (s.ksp) {
    case PCD_KSP_CG:
      CHK(s.t0.ensure(n)); CHK(s.t1.ensure(n)); CHK(s.t2.ensure(n));
      CHK(s.t3.ensure(n)); CHK(s.t4.ensure(n)); CHK(s.parts.ensure(3 * kMaxParts));
      CHK(s.slots.ensure(4));
      CHK(s.state.ensure(2));
      break;
    case PCD_KSP_CG_SR:
      CHK(s.t0.ensure(n)); CHK(s.t1.ensure(n)); CHK(s.t2.ensure(n));
      CHK(s.t3.ensure(n)); CHK(s.t4.ensure(n)); CHK(s.parts.ensure(3 * kMaxParts));
      CHK(s.slots.ensure(4));
      CHK(s.state.ensure(2));
      break;
    case PCD_KSP_CHEBYSHEV:
      CHK(s.t0.ensure(n)); CHK(s.t1.ensure(n));
      // (all its steps in one launch where the operator is small enough)
      if (s.pc == PCD_PC_JACOBI) {
        const ChebPatch& cp = h->mat[kSlotMat[slot]].cp;
        const bool have = cp.ready && cp.m == s.max_it;
        const bool declined = !cp.ready && cp.tried_m == s.max_it;
        if (!have && !declined) CHK(build_cheb_patch(h, h->mat[kSlotMat[slot]], s.max_it));
      }
      break;
    case PCD_KSP_RICHARDSON:
      CHK(s.t0.ensure(n));
      break;
    default: break;
  }
  return 0;
}

int launch_cheb_step(Engine* h, const DCsr& A, const double* dinv,
                            const double* b, const double* pm, const double* pk,
                            double* pn, double c0, double c1, double c2) {
  const int n = (int)A.nrows;
  const bool probe = h->probe_on && &A == &h->mat[PCD_MAT_A00];
  struct Probe {                 // (every return path below records the closing event)
    Engine* h; bool on;
    Probe(Engine* h_, bool on_) : h(h_), on(on_) { if (on) mark(); }
    ~Probe() { if (on) mark(); }
    void mark() {
      hipEvent_t e = nullptr;
      if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, h->stream) == hipSuccess) h->probe_ev.push_back(e);
    }
  } probe_guard(h, probe);
  const bool tile = dinv && A.vt && kron_ok(A, b, pm, pk, pn, true);
  auto run = [&]() {
  if (tile && A.vt_lm) {
    const VtBlocks vb = vt_blocks_now(h, A);
    if (!vb.n) return;
    const int gt = grid_stream(vb.n, 1);
    const int nloc = (int)(A.ncols / A.kron);
#define PCD_CHEB_LM(NC)                                                                        \
    hipLaunchKernelGGL((k_cheb_step_lm<NC>), dim3(gt), dim3(kBlock), 0, h->stream,            \
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.vt_val.p,     \
                       A.vt_loc.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p, nloc)
    if (A.kron == 2) PCD_CHEB_LM(2); else PCD_CHEB_LM(3);
#undef PCD_CHEB_LM
  } else if (tile) {
    const VtBlocks vb = vt_blocks_now(h, A);
    if (!vb.n) return;
    const int gt = grid_stream(vb.n, 1);
    const int nloc = (int)(A.ncols / A.kron);
#define PCD_CHEB_TC(NC, RW)                                                                    \
    hipLaunchKernelGGL((k_cheb_step_tc<NC, RW>), dim3(gt), dim3(kBlock), 0, h->stream,        \
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.val2.p,       \
                       A.vt_loc.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p, nloc)
    if (A.vt_rows == 16) { if (A.kron == 2) PCD_CHEB_TC(2, 16); else PCD_CHEB_TC(3, 16); }
    else if (A.kron == 2) PCD_CHEB_TC(2, 64); else PCD_CHEB_TC(3, 64);
#undef PCD_CHEB_TC
  } else if (dinv && A.rb2 && kron_ok(A, b, pm, pk, pn, true)) {
    const int nn = n / A.kron;
    LAUNCH_RBC(A, k_cheb_step_sc, grid_stream(nn, A.rb2), nn, A.rowptr2.p, A.col2.p,
               A.val2.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p,
               (int)(A.ncols / A.kron));
  } else if (A.rb) {
    LAUNCH_RB(A, k_cheb_step_s, grid_stream(n, A.rb), n, A.rowptr.p, A.col.p,
              A.val.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p, (int)A.ncols);
  } else {
    LAUNCH_LPR(A, k_cheb_step, grid_rows(n, A.lpr), n, A.rowptr.p, A.col.p,
               A.val.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p, (int)A.ncols);
  }
  };
  if (tile && overlap_ok(h, A)) {
    CHK(halo_send(h, A, pk));
    h->ov_phase = 1; run();
    h->ov_phase = 0;
    CHK(halo_wait(h, A));
    h->ov_phase = 2; run();
    h->ov_phase = 0;
  } else {
    CHK(halo_exchange(h, A, pk));
    run();
  }
  return 0;
}

// zero-guess start fused with the first step (single GPU, stream kernels):
// p0 = s D^-1 b (also written to `p0` unless null), pn = c1 p0 + c2 D^-1(b - A p0)
bool can_fuse_first(const Engine* h, const DCsr& A, const double* dinv) {
  (void)h;
  // (a multi-component operator none of whose kernels takes the step falls
  // through to the scalar stream kernel and its column-scaled values)
  return A.rb && dinv != nullptr;
}
int launch_cheb_first(Engine* h, const DCsr& A, const double* dinv,
                             const double* b, double* p0, double* pn, double s,
                             double c1, double c2) {
  const int n = (int)A.nrows;
  const double* ghost = (h->comm && !A.replicated) ? A.ghost.p : b;
  const bool tile = A.vt && kron_ok(A, b, p0, pn, nullptr, true);
  auto run = [&]() {
  if (tile && A.vt_lm) {
    const VtBlocks vb = vt_blocks_now(h, A);
    if (!vb.n) return;
    const int gt = grid_stream(vb.n, 1);
#define PCD_FIRST_LM(NC)                                                                       \
    hipLaunchKernelGGL((k_cheb_first_lm<NC>), dim3(gt), dim3(kBlock), 0, h->stream,           \
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.vt_val.p,     \
                       A.vt_loc.p, dinv, b, p0, pn, s, c1, c2, ghost, (int)(A.ncols / A.kron), \
                       A.dghost.p ? A.dghost.p : dinv)
    if (A.kron == 2) PCD_FIRST_LM(2); else PCD_FIRST_LM(3);
#undef PCD_FIRST_LM
  } else if (tile) {
    const VtBlocks vb = vt_blocks_now(h, A);
    if (!vb.n) return;
    const int gt = grid_stream(vb.n, 1);
#define PCD_FIRST_TC(NC, RW)                                                                   \
    hipLaunchKernelGGL((k_cheb_first_tc<NC, RW>), dim3(gt), dim3(kBlock), 0, h->stream,       \
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.val2.p,       \
                       A.vt_loc.p, dinv, b, p0, pn, s, c1, c2, ghost, (int)(A.ncols / A.kron), \
                       A.dghost.p ? A.dghost.p : dinv)
    if (A.vt_rows == 16) { if (A.kron == 2) PCD_FIRST_TC(2, 16); else PCD_FIRST_TC(3, 16); }
    else if (A.kron == 2) PCD_FIRST_TC(2, 64); else PCD_FIRST_TC(3, 64);
#undef PCD_FIRST_TC
  } else if (A.rb2 && kron_ok(A, b, p0, pn, nullptr, true)) {
    const int nn = n / A.kron;
    LAUNCH_RBC(A, k_cheb_first_sc, grid_stream(nn, A.rb2), nn, A.rowptr2.p, A.col2.p,
               A.val2s.p, dinv, b, p0, pn, s, c1, c2, ghost, (int)(A.ncols / A.kron));
  } else {
    LAUNCH_RB(A, k_cheb_first_s, grid_stream(n, A.rb), n, A.rowptr.p, A.col.p,
              A.vals.p, dinv, b, p0, pn, s, c1, c2, ghost, (int)A.ncols);
  }
  };
  if (tile && overlap_ok(h, A)) {
    CHK(halo_send(h, A, b));               // (several ranks: the halo of b)
    h->ov_phase = 1; run();
    h->ov_phase = 0;
    CHK(halo_wait(h, A));
    h->ov_phase = 2; run();
    h->ov_phase = 0;
  } else {
    CHK(halo_exchange(h, A, b));
    run();
  }
  return 0;
}

// CG with the direction update fused into the SpMV: two launches per
// iteration (k_cg_spmv_s, k_cg_update); p ping-pongs between two buffers
int solve_cg_stream(Engine* h, const DCsr& A, Inner& s, const double* b,
                           double* x) {
  const int n = (int)A.nrows;
  const double* dinv = (s.pc == PCD_PC_JACOBI) ? A.dinv.p : nullptr;
  double *r = s.t0.p, *z = s.t1.p, *q = s.t3.p;
  double* P[2] = {s.t2.p, s.t4.p};
  double* R[2] = {s.parts.p, s.parts.p + kMaxParts};
  double* PQ = s.parts.p + 2 * kMaxParts;
  CgState* st = s.state.p;
  const int ge = grid1d(n, 2, kMaxParts);
  const int gs = grid_stream(n, A.rb, kMaxParts);
  hipLaunchKernelGGL(k_cg_init, dim3(ge), dim3(kBlock), 0, h->stream, n, dinv,
                     b, x, r, z, P[1], R[0], st);
  const int check = 32;
  for (int it = 0; it < s.max_it; ++it) {
    double* pn = P[it & 1];
    const double* po = P[(it + 1) & 1];
    LAUNCH_RB(A, k_cg_spmv_s, gs, n, A.rowptr.p, A.col.p, A.val.p, z, po, pn, q,
              R[it & 1], R[(it + 1) & 1], ge, s.rtol, it == 0 ? 1 : 0, PQ, st);
    hipLaunchKernelGGL(k_cg_update, dim3(ge), dim3(kBlock), 0, h->stream, n,
                       dinv, pn, q, x, r, z, R[it & 1], ge, PQ, gs,
                       R[(it + 1) & 1], it, st);
    if (s.rtol > 0.0 && (it % check) == check - 1 && it + 1 < s.max_it) {
      CHK(ensure_pinned(h, 8));
      int* flag = reinterpret_cast<int*>(h->pinned);
      HIPCHK(hipMemcpyAsync(flag, &st->done, sizeof(int), hipMemcpyDeviceToHost,
                            h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      if (*flag) break;
    }
  }
  HIPCHK(hipGetLastError());
  s.its_on_device = true; s.state_idx = 0;
  return 0;
}

int solve_cg(Engine* h, const DCsr& A, Inner& s, const double* b,
                    double* x) {
  // the fused two-launch form gathers z and p_old: single GPU only (the halo
  // would have to carry both); with several ranks the direction update is its
  // own launch, followed by the halo of p
  if (A.rb && !h->comm) return solve_cg_stream(h, A, s, b, x);
  const int n = (int)A.nrows;
  const double* dinv = (s.pc == PCD_PC_JACOBI) ? A.dinv.p : nullptr;
  double *r = s.t0.p, *z = s.t1.p, *p = s.t2.p, *q = s.t3.p;
  double* R[2] = {s.parts.p, s.parts.p + kMaxParts};
  double* PQ = s.parts.p + 2 * kMaxParts;
  double* slot = s.slots.p;                             // [R0, R1, PQ]
  CgState* st = s.state.p;
  const int ge = grid1d(n, 4, kMaxParts);               // element-wise grid
  const int gs = grid_rows(n, A.lpr, kMaxParts);        // SpMV + dot grid
  PartsRef rz[2], pq;
  hipLaunchKernelGGL(k_cg_init, dim3(ge), dim3(kBlock), 0, h->stream, n, dinv,
                     b, x, r, z, p, R[0], st);
  CHK(reduce_global(h, R[0], ge, slot + 0, &rz[0]));
  const int check = 16;
  for (int it = 0; it < s.max_it; ++it) {
    if (it > 0)
      hipLaunchKernelGGL(k_cg_pupdate, dim3(ge), dim3(kBlock), 0, h->stream, n,
                         z, p, rz[it & 1].p, rz[(it - 1) & 1].p, rz[it & 1].n,
                         s.rtol, st);
    CHK(halo_exchange(h, A, p));
    LAUNCH_LPR(A, k_cg_spmv_dot, gs, n, A.rowptr.p, A.col.p, A.val.p, p, q, PQ,
               st, A.ghost.p, (int)A.ncols);
    CHK(reduce_global(h, PQ, gs, slot + 2, &pq));
    hipLaunchKernelGGL(k_cg_update, dim3(ge), dim3(kBlock), 0, h->stream, n,
                       dinv, p, q, x, r, z, rz[it & 1].p, rz[it & 1].n, pq.p,
                       pq.n, R[(it + 1) & 1], it, st);
    CHK(reduce_global(h, R[(it + 1) & 1], ge, slot + ((it + 1) & 1),
                      &rz[(it + 1) & 1]));
    if (s.rtol > 0.0 && (it % check) == check - 1 && it + 1 < s.max_it) {
      CHK(ensure_pinned(h, 8));
      int* flag = reinterpret_cast<int*>(h->pinned);
      HIPCHK(hipMemcpyAsync(flag, &st->done, sizeof(int), hipMemcpyDeviceToHost,
                            h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      if (*flag) break;
    }
  }
  HIPCHK(hipGetLastError());
  s.its_on_device = true; s.state_idx = 0;
  return 0;
}

// [ext PETSc] KSPCG with -ksp_cg_single_reduction: per iteration one SpMV
// fused with both dot products, (several ranks: ONE all-reduce of two
// doubles,) one kernel with every vector update.
int solve_cg_sr(Engine* h, const DCsr& A, Inner& s, const double* b,
                       double* x) {
  const int n = (int)A.nrows;
  const double* dinv = (s.pc == PCD_PC_JACOBI) ? A.dinv.p : nullptr;
  double *r = s.t0.p, *z = s.t1.p, *p = s.t2.p, *sv = s.t3.p, *w = s.t4.p;
  double *PB = s.parts.p, *PD = s.parts.p + kMaxParts;
  double* slot = s.slots.p;                              // [beta, delta]
  CgState* st = s.state.p;
  const int ge = grid1d(n, 4, kMaxParts);
  const int gs = grid_rows(n, A.lpr, kMaxParts);
  hipLaunchKernelGGL(k_cgsr_init, dim3(ge), dim3(kBlock), 0, h->stream, n, dinv, b, x, r, z, st);
  const int check = 16;
  int it = 0;
  for (; it < s.max_it; ++it) {
    const CgState* sin = st + (it & 1);
    CgState* sout = st + ((it + 1) & 1);
    CHK(halo_exchange(h, A, z));
    LAUNCH_LPR(A, k_cgsr_spmv_dots, gs, n, A.rowptr.p, A.col.p, A.val.p, z, r, sv,
               PB, PD, sin, A.ghost.p, (int)A.ncols);
    const double *pb = PB, *pd = PD;
    int nb = gs, nd = gs;
    if (h->comm) {
      hipLaunchKernelGGL(k_sum_parts, dim3(2), dim3(kBlock), 0, h->stream, PB, gs, kMaxParts, slot);
      if (h->comm->allreduce(slot, 2, h->stream))
        return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
      pb = slot; pd = slot + 1; nb = nd = 1;
    }
    hipLaunchKernelGGL(k_cgsr_update, dim3(ge), dim3(kBlock), 0, h->stream, n, dinv, z, sv, p, w,
                       x, r, pb, nb, pd, nd, s.rtol, it, sin, sout);
    if (s.rtol > 0.0 && (it % check) == check - 1 && it + 1 < s.max_it) {
      CHK(ensure_pinned(h, 8));
      int* flag = reinterpret_cast<int*>(h->pinned);
      HIPCHK(hipMemcpyAsync(flag, &sout->done, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      if (*flag) { ++it; break; }
    }
  }
  HIPCHK(hipGetLastError());
  s.its_on_device = true;
  s.state_idx = it & 1;                  // the record the last update wrote
  return 0;
}

// [ext PETSc] KSPCHEBYSHEV recurrence coefficients are data independent, so
// the host computes them and every step is one fused launch.
int solve_cheb(Engine* h, const DCsr& A, Inner& s, const double* b,
                      double* x, double out_scale) {
  const int n = (int)A.nrows;
  // (PCD_PC_EXPLICIT: the step-by-step form of a stale factor chain)
  const double* dinv = (s.pc == PCD_PC_JACOBI || s.pc == PCD_PC_EXPLICIT) ? A.dinv.p : nullptr;
  const double scale = 2.0 / (s.emax + s.emin);
  const double alpha = 1.0 - scale * s.emin;
  const double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
  double c_km1 = 1.0, c_k = mu;
  // ring of three vectors arranged so that the last update lands in x
  double* ring[3];
  const int m = s.max_it;
  ring[m % 3] = x; ring[(m + 1) % 3] = s.t0.p; ring[(m + 2) % 3] = s.t1.p;
  if (A.cp.ready && A.cp.m == m && s.pc == PCD_PC_JACOBI && dinv && !h->comm) {
    ChebPatchCoef cf;
    cf.scale = scale; cf.m = m;
    for (int it = 0; it < m; ++it) {
      const double c_kp1 = 2.0 * mu * c_k - c_km1;
      const double omega = omegaprod * c_k / c_kp1;
      const double f = (it == m - 1) ? out_scale : 1.0;
      cf.c0[it] = it == 0 ? 0.0 : f * (1.0 - omega);
      cf.c1[it] = f * omega; cf.c2[it] = f * omega * scale;
      c_km1 = c_k; c_k = c_kp1;
    }
    if (A.cp.wmax <= 8)
      hipLaunchKernelGGL(k_cheb_patch<8>, dim3(A.cp.nblocks), dim3(kPatchThreads), 0, h->stream, A.cp.desc.p,
                         A.cp.cnt.p, A.cp.node.p, A.cp.col.p, A.cp.val.p, dinv, b, x, cf);
    else
      hipLaunchKernelGGL(k_cheb_patch<16>, dim3(A.cp.nblocks), dim3(kPatchThreads), 0, h->stream, A.cp.desc.p,
                         A.cp.cnt.p, A.cp.node.p, A.cp.col.p, A.cp.val.p, dinv, b, x, cf);
    HIPCHK(hipGetLastError());
    s.last_its = m; s.its_on_device = false;
    return 0;
  }
  const int g1 = grid1d(n, 1);
  const bool fuse = m >= 1 && can_fuse_first(h, A, dinv);
  if (!fuse)
    hipLaunchKernelGGL(k_scale_dinv, dim3(g1), dim3(kBlock), 0, h->stream, n,
                       dinv, b, m == 0 ? scale * out_scale : scale, ring[0]);
  for (int it = 0; it < m; ++it) {
    const double c_kp1 = 2.0 * mu * c_k - c_km1;
    const double omega = omegaprod * c_k / c_kp1;
    const double f = (it == m - 1) ? out_scale : 1.0;   // sign folded in
    double* pk = ring[it % 3];
    double* pn = ring[(it + 1) % 3];
    if (it == 0 && fuse) {
      // p_{-1} = 0: p1 = omega p0 + omega scale D^-1 (b - A p0)
      CHK(launch_cheb_first(h, A, dinv, b, m >= 2 ? pk : nullptr, pn, scale,
                            f * omega, f * omega * scale));
    } else {
      // p_{-1} = 0 at the first step: coefficient forced to zero, never read
      double* pm = (it == 0) ? pk : ring[(it + 2) % 3];
      const double c0 = (it == 0) ? 0.0 : 1.0 - omega;
      CHK(launch_cheb_step(h, A, dinv, b, pm, pk, pn, f * c0, f * omega,
                           f * omega * scale));
    }
    c_km1 = c_k; c_k = c_kp1;
  }
  HIPCHK(hipGetLastError());
  s.last_its = m; s.its_on_device = false;
  return 0;
}

int solve_rich(Engine* h, const DCsr& A, Inner& s, const double* b,
                      double* x) {
  const int n = (int)A.nrows;
  const double* dinv = (s.pc == PCD_PC_JACOBI) ? A.dinv.p : nullptr;
  const int m = std::max(s.max_it, 1);
  // iteration 0 with zero guess is x = B b; then m-1 fused sweeps ping-pong
  double* bufs[2];
  bufs[(m - 1) % 2] = x; bufs[m % 2] = s.t0.p;
  hipLaunchKernelGGL(k_scale_dinv, dim3(grid1d(n, 1)), dim3(kBlock), 0,
                     h->stream, n, dinv, b, 1.0, bufs[0]);
  for (int it = 1; it < m; ++it) {
    double* pk = bufs[(it - 1) % 2];
    double* pn = bufs[it % 2];
    CHK(launch_cheb_step(h, A, dinv, b, pk, pk, pn, 0.0, 1.0, 1.0));
  }
  HIPCHK(hipGetLastError());
  s.last_its = m; s.its_on_device = false;
  return 0;
}

// ---- [ext PETSc] PCMG: multiplicative V-cycle on the device -----------------
// Chebyshev-Jacobi smoothing; every step is one fused k_cheb_step launch.
// Iterates rotate through bufs[0..2]; with a nonzero guess the guess sits in
// bufs[0].  *result points at the buffer holding the smoothed vector.
int mg_smooth(Engine* h, const DCsr& A, double emin, double emax, int nu,
                     const double* b, double* bufs[3], bool zero_guess,
                     double** result) {
  const int n = (int)A.nrows;
  if (nu == 0) {
    if (zero_guess) HIPCHK(hipMemsetAsync(bufs[0], 0, n * sizeof(double), h->stream));
    *result = bufs[0];
    return 0;
  }
  const double* dinv = A.dinv.p;
  const double scale = 2.0 / (emax + emin);
  const double alpha = 1.0 - scale * emin;
  const double mu = 1.0 / alpha, omegaprod = 2.0 / alpha;
  double c_km1 = 1.0, c_k = mu;
  int cur;                       // index in bufs of the newest iterate
  bool have_pm;                  // p_{k-1} is a real vector (not zero)
  if (zero_guess && nu >= 2 && can_fuse_first(h, A, dinv)) {
    // Jacobi start + first step in one launch; p0 kept only if a later step
    // needs it as p_{k-1}
    const double c_kp1 = 2.0 * mu * c_k - c_km1;
    const double omega = omegaprod * c_k / c_kp1;
    CHK(launch_cheb_first(h, A, dinv, b, nu >= 3 ? bufs[0] : nullptr, bufs[1],
                          scale, omega, omega * scale));
    c_km1 = c_k; c_k = c_kp1;
    cur = 1; have_pm = true;
    --nu;                                  // one step already done
  } else if (zero_guess) {
    hipLaunchKernelGGL(k_scale_dinv, dim3(grid1d(n, 1)), dim3(kBlock), 0,
                       h->stream, n, dinv, b, scale, bufs[0]);
    cur = 0; have_pm = false;
  } else {
    CHK(launch_cheb_step(h, A, dinv, b, bufs[0], bufs[0], bufs[1], 0.0, 1.0, scale));
    cur = 1; have_pm = true;
  }
  for (int it = 0; it < nu - 1; ++it) {
    const double c_kp1 = 2.0 * mu * c_k - c_km1;
    const double omega = omegaprod * c_k / c_kp1;
    double* pk = bufs[cur % 3];
    double* pn = bufs[(cur + 1) % 3];
    double* pm = have_pm ? bufs[(cur + 2) % 3] : pk;
    CHK(launch_cheb_step(h, A, dinv, b, pm, pk, pn, have_pm ? 1.0 - omega : 0.0,
                         omega, omega * scale));
    c_km1 = c_k; c_k = c_kp1;
    ++cur; have_pm = true;
  }
  HIPCHK(hipGetLastError());
  *result = bufs[cur % 3];
  return 0;
}

// x_l = V-cycle(b) on level l; *out points at the level buffer with the result
int mg_vcycle(Engine* h, const DCsr& Afine, Inner& s, int l,
                     const double* b, double** out, double* target) {
  MgLevel& L = s.mg[l];
  if (l == 0) {
    double* dst = target ? target : L.x.p;
    CHK(spmv(h, L.A, b, dst));                // explicit coarse inverse
    *out = dst;
    return 0;
  }
  const DCsr& A = (l == (int)s.mg.size() - 1) ? Afine : L.A;
  MgLevel& C = s.mg[l - 1];
  if (L.fused) {
    // pre-composed level: x1 = smooth(b); r_c = Wd b; e_c = cycle(r_c);
    // x = Wu [T | b]
    const int64_t n = A.nrows, nc = L.P.ncols;
    double* T = L.T.p;
    {
      // ring arranged so that the smoothed vector lands in T[0, n)
      const int last = (s.nu_pre - 1) % 3;
      double* ring[3];
      ring[last] = T; ring[(last + 1) % 3] = L.t0.p; ring[(last + 2) % 3] = L.t1.p;
      double* px = nullptr;
      CHK(mg_smooth(h, A, L.emin, L.emax, s.nu_pre, b, ring, true, &px));
      if (px != T)
        HIPCHK(hipMemcpyAsync(T, px, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    CHK(spmv(h, L.Wd, b, T + n));
    double* pe = nullptr;
    CHK(mg_vcycle(h, Afine, s, l - 1, T + n, &pe, T + n + nc));
    if (pe != T + n + nc)
      HIPCHK(hipMemcpyAsync(T + n + nc, pe, nc * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    double* dst = target ? target : L.x.p;
    CHK(spmv(h, L.Wu, T, dst, 0, nullptr, b, n + 2 * nc));
    *out = dst;
    return 0;
  }
  double* bufs[3] = {L.x.p, L.t0.p, L.t1.p};
  double* px = nullptr;
  CHK(mg_smooth(h, A, L.emin, L.emax, s.nu_pre, b, bufs, true, &px));
  const double* r = b;
  if (s.nu_pre > 0) {
    CHK(spmv(h, A, px, L.r.p, 2, b));        // r = b - A x
    r = L.r.p;
  }
  CHK(spmv(h, L.R, r, C.b.p));                // restrict
  if (L.transition && h->comm->allreduce(C.b.p, (size_t)L.n_coarse, h->stream))
    return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  double* pe = nullptr;
  CHK(mg_vcycle(h, Afine, s, l - 1, C.b.p, &pe));
  double* post[3];
  if (target) {
    // arrange the ring so that the last smoothing step writes the caller's
    // vector: no copy at the end of the cycle
    const int last = s.nu_post % 3;
    double* spare[2] = {L.t0.p, L.t1.p};
    if (px == spare[0] || px == spare[1]) spare[px == spare[0] ? 0 : 1] = L.x.p;
    post[last] = target;
    post[(last + 1) % 3] = spare[0];
    post[(last + 2) % 3] = spare[1];
  } else {
    int j = 0;
    post[0] = px;
    for (double* q : bufs) if (q != px) post[++j] = q;
  }
  CHK(spmv(h, L.P, pe, post[0], 1, px));      // post[0] = x + P e
  CHK(mg_smooth(h, A, L.emin, L.emax, s.nu_post, b, post, false, out));
  return 0;
}

// KSPPREONLY (one cycle) / KSPRICHARDSON (max_it cycles) around the V-cycle
int solve_mg(Engine* h, const DCsr& A, Inner& s, const double* b,
                    double* x) {
  const int n = (int)A.nrows;
  const int L = (int)s.mg.size();
  if (L < 1) return fail(PCD_ERR_STATE, "pc mg: no hierarchy (pcd_mg_begin / pcd_mg_set_level)");
  for (int l = 0; l < L; ++l) {
    const MgLevel& M = s.mg[l];
    if ((l < L - 1 && !M.A.set) || (l > 0 && !M.P.set))
      return fail(PCD_ERR_STATE, "pc mg: level %d incomplete", l);
  }
  if (s.mg[L - 1].P.set && s.mg[L - 1].P.nrows != n)
    return fail(PCD_ERR_ARG, "pc mg: finest prolongation has %lld rows, operator %d",
                (long long)s.mg[L - 1].P.nrows, n);
  const int its = (s.ksp == PCD_KSP_PREONLY) ? 1 : std::max(s.max_it, 1);
  const int g = grid1d(n, 1);
  for (int it = 0; it < its; ++it) {
    const double* r = b;
    if (it > 0) {
      CHK(spmv(h, A, x, s.t0.p, 2, b));       // r = b - A x
      r = s.t0.p;
    }
    double* z = nullptr;
    if (L == 1) { CHK(spmv(h, s.mg[0].A, r, it == 0 ? x : s.mg[0].x.p)); z = it == 0 ? x : s.mg[0].x.p; }
    else CHK(mg_vcycle(h, A, s, L - 1, r, &z, it == 0 ? x : nullptr));
    if (it == 0) {
      if (z != x) hipLaunchKernelGGL(k_copy, dim3(g), dim3(kBlock), 0, h->stream, n, z, x);
    } else hipLaunchKernelGGL(k_axpby, dim3(g), dim3(kBlock), 0, h->stream, n, 1.0, z, 1.0, x);
  }
  HIPCHK(hipGetLastError());
  s.last_its = its; s.its_on_device = false;
  return 0;
}

// KSP.solve(b, x): b and x must not alias.  `out_scale` asks for x scaled by
// a constant; *scaled tells whether the solver folded it in (for free) or the
// caller still has to apply it.
int inner_solve(Engine* h, int slot, const double* b, double* x,
                       double out_scale, bool* scaled) {
  const DCsr& A = h->mat[kSlotMat[slot]];
  Inner& s = h->inner[slot];
  if (!A.set) return fail(PCD_ERR_STATE, "inner_solve: operator of slot %d not set", slot);
  if (b == x) return fail(PCD_ERR_ARG, "inner_solve: b and x alias");
  if (scaled) *scaled = false;
  if (s.pc != PCD_PC_MG && s.ksp == PCD_KSP_CHEBYSHEV) {
    if (scaled) *scaled = true;
    return solve_cheb(h, A, s, b, x, scaled ? out_scale : 1.0);
  }
  if (s.pc == PCD_PC_MG) {
    if (s.ksp != PCD_KSP_PREONLY && s.ksp != PCD_KSP_RICHARDSON)
      return fail(PCD_ERR_ARG, "pc mg is supported under preonly / richardson only");
    return solve_mg(h, A, s, b, x);
  }
  if (s.pc == PCD_PC_EXPLICIT) {
    // x = W_{m-1} ... W_0 b: one sparse product per factor; a sign asked for
    // by the caller rides on the last one
    const int m = (int)s.chain.size();
    if (m < 1) return fail(PCD_ERR_STATE, "pc explicit: no factors (pcd_set_inner_factor)");
    if (s.chain_stale) {
      // the operator changed after the factors were composed: run the
      // recurrence they stand for - max_it Chebyshev-Jacobi steps with the
      // bounds kept in emin / emax - until new factors arrive
      if (!(s.emax > s.emin && s.emin > 0.0))
        return fail(PCD_ERR_STATE, "pc explicit: the operator of slot %d was updated after its factors "
                                   "were composed and no Chebyshev bounds were given to fall back on; "
                                   "hand over new factors (pcd_set_inner_factor)", slot);
      if (scaled) *scaled = true;
      return solve_cheb(h, A, s, b, x, scaled ? out_scale : 1.0);
    }
    for (const DCsr& F : s.chain)
      if (!F.set || F.nrows != A.nrows || F.ncols != A.ncols)
        return fail(PCD_ERR_STATE, "pc explicit: factors incomplete or of the wrong size");
    const bool neg = scaled && out_scale == -1.0;
    if (scaled) *scaled = neg || out_scale == 1.0;
    const double* in = b;
    for (int j = 0; j < m; ++j) {
      double* dst = (j == m - 1) ? x : ((j & 1) ? s.t1.p : s.t0.p);
      CHK(spmv(h, s.chain[j], in, dst, (j == m - 1 && neg) ? 3 : 0));
      in = dst;
    }
    s.last_its = s.max_it; s.its_on_device = false;
    return 0;
  }
  switch (s.ksp) {
    case PCD_KSP_PREONLY: {
      const double* dinv = (s.pc == PCD_PC_JACOBI) ? A.dinv.p : nullptr;
      hipLaunchKernelGGL(k_scale_dinv, dim3(grid1d(A.nrows, 1)), dim3(kBlock), 0,
                         h->stream, (int)A.nrows, dinv, b, 1.0, x);
      HIPCHK(hipGetLastError());
      s.last_its = 1; s.its_on_device = false;
      return 0;
    }
    case PCD_KSP_RICHARDSON: return solve_rich(h, A, s, b, x);
    case PCD_KSP_CHEBYSHEV: return solve_cheb(h, A, s, b, x);
    case PCD_KSP_CG: return solve_cg(h, A, s, b, x);
    case PCD_KSP_CG_SR: return solve_cg_sr(h, A, s, b, x);
  }
  return fail(PCD_ERR_ARG, "inner_solve: unknown ksp type %d", s.ksp);
}

// ---------------------------------------------------------- apply bodies
int apply_bc_dev(Engine* h, double* x) {
  if (h->n_bc == 0) return 0;
  hipLaunchKernelGGL(k_bc_set, dim3(grid1d(h->n_bc, 1, 1 << 30)), dim3(kBlock), 0,
                     h->stream, (int)h->n_bc, h->bc_idx.p, h->bc_val.p, x);
  HIPCHK(hipGetLastError());
  return 0;
}

// The four PCPYTHON apply bodies on device pointers (x, y distinct, n_p long)
int pcd_apply_dev(Engine* h, const double* x, double* y) {
  const int n = (int)h->np_loc;
  const int g = grid1d(n, 1);
  const bool reaction = h->variant == PCDR_BRM1 || h->variant == PCDR_BRM2;
  if (h->variant == PCD_BRM1 || h->variant == PCDR_BRM1) {
    double* z = h->w[0].p;
    // z = x; bcs_applier(z): copy and VecSetValues(INSERT) in one launch
    hipLaunchKernelGGL(k_copy_bc, dim3(g), dim3(kBlock), 0, h->stream, n, x,
                       h->bc_slot.p, h->bc_val.p, z);
    CHK(inner_solve(h, PCD_KSP_AP, z, y));                      // y = Ap^-1 z
    CHK(spmv(h, h->mat[PCD_MAT_KP], y, z, 1, x));               // z = Kp y + x
    if (reaction) {
      CHK(inner_solve(h, PCD_KSP_MP, z, y));                    // y = Mp^-1 z
      CHK(inner_solve(h, PCD_KSP_RP, x, z));                    // z = Rp^-1 x
      hipLaunchKernelGGL(k_axpby, dim3(g), dim3(kBlock), 0, h->stream, n, -1.0,
                         z, -1.0, y);                           // y = -(y + z)
    } else {
      // y = -(Mp^-1 z): the sign rides on the last Chebyshev step when it can
      bool scaled = false;
      CHK(inner_solve(h, PCD_KSP_MP, z, y, -1.0, &scaled));
      if (!scaled)
        hipLaunchKernelGGL(k_axpby, dim3(g), dim3(kBlock), 0, h->stream, n, -1.0,
                           y, 0.0, y);                          // y = -y
    }
  } else {
    double *z0 = h->w[0].p, *z1 = h->w[1].p;
    CHK(inner_solve(h, PCD_KSP_MP, x, y));                      // y = Mp^-1 x
    CHK(spmv(h, h->mat[PCD_MAT_KP], y, z1));                    // z1 = Kp y
    CHK(apply_bc_dev(h, z1));                                   // bcs_applier(z1)
    CHK(inner_solve(h, PCD_KSP_AP, z1, z0));                    // z0 = Ap^-1 z1
    if (reaction) {
      hipLaunchKernelGGL(k_axpby, dim3(g), dim3(kBlock), 0, h->stream, n, 1.0, z0,
                         1.0, y);                               // y += z0
      CHK(inner_solve(h, PCD_KSP_RP, x, z0));                   // z0 = Rp^-1 x
    }
    hipLaunchKernelGGL(k_axpby, dim3(g), dim3(kBlock), 0, h->stream, n, -1.0, z0,
                       -1.0, y);                                // y = -(y + z0)
  }
  HIPCHK(hipGetLastError());
  ++h->num_pcd;
  return 0;
}

// [ext PETSc] PCApply_FieldSplit_Schur (UPPER) on split-ordered vectors
int fs_apply_eager(Engine* h, const double* x, double* y) {
  const int64_t nu = h->nu_loc;
  const double *xu = x, *xp = x + nu;
  double *yu = y, *yp = y + nu, *t = h->wu.p;
  CHK(pcd_apply_dev(h, xp, yp));                                // y_p = S^-1 x_p
  CHK(spmv(h, h->mat[PCD_MAT_A01], yp, t, 2, xu));              // t = x_u - A01 y_p
  CHK(inner_solve(h, PCD_KSP_A00, t, yu));                      // y_u = A00^-1 t
  return 0;
}

// A PCApply whose inner solvers all run a fixed number of steps contains no
// host decision: ~100 short launches.  It is captured once into a hipGraph on
// the fixed staging vectors (xs -> ys) and replayed (hard part 3:
// this path is launch-bound at the 2D sizes).
bool graph_capturable(const Engine* h) {
  const bool reaction = h->variant == PCDR_BRM1 || h->variant == PCDR_BRM2;
  for (int slot : {PCD_KSP_AP, PCD_KSP_MP, PCD_KSP_RP, PCD_KSP_A00}) {
    if (slot == PCD_KSP_RP && !reaction) continue;
    const Inner& s = h->inner[slot];
    if (s.pc != PCD_PC_MG && (s.ksp == PCD_KSP_CG || s.ksp == PCD_KSP_CG_SR) && s.rtol > 0.0) return false;
  }
  return true;
}

int fs_apply_split(Engine* h, const double* x, double* y) {
  ++h->num_fs;
  if (!h->graph_on || !graph_capturable(h)) return fs_apply_eager(h, x, y);
  if (h->comm) {
    if (!h->comm->peer()) return fs_apply_eager(h, x, y);
    PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
    if (h->gcheck_gen != h->gen) {
      pb->boot_calls = 0; h->boot_exchanges = 0;
      const int rc = fs_apply_eager(h, x, y);
      h->gcheck_gen = h->gen;
      h->g_ok = rc == 0 && pb->boot_calls == 0 && h->boot_exchanges == 0;
      return rc;
    }
    if (!h->g_ok) return fs_apply_eager(h, x, y);
  }
  const int n = (int)(h->nu_loc + h->np_loc);
  if (!h->gexec || h->ggen != h->gen) {
    if (h->gexec) { (void)hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
    if (!h->cap_stream)
      HIPCHK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamSynchronize(h->stream));
    hipStream_t saved = h->stream;
    h->stream = h->cap_stream;
    hipError_t e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
      h->stream = saved;
      return fail(PCD_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    }
    const int rc = fs_apply_eager(h, h->xs.p, h->ys.p);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(h->cap_stream, &g);
    h->stream = saved;
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(PCD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&h->gexec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { h->gexec = nullptr; return fail(PCD_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    h->ggen = h->gen;
  }
  const int g1 = grid1d(n, 2);
  if (x != h->xs.p)
    hipLaunchKernelGGL(k_copy, dim3(g1), dim3(kBlock), 0, h->stream, n, x, h->xs.p);
  HIPCHK(hipGraphLaunch(h->gexec, h->stream));
  if (y != h->ys.p)
    hipLaunchKernelGGL(k_copy, dim3(g1), dim3(kBlock), 0, h->stream, n, h->ys.p, y);
  HIPCHK(hipGetLastError());
  return 0;
}


// A/B switches of the XCD-aware row-block mapping: __constant__ data of THIS
// translation unit, where every kernel that reads them is launched
int apply_configure_constants() {
  { const char* e = getenv("PCD_XCD_REMAP_NT");            // A/B: mapping of the non-temporal kernels
    if (e) { const int v = atoi(e); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(pcd::g_xcd_remap_nt), &v, sizeof(int))); } }
  { const char* e = getenv("PCD_XCD_REMAP_NT3_ROWS");      // node rows from which the 3-component kernels map too
    if (e) { const int v = atoi(e); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(pcd::g_xcd_remap_nt3_rows), &v, sizeof(int))); } }
  { const char* e = getenv("PCD_XCD_REMAP_MAX_ROWS");      // A/B: threshold of the XCD-aware mapping
    if (e) { const int v = atoi(e); HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(pcd::g_xcd_remap_max_rows), &v, sizeof(int))); } }
  { const char* e = getenv("PCD_NO_XCD_REMAP");
    if (e && e[0] == '1') {
      const int none = 0;
      HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(pcd::g_xcd_remap_max_rows), &none, sizeof(int)));
    } }
  return 0;
}

