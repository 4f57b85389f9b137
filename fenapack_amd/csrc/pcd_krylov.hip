// pcd_krylov.hip - outer GMRES on the device
// (one of the engine's translation units; shared declarations: pcd_internal.hpp)
#include "pcd_internal.hpp"

// sqrt(v.v) over all ranks; synchronises
int dev_norm(Engine* h, int64_t n, const double* v, double* out) {
  const int G = grid1d(n, 4, 512);
  hipLaunchKernelGGL(k_mdot, dim3(G, 1), dim3(kBlock), 0, h->stream, n, v, (int64_t)0, 1, v, h->gparts.p, G);
  hipLaunchKernelGGL(k_mdot_reduce, dim3(1), dim3(kBlock), 0, h->stream, h->gparts.p, G, h->gh.p);
  if (h->comm && h->comm->allreduce(h->gh.p, 1, h->stream))
    return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  HIPCHK(hipMemcpyAsync(h->pinned, h->gh.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *out = std::sqrt(h->pinned[0]);
  return 0;
}

// w = A z on split-ordered vectors.  One GPU with P = A: block-wise through
// A00 (F x I fast path), A01, A10 (and A11 if it is not zero) - about half
// the bytes of the monolithic CSR; otherwise the monolithic operator.
int apply_system(Engine* h, const double* z, double* w) {
  static const bool mono = [] { const char* e = getenv("PCD_SYSTEM_MONOLITHIC"); return e && e[0] == '1'; }();
  if (mono || !h->p_is_a || !h->a10.set || !h->mat[PCD_MAT_A00].set || !h->mat[PCD_MAT_A01].set)
    return spmv(h, h->mat[PCD_MAT_A], z, w);
  const int64_t nu = h->nu_loc;
  const DCsr &A00 = h->mat[PCD_MAT_A00], &A01 = h->mat[PCD_MAT_A01];
  const bool with11 = !h->a11_zero;
  if (h->comm) {
    // one grouped exchange carries the ghosts of every block
    if (with11) CHK(halo_exchange_group(h, {{&A00, z}, {&A01, z + nu}, {&h->a10, z}, {&h->a11, z + nu}}));
    else CHK(halo_exchange_group(h, {{&A00, z}, {&A01, z + nu}, {&h->a10, z}}));
  }
  CHK(spmv(h, A00, z, w, 0, nullptr, nullptr, 0, true));
  CHK(spmv(h, A01, z + nu, w, 1, w, nullptr, 0, true));
  CHK(spmv(h, h->a10, z, w + nu, 0, nullptr, nullptr, 0, true));
  if (with11) CHK(spmv(h, h->a11, z + nu, w + nu, 1, w + nu, nullptr, 0, true));
  return 0;
}

int pcd_gmres_solve(pcd_handle h, const double* b, double* x, int mem,
                    double rtol, double atol, int m, int max_it, int* its,
                    double* rnorm) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!h->ready || !h->mat[PCD_MAT_A].set)
    return fail(PCD_ERR_STATE, "gmres_solve: pcd_set_system + pcd_setup first");
  if (!b || !x || m < 1 || max_it < 0) return fail(PCD_ERR_ARG, "gmres_solve: bad arguments");
  if (m > 255) return fail(PCD_ERR_ARG, "gmres_solve: restart %d exceeds 255", m);
  const int64_t nglob = h->n_u + h->n_p;
  const int64_t n = h->nu_loc + h->np_loc;               // rows of this rank
  const bool local_io = h->comm && mem == PCD_MEM_DEVICE;
  const int64_t ld = (n + 15) / 16 * 16;
  if (h->V_m < m || h->V_ld != ld) {
    CHK(h->V.ensure((size_t)ld * (m + 1)));
    h->V_m = m; h->V_ld = ld;
  }
  const int G = grid1d(n, 4, 512);
  CHK(h->gz.ensure(n)); CHK(h->gw.ensure(n)); CHK(h->gxs.ensure(n)); CHK(h->gbs.ensure(n));
  CHK(h->gparts.ensure((size_t)(m + 2) * 512)); CHK(h->gh.ensure(m + 2)); CHK(h->gy.ensure(m + 2));
  CHK(ensure_pinned(h, (size_t)m + 24));
  IoMap io;
  if (local_io) { io.h = h; io.mem = mem; io.dx = b; io.dy = x; }
  else CHK(io_begin(h, io, b, nglob, x, nglob, mem));
  double *V = h->V.p, *z = h->gz.p, *xs = h->gxs.p, *bs = h->gbs.p;
  const int g1 = grid1d(n, 1);
  if (local_io) HIPCHK(hipMemcpyAsync(bs, io.dx, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  else hipLaunchKernelGGL(k_gather, dim3(g1), dim3(kBlock), 0, h->stream, (int)n, h->perm.p, io.dx, bs);
  HIPCHK(hipMemsetAsync(xs, 0, n * sizeof(double), h->stream));
  double bnorm = 0.0;
  CHK(dev_norm(h, n, bs, &bnorm));
  const double tol = std::max(rtol * bnorm, atol);
  // Hessenberg matrix, Givens rotations and the stopping test live on the
  // device (k_gmres_column); the host reads a 32-byte status ONE ITERATION
  // LATE: iteration k is queued before the status of iteration k-1 is waited
  // for, so the device never idles on the host and every rank of a
  // partitioned run takes the same decision at the same point (the status is
  // computed from all-reduced numbers).  Price: one over-run iteration per
  // solve, whose results are ignored (the state is frozen once `done`).
  CHK(h->gH.ensure((size_t)(m + 1) * m)); CHK(h->gcs.ensure(m)); CHK(h->gsn.ensure(m));
  CHK(h->gg.ensure(m + 1)); CHK(h->gstat.ensure(1));
  for (auto& e : h->gev) if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // (pinned[0..7] carry the inner solvers' convergence flag and dev_norm's
  // result: the two status slots live behind them)
  GmresStatus* pst = reinterpret_cast<GmresStatus*>(h->pinned + 8);   // 2 slots
  static_assert(sizeof(GmresStatus) == 24 || sizeof(GmresStatus) == 32, "status layout");
  int it = 0;
  double res = bnorm;
  // r0 = b (zero initial guess), kept in V_0 storage
  HIPCHK(hipMemcpyAsync(V, bs, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  double beta = bnorm;
  const DCsr& A = h->mat[PCD_MAT_A];
  while (it < max_it && res > tol) {
    res = beta;
    if (beta <= tol) break;
    hipLaunchKernelGGL(k_axpby, dim3(g1), dim3(kBlock), 0, h->stream, (int)n, 1.0 / beta, V, 0.0, V);
    hipLaunchKernelGGL(k_gmres_reset, dim3(1), dim3(64), 0, h->stream, beta, m, h->gg.p, h->gstat.p);
    int k = 0;                 // iterations of this cycle queued so far
    int kfin = -1;             // columns that count, once known
    GmresStatus fin = {};
    auto enqueue = [&](int kk) -> int {
      double* vk = V + (size_t)kk * ld;
      double* vn = V + (size_t)(kk + 1) * ld;
      CHK(fs_apply_split(h, vk, z));                           // z = M^-1 v_k
      CHK(apply_system(h, z, vn));                             // w = A z
      const int nvec = kk + 1;
      const int tiles = (nvec + kDotTile - 1) / kDotTile;
      // classical Gram-Schmidt: all k+1 dots in one batch, ONE all-reduce
      hipLaunchKernelGGL(k_mdot, dim3(G, tiles), dim3(kBlock), 0, h->stream, n, V, ld, nvec, vn, h->gparts.p, G);
      hipLaunchKernelGGL(k_mdot_reduce, dim3(nvec), dim3(kBlock), 0, h->stream, h->gparts.p, G, h->gh.p);
      if (h->comm && h->comm->allreduce(h->gh.p, nvec, h->stream))
        return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
      hipLaunchKernelGGL(k_maxpy_norm, dim3(G), dim3(kBlock), 0, h->stream, n, V, ld, nvec, h->gh.p, vn, -1.0, h->gparts.p);
      PartsRef nr;
      CHK(reduce_global(h, h->gparts.p, G, h->gy.p + m + 1, &nr));
      hipLaunchKernelGGL(k_normalize, dim3(G), dim3(kBlock), 0, h->stream, n, vn, nr.p, nr.n, h->gh.p + nvec);
      hipLaunchKernelGGL(k_gmres_column, dim3(1), dim3(64), 0, h->stream, kk, m, h->gh.p, h->gH.p,
                         h->gcs.p, h->gsn.p, h->gg.p, tol, h->gstat.p);
      HIPCHK(hipMemcpyAsync(&pst[kk & 1], h->gstat.p, sizeof(GmresStatus), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipEventRecord(h->gev[kk & 1], h->stream));
      return 0;
    };
    while (k < m && it + k < max_it) {
      CHK(enqueue(k));
      ++k;
      if (k >= 2) {                       // status of iteration k-2, one late
        HIPCHK(hipEventSynchronize(h->gev[(k - 2) & 1]));
        const GmresStatus st = pst[(k - 2) & 1];
        if (st.done) { fin = st; kfin = st.kconv; break; }
      }
    }
    if (kfin < 0) {                       // natural end of the cycle
      HIPCHK(hipEventSynchronize(h->gev[(k - 1) & 1]));
      fin = pst[(k - 1) & 1];
      kfin = fin.done ? fin.kconv : k;
    }
    if (fin.code == 1) return fail(PCD_ERR_BREAKDOWN, "gmres: non-finite Hessenberg entry at iteration %d", it + kfin);
    if (fin.code == 2) return fail(PCD_ERR_BREAKDOWN, "gmres: singular Hessenberg column at iteration %d", it + kfin);
    const bool breakdown = fin.code == 3;
    it += kfin;
    res = fin.res;
    k = kfin;
    // y = H^-1 g on the device; x += M^-1 (V y)
    if (k > 0) {
      hipLaunchKernelGGL(k_gmres_ysolve, dim3(1), dim3(64), 0, h->stream, k, m, h->gH.p, h->gg.p, h->gy.p);
      hipLaunchKernelGGL(k_combine, dim3(G), dim3(kBlock), 0, h->stream, n, V, ld, k, h->gy.p, h->gw.p);
      CHK(fs_apply_split(h, h->gw.p, z));
      hipLaunchKernelGGL(k_axpby, dim3(g1), dim3(kBlock), 0, h->stream, (int)n, 1.0, z, 1.0, xs);
    }
    if (res <= tol || it >= max_it || breakdown) break;
    CHK(spmv(h, A, xs, V, 2, bs));                              // r = b - A x
    CHK(dev_norm(h, n, V, &beta));
  }
  if (local_io) {
    HIPCHK(hipMemcpyAsync(io.dy, xs, n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else {
    if (h->comm) HIPCHK(hipMemsetAsync(io.dy, 0, nglob * sizeof(double), h->stream));
    hipLaunchKernelGGL(k_scatter, dim3(g1), dim3(kBlock), 0, h->stream, (int)n, h->perm.p, xs, io.dy);
    HIPCHK(hipGetLastError());
    if (h->comm && h->comm->allreduce(io.dy, nglob, h->stream))
      return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
    CHK(io_end(io));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  h->gmres_its = it; h->gmres_rnorm = res;
  if (its) *its = it;
  if (rnorm) *rnorm = res;
  return peer_check(h);
} PCD_ABI_CATCH(pcd_gmres_solve)

