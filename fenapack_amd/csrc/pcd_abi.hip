// pcd_abi.hip - lifetime, vectors across the ABI, info, communicators
// (one of the engine's translation units; shared declarations: pcd_internal.hpp)
#include "pcd_internal.hpp"


const char* pcd_last_error(void) { return g_err; }

int pcd_create(pcd_handle* out, int variant, int device) try {
  if (!out) return fail(PCD_ERR_ARG, "create: null out");
  if (variant < PCD_BRM1 || variant > PCDR_BRM2)
    return fail(PCD_ERR_ARG, "create: bad variant %d", variant);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev)
    return fail(PCD_ERR_ARG, "create: device %d not in [0,%d)", device, ndev);
  HIPCHK(hipSetDevice(device));
  { const char* e = getenv("PCD_FORCE_CSR_VECTOR"); g_force_vector = e && e[0] == '1'; }
  { const char* e = getenv("PCD_NO_KRON2"); g_no_kron = e && e[0] == '1'; }
  { const char* e = getenv("PCD_MAX_RB"); if (e && atoi(e) >= 32) g_max_rb = atoi(e); }
  { const char* e = getenv("PCD_MIN_WGS"); if (e) g_min_wgs = atoi(e); }
  { const char* e = getenv("PCD_MAX_CHUNKS"); if (e && atoi(e) >= 1) g_max_chunks = atoi(e); }
  { const char* e = getenv("PCD_NT_BYTES"); g_nt_bytes = e ? atoll(e) : (256ll << 20); }
  { const char* e = getenv("PCD_NO_SMALL_TILE"); g_no_small_tile = e && e[0] == '1'; }
  { hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      g_num_cus = prop.multiProcessorCount; }
  CHK(apply_configure_constants());
  Engine* h = new (std::nothrow) Engine();
  if (!h) return fail(PCD_ERR_NOMEM, "create: out of host memory");
  h->variant = variant; h->device = device;
  *out = h;
  return 0;
} PCD_ABI_CATCH(pcd_create)

int pcd_destroy(pcd_handle h) try {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  fe_release(h);
  h->a10.release(); h->a11.release();
  for (auto& m : h->mat) m.release();
  for (auto& s : h->inner) s.release();
  h->bc_idx.release(); h->bc_val.release(); h->bc_slot.release(); h->perm.release();
  h->sysvals.release(); h->psysvals.release(); h->valstage.release();
  h->w[0].release(); h->w[1].release(); h->wu.release();
  h->xs.release(); h->ys.release(); h->io_x.release(); h->io_y.release();
  h->V.release(); h->gz.release(); h->gw.release(); h->gparts.release();
  h->gh.release(); h->gy.release(); h->gxs.release(); h->gbs.release();
  h->gH.release(); h->gcs.release(); h->gsn.release(); h->gg.release(); h->gstat.release();
  for (auto& e : h->gev) if (e) (void)hipEventDestroy(e);
  h->loc_x.release(); h->loc_y.release();
  delete h->comm;
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  delete h;
  return 0;
} PCD_ABI_CATCH(pcd_destroy)

int pcd_set_stream(pcd_handle h, void* hip_stream) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  HIPCHK(hipStreamSynchronize(h->stream));
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  ++h->gen;
  return 0;
} PCD_ABI_CATCH(pcd_set_stream)


// Field vectors cross the ABI in the CALLER's numbering (global vectors); the
// engine renumbering of their space (rx / ry, may be null) is applied here.
// Device-pointer calls of a partitioned engine carry the rank's slice in the
// engine's own order and pass through untouched.
int fio_begin(FieldIo& f, Engine* h, const Space* spx, int64_t nx_glob, int64_t nx_loc,
                     const Space* spy, int64_t ny_glob, int64_t ny_loc,
                     const double* x, double* y, int mem, bool y_in,
                     const Reorder* rx, const Reorder* ry) {
  f.h = h; f.sp = spy; f.nglob = ny_glob; f.nloc = ny_loc;
  if (rx && !rx->active()) rx = nullptr;
  if (ry && !ry->active()) ry = nullptr;
  if (h->comm && mem == PCD_MEM_DEVICE) {         // local slices, engine order
    CHK(io_begin(h, f.io, x, nx_glob, y, ny_glob, mem, y_in));
    f.io.dx = x; f.io.dy = y;
    f.lx = f.io.dx; f.ly = f.io.dy;
    return 0;
  }
  CHK(io_begin(h, f.io, x, nx_glob, y, ny_glob, mem, y_in));
  const double* gx = f.io.dx;                     // global, caller numbering
  double* gy = f.io.dy;
  if (rx && x) {
    CHK(h->px_s.ensure(nx_glob));
    hipLaunchKernelGGL(k_gather, dim3(grid1d(nx_glob, 4)), dim3(kBlock), 0, h->stream,
                       (int)nx_glob, rx->d_n2o.p, gx, h->px_s.p);
    gx = h->px_s.p;
  }
  if (ry && y) {
    CHK(h->py_s.ensure(ny_glob));
    if (y_in)
      hipLaunchKernelGGL(k_gather, dim3(grid1d(ny_glob, 4)), dim3(kBlock), 0, h->stream,
                         (int)ny_glob, ry->d_n2o.p, gy, h->py_s.p);
    f.ry = ry; f.y_caller = gy; f.y_engine = h->py_s.p;
    gy = h->py_s.p;
  }
  HIPCHK(hipGetLastError());
  if (!h->comm) { f.lx = gx; f.ly = gy; return 0; }
  CHK(h->loc_x.ensure(nx_loc)); CHK(h->loc_y.ensure(ny_loc));
  if (x) CHK(slice_in(h, *spx, gx, h->loc_x.p));
  if (y_in) CHK(slice_in(h, *spy, gy, h->loc_y.p));
  f.lx = h->loc_x.p; f.ly = h->loc_y.p;
  if (!f.ry) f.y_engine = gy;
  return 0;
}

int fio_end(FieldIo& f) {
  Engine* h = f.h;
  if (h->comm && f.io.mem == PCD_MEM_HOST)
    CHK(slice_out(h, *f.sp, f.ly, f.ry ? f.y_engine : f.io.dy));
  if (f.ry && !(h->comm && f.io.mem == PCD_MEM_DEVICE)) {
    hipLaunchKernelGGL(k_scatter, dim3(grid1d(f.nglob, 4)), dim3(kBlock), 0, h->stream,
                       (int)f.nglob, f.ry->d_n2o.p, f.y_engine, f.y_caller);
    HIPCHK(hipGetLastError());
  }
  return io_end(f.io);
}

int pcd_apply(pcd_handle h, const double* x, double* y, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!h->ready) return fail(PCD_ERR_STATE, "apply: call pcd_setup first");
  if (!x || !y || x == y) return fail(PCD_ERR_ARG, "apply: x and y must be distinct non-null vectors");
  FieldIo f;
  CHK(fio_begin(f, h, &h->sp_p, h->n_p, h->np_loc, &h->sp_p, h->n_p, h->np_loc, x, y, mem, false,
                &h->rp, &h->rp));
  CHK(pcd_apply_dev(h, f.lx, f.ly));
  return fio_end(f);
} PCD_ABI_CATCH(pcd_apply)

int pcd_fieldsplit_apply(pcd_handle h, const double* x, double* y, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!h->ready || !h->mat[PCD_MAT_A00].set)
    return fail(PCD_ERR_STATE, "fieldsplit_apply: pcd_set_system + pcd_setup first");
  if (!x || !y || x == y) return fail(PCD_ERR_ARG, "fieldsplit_apply: x and y must be distinct non-null vectors");
  const int64_t n = h->n_u + h->n_p, nloc = h->nu_loc + h->np_loc;
  if (h->comm && mem == PCD_MEM_DEVICE)          // local split-ordered slices
    return fs_apply_split(h, x, y);
  IoMap io;
  CHK(io_begin(h, io, x, n, y, n, mem));
  const int g = grid1d(nloc, 1);
  hipLaunchKernelGGL(k_gather, dim3(g), dim3(kBlock), 0, h->stream, (int)nloc, h->perm.p, io.dx, h->xs.p);
  CHK(fs_apply_split(h, h->xs.p, h->ys.p));
  if (h->comm) HIPCHK(hipMemsetAsync(io.dy, 0, n * sizeof(double), h->stream));
  hipLaunchKernelGGL(k_scatter, dim3(g), dim3(kBlock), 0, h->stream, (int)nloc, h->perm.p, h->ys.p, io.dy);
  HIPCHK(hipGetLastError());
  if (h->comm && h->comm->allreduce(io.dy, n, h->stream))
    return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  return io_end(io);
} PCD_ABI_CATCH(pcd_fieldsplit_apply)

// The dominant kernel where it runs: `reps` EAGER fieldsplit applies (device
// vectors) with an event pair around every fused Chebyshev step on the finest
// velocity operator - the caches in the state the multigrid cycle leaves them
// in, where a back-to-back loop on one operator keeps them warm.  An event
// pair adds about a microsecond of its own.
int pcd_probe_a00_step(pcd_handle h, const double* x, double* y, int reps, double* us, int* launches) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!us || !launches || reps < 1) return fail(PCD_ERR_ARG, "probe_a00_step: bad arguments");
  const bool graph = h->graph_on;
  h->graph_on = false;
  int rc = pcd_fieldsplit_apply(h, x, y, PCD_MEM_DEVICE);        // warm
  h->probe_ev.clear();
  h->probe_on = true;
  for (int r = 0; r < reps && !rc; ++r) rc = pcd_fieldsplit_apply(h, x, y, PCD_MEM_DEVICE);
  h->probe_on = false;
  h->graph_on = graph;
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(PCD_ERR_HIP, "probe_a00_step: synchronize");
  double sum = 0.0;
  int cnt = 0;
  for (size_t i = 0; i + 1 < h->probe_ev.size() && !rc; i += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->probe_ev[i], h->probe_ev[i + 1]) == hipSuccess) { sum += 1e3 * ms; ++cnt; }
  }
  for (hipEvent_t e : h->probe_ev) (void)hipEventDestroy(e);
  h->probe_ev.clear();
  *us = cnt ? sum / cnt : 0.0;
  *launches = cnt;
  return rc;
} PCD_ABI_CATCH(pcd_probe_a00_step)

// which -> (row space, column space) of a stored operator
void mat_spaces(Engine* h, int which, const Space** rs, const Space** cs) {
  switch (which) {
    case PCD_MAT_A00: *rs = *cs = &h->sp_u; break;
    case PCD_MAT_A01: *rs = &h->sp_u; *cs = &h->sp_p; break;
    case PCD_MAT_A: *rs = *cs = &h->sp_sys; break;
    default: *rs = *cs = &h->sp_p; break;
  }
}

int pcd_spmv(pcd_handle h, int which, const double* x, double* y, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (which < 0 || which >= PCD_MAT_COUNT || !h->mat[which].set)
    return fail(PCD_ERR_STATE, "spmv: operator %d not set", which);
  if (!x || !y || x == y) return fail(PCD_ERR_ARG, "spmv: x and y must be distinct non-null vectors");
  const DCsr& A = h->mat[which];
  const Space *rs, *cs;
  mat_spaces(h, which, &rs, &cs);
  FieldIo f;
  const Reorder* rx = which == PCD_MAT_A00 ? &h->ru : which == PCD_MAT_A ? &h->rs : &h->rp;
  const Reorder* ry = (which == PCD_MAT_A00 || which == PCD_MAT_A01) ? &h->ru
                      : which == PCD_MAT_A ? &h->rs : &h->rp;
  CHK(fio_begin(f, h, cs, h->comm ? cs->total() : A.ncols, A.ncols,
                rs, h->comm ? rs->total() : A.nrows, A.nrows, x, y, mem, false, rx, ry));
  CHK(spmv(h, A, f.lx, f.ly));
  return fio_end(f);
} PCD_ABI_CATCH(pcd_spmv)

int pcd_inner_solve(pcd_handle h, int slot, const double* b, double* x, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "inner_solve: bad slot %d", slot);
  const DCsr& A = h->mat[kSlotMat[slot]];
  if (!A.set) return fail(PCD_ERR_STATE, "inner_solve: operator of slot %d not set", slot);
  if (!b || !x || b == x) return fail(PCD_ERR_ARG, "inner_solve: b and x must be distinct non-null vectors");
  CHK(inner_prepare(h, slot));
  const Space *rs, *cs;
  mat_spaces(h, kSlotMat[slot], &rs, &cs);
  const int64_t ng = h->comm ? rs->total() : A.nrows;
  FieldIo f;
  const Reorder* rr = slot == PCD_KSP_A00 ? &h->ru : &h->rp;
  CHK(fio_begin(f, h, rs, ng, A.nrows, rs, ng, A.nrows, b, x, mem, false, rr, rr));
  CHK(inner_solve(h, slot, f.lx, f.ly));
  return fio_end(f);
} PCD_ABI_CATCH(pcd_inner_solve)

int pcd_apply_bc(pcd_handle h, double* x, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!x) return fail(PCD_ERR_ARG, "apply_bc: null vector");
  if (!h->ready) return fail(PCD_ERR_STATE, "apply_bc: call pcd_setup first");
  FieldIo f;
  CHK(fio_begin(f, h, &h->sp_p, h->n_p, h->np_loc, &h->sp_p, h->n_p, h->np_loc, nullptr, x, mem, true,
                nullptr, &h->rp));
  CHK(apply_bc_dev(h, f.ly));
  return fio_end(f);
} PCD_ABI_CATCH(pcd_apply_bc)

int pcd_get_info(pcd_handle h, int key, double* out) try {
  if (!h || !out) return fail(PCD_ERR_ARG, "get_info: null argument");
  switch (key) {
    case PCD_INFO_N_U: *out = (double)h->n_u; return 0;
    case PCD_INFO_N_P: *out = (double)h->n_p; return 0;
    case PCD_INFO_ITS_AP: case PCD_INFO_ITS_MP: case PCD_INFO_ITS_RP:
    case PCD_INFO_ITS_A00: {
      Inner& s = h->inner[key - PCD_INFO_ITS_AP];
      if (s.its_on_device && s.state.p) {
        CgState st;
        HIPCHK(hipMemcpyAsync(&st, s.state.p + s.state_idx, sizeof st, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if (st.its < 0)
          return fail(PCD_ERR_BREAKDOWN, "cg (single reduction): p.Ap <= 0 at iteration %d - the "
                                         "operator of slot %d is not positive definite",
                      -st.its - 1, key - PCD_INFO_ITS_AP);
        s.last_its = st.its;
      }
      *out = (double)s.last_its;
      return 0;
    }
    case PCD_INFO_NUM_PCD_APPLY: *out = (double)h->num_pcd; return 0;
    case PCD_INFO_NUM_FS_APPLY: *out = (double)h->num_fs; return 0;
    case PCD_INFO_GMRES_ITS: *out = (double)h->gmres_its; return 0;
    case PCD_INFO_GMRES_RNORM: *out = h->gmres_rnorm; return 0;
    case PCD_INFO_N_U_LOCAL: *out = (double)h->nu_loc; return 0;
    case PCD_INFO_N_P_LOCAL: *out = (double)h->np_loc; return 0;
    case PCD_INFO_A00_COMPONENTS: *out = (double)h->mat[PCD_MAT_A00].kron; return 0;
    case PCD_INFO_RANKS: *out = h->comm ? (double)h->nranks : 0.0; return 0;
    case PCD_INFO_REORDERED: *out = (h->ru.active() ? 1.0 : 0.0) + (h->rp.active() ? 2.0 : 0.0); return 0;
    case PCD_INFO_LAUNCHES: *out = (double)launch_count(); return 0;
    case PCD_INFO_A00_MODEL_BYTES: {
      // what one launch of the fused Chebyshev step on A00 moves by construction
      const DCsr& A = h->mat[PCD_MAT_A00];
      const double vec = 40.0 * (double)A.nrows;        // b, D^-1, p_k, p_{k-1} read, p_{k+1} written
      if (A.kron && A.vt && A.vt_lm)
        *out = 10.0 * (double)A.vt_slots + 4.0 * (double)A.vt_nsrc +
               (16.0 + 2.0 * (A.vt_rows + 2)) * (double)A.vt_blocks + vec;
      else if (A.kron && A.vt)
        *out = 10.0 * (double)A.nnz2 + 4.0 * (double)A.vt_nsrc +
               (16.0 + 2.0 * (A.vt_rows + 2)) * (double)A.vt_blocks + vec;
      else if (A.kron)
        *out = 12.0 * (double)A.nnz2 + 4.0 * ((double)A.nrows / A.kron + 1.0) + vec;
      else
        *out = 12.0 * (double)A.nnz + 4.0 * ((double)A.nrows + 1.0) + vec;
      return 0;
    }
    case PCD_INFO_PEER_CALLS:
      *out = (h->comm && h->comm->peer()) ? (double)static_cast<PeerBackend*>(h->comm)->peer_calls : 0.0;
      return 0;
    case PCD_INFO_A00_KERNEL: {
      const DCsr& A = h->mat[PCD_MAT_A00];
      *out = (A.kron && A.vt) ? (A.vt_lm ? 4.0 : 3.0) : (A.kron && A.rb2) ? 2.0 : A.rb ? 1.0 : 0.0;
      return 0;
    }
    case PCD_INFO_PEER_DECLINED:
      *out = (h->comm && h->comm->peer()) ? (double)static_cast<PeerBackend*>(h->comm)->declined : 0.0;
      return 0;
    case PCD_INFO_BOOT_CALLS:
      *out = (double)h->boot_exchanges +
             ((h->comm && h->comm->peer()) ? (double)static_cast<PeerBackend*>(h->comm)->boot_calls : 0.0);
      return 0;
    case PCD_INFO_A00_ROWS_PER_WG:
      *out = (double)(h->mat[PCD_MAT_A00].kron ? h->mat[PCD_MAT_A00].rb2 : h->mat[PCD_MAT_A00].rb);
      if (h->mat[PCD_MAT_A00].kron && h->mat[PCD_MAT_A00].vt) *out = -(double)h->mat[PCD_MAT_A00].vt_rows;
      return 0;
    default:
      if (key >= PCD_INFO_NNZ_BASE && key < PCD_INFO_NNZ_BASE + PCD_MAT_COUNT) {
        const DCsr& A = h->mat[key - PCD_INFO_NNZ_BASE];
        *out = (double)(h->comm ? A.gnnz : A.nnz);
        return 0;
      }
  }
  return fail(PCD_ERR_ARG, "get_info: unknown key %d", key);
} PCD_ABI_CATCH(pcd_get_info)

int pcd_set_velocity_block(pcd_handle h, int ncomp) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (ncomp < 1 || ncomp > 3) return fail(PCD_ERR_ARG, "set_velocity_block: 1..3 components");
  for (auto& m : h->mat) if (m.set) return fail(PCD_ERR_STATE, "set_velocity_block: call before any operator is handed over");
  h->vel_block = ncomp;
  return 0;
} PCD_ABI_CATCH(pcd_set_velocity_block)

// Streaming bandwidth of this GPU as a kernel of this library sees it:
// kind 0 copy, 1 triad, 2 read-only, 3 read-mostly (6 % writes), 4 read-only
// with non-temporal loads, on arrays of
// `bytes` each (>= 256 MiB: beyond the
// Infinity Cache), best of `reps` launches, timed with events on the engine's
// stream.  *gbs = bytes moved (reads + writes) per second / 1e9.
int pcd_bandwidth_probe(pcd_handle h, int kind, int64_t bytes, int reps, double* gbs) try {
  if (!h || !gbs) return fail(PCD_ERR_ARG, "bandwidth_probe: null argument");
  if (kind < 0 || kind > 4 || bytes < 4096 || reps < 1) return fail(PCD_ERR_ARG, "bandwidth_probe: bad arguments");
  HIPCHK(hipSetDevice(h->device));
  const int64_t n2 = bytes / 16;
  DBuf<double> a, b, c;
  // (read kinds write n2/16 + one 16-byte word per thread into `a`)
  CHK(a.ensure(kind >= 2 ? 2 * (n2 / 16 + (int64_t)g_num_cus * 8 * kBlock + 16) : 2 * n2));
  CHK(b.ensure(2 * n2));
  if (kind == 1) CHK(c.ensure(2 * n2));
  HIPCHK(hipMemsetAsync(b.p, 0, 16 * n2, h->stream));
  if (kind == 1) HIPCHK(hipMemsetAsync(c.p, 0, 16 * n2, h->stream));
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  // copy / triad: grid-stride, 32 workgroups per CU; read sweeps: one chunk per
  // workgroup, best of 2, 4 and 8 workgroups per CU
  const int grids_rw[1] = {g_num_cus * 32};
  const int grids_rd[3] = {g_num_cus * 2, g_num_cus * 4, g_num_cus * 8};   // (1 per CU never won the sweep)
  const int* grids = kind >= 2 ? grids_rd : grids_rw;
  const int ngrids = kind >= 2 ? 3 : 1;
  const double moved = (kind == 1 ? 3.0 : kind == 0 ? 2.0 : kind == 3 ? 1.0 + 1.0 / 16.0 : 1.0) * 16.0 * (double)n2;
  double best = 0.0;
  for (int gi = 0; gi < ngrids; ++gi)
    for (int r = 0; r < reps + 1; ++r) {
      HIPCHK(hipEventRecord(e0, h->stream));
      const double2* bp = reinterpret_cast<const double2*>(b.p);
      double2* ap = reinterpret_cast<double2*>(a.p);
      if (kind == 2) hipLaunchKernelGGL(k_bw_read<2>, dim3(grids[gi]), dim3(kBlock), 0, h->stream, n2, bp, ap);
      else if (kind == 3) hipLaunchKernelGGL(k_bw_read<3>, dim3(grids[gi]), dim3(kBlock), 0, h->stream, n2, bp, ap);
      else if (kind == 4) hipLaunchKernelGGL(k_bw_read<4>, dim3(grids[gi]), dim3(kBlock), 0, h->stream, n2, bp, ap);
      else hipLaunchKernelGGL(k_bw_probe, dim3(grids[gi]), dim3(kBlock), 0, h->stream, kind, n2, bp,
                              reinterpret_cast<const double2*>(c.p), 3.0, ap);
      HIPCHK(hipEventRecord(e1, h->stream));
      HIPCHK(hipEventSynchronize(e1));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0 && ms > 0.f) best = std::max(best, moved / (ms * 1e-3) / 1e9);
    }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  a.release(); b.release(); c.release();
  *gbs = best;
  return 0;
} PCD_ABI_CATCH(pcd_bandwidth_probe)

int pcd_graph_enable(pcd_handle h, int on) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  h->graph_on = on != 0;
  return 0;
} PCD_ABI_CATCH(pcd_graph_enable)

// ---- multi-GPU bootstrap ---------------------------------------------------
int pcd_comm_unique_id(void* out128) try {
  if (!out128) return fail(PCD_ERR_ARG, "comm_unique_id: null buffer");
  std::string err;
  if (!rccl_api().load(err)) return fail(PCD_ERR_COMM, "%s", err.c_str());
  ncclUniqueId id;
  ncclResult_t r = rccl_api().GetUniqueId(&id);
  if (r != ncclSuccess) return fail(PCD_ERR_COMM, "ncclGetUniqueId: %s", rccl_api().GetErrorString(r));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, sizeof id);
  return 0;
} PCD_ABI_CATCH(pcd_comm_unique_id)

int comm_attach(Engine* h, CommBackend* c, int rank, int nranks) {
  for (auto& m : h->mat) if (m.set) { delete c; return fail(PCD_ERR_STATE, "comm_init: call before any operator is handed over"); }
  delete h->comm;
  c->rank = rank; c->nranks = nranks;
  h->comm = c; h->rank = rank; h->nranks = nranks;
  h->sp_u = Space(); h->sp_p = Space(); h->sp_sys = Space();
  ++h->gen;
  return 0;
}

// Put the one-shot peer-write protocol (pcd_peer.hpp) in front of a bootstrap
// backend.  PCD_COMM_PEER: "0" never, "1" always; default: on for one process
// per GPU (RCCL bootstrap), off for the thread ranks of the single-GPU tests
// (their ranks share the legacy stream unless the caller gives each one its
// own - a kernel that waits for another rank's kernel must not sit in front
// of it in one in-order stream).  If the arenas cannot be set up (no IPC
// between the devices) every rank falls back to the bootstrap backend alone:
// the decision is an all-reduce.
static void thread_group_barrier(void* g) { (void)static_cast<ThreadGroup*>(g)->barrier(); }
CommBackend* wrap_peer(Engine* h, CommBackend* boot, int rank, int nranks, ThreadGroup* tg,
                              bool default_on) {
  const char* e = getenv("PCD_COMM_PEER");
  const bool want = e ? e[0] == '1' : default_on;
  if (!want || nranks > kPeerMaxPeers) return boot;
  size_t cap = 512ull << 20;
  if (const char* m = getenv("PCD_PEER_ARENA_MB")) cap = (size_t)std::max(1ll, atoll(m)) << 20;
  PeerBackend* pb = new PeerBackend();
  pb->boot = boot; pb->rank = rank; pb->nranks = nranks;
  boot->rank = rank; boot->nranks = nranks;
  if (const char* t = getenv("PCD_PEER_TIMEOUT_S")) pb->spin_limit = (long long)(atof(t) * 1.0e8);
  int bad = pb->init(cap, tg ? tg->arenas.data() : nullptr, thread_group_barrier, tg, h->stream);
  // self-test before anything depends on it: one peer all-reduce of a known
  // vector with a short time-out - remote stores, flags and the mapped
  // arenas of EVERY pair of ranks are exercised once; a platform where that
  // does not work (no peer access between two devices, ...) falls back to
  // the bootstrap backend on all ranks instead of failing in the first solve.
  // (Thread ranks on the legacy stream cannot run it: their kernels would
  // queue behind each other - they only use the protocol with own streams.)
  if (!bad && pb->usable(h->stream)) {
    const long long keep = pb->spin_limit;
    pb->spin_limit = 500000000ll;                        // 5 s
    double probe[2] = {(double)(rank + 1), 1.0};
    double* dp = nullptr;
    if (hipMalloc((void**)&dp, sizeof probe) != hipSuccess) bad = 1;
    if (!bad) {
      (void)hipMemcpyAsync(dp, probe, sizeof probe, hipMemcpyHostToDevice, h->stream);
      if (pb->allreduce(dp, 2, h->stream)) bad = 1;
      (void)hipMemcpyAsync(probe, dp, sizeof probe, hipMemcpyDeviceToHost, h->stream);
      if (hipStreamSynchronize(h->stream) != hipSuccess) bad = 1;
      if (pb->take_error(h->stream)) bad = 1;
      if (probe[0] != 0.5 * nranks * (nranks + 1) || probe[1] != (double)nranks) {
        bad = 1;
        if (pb->err.empty()) pb->err = "peer self-test: wrong sum";
      }
      (void)hipFree(dp);
    }
    pb->spin_limit = keep;
    pb->peer_calls = 0;
  }
  // every rank must have its arena and its mappings, or nobody uses them
  double flag = bad ? 1.0 : 0.0;
  double* dflag = nullptr;
  if (hipMalloc((void**)&dflag, sizeof(double)) == hipSuccess) {
    (void)hipMemcpyAsync(dflag, &flag, sizeof flag, hipMemcpyHostToDevice, h->stream);
    if (boot->allreduce(dflag, 1, h->stream) == 0) {
      (void)hipMemcpyAsync(&flag, dflag, sizeof flag, hipMemcpyDeviceToHost, h->stream);
      (void)hipStreamSynchronize(h->stream);
    } else flag = 1.0;
    (void)hipFree(dflag);
  } else flag = 1.0;
  if (flag != 0.0) {
    if (getenv("PCD_COMM_VERBOSE"))
      fprintf(stderr, "[pcd comm] rank %d: peer protocol unavailable (%s); bootstrap backend only\n",
              rank, pb->err.c_str());
    pb->boot = nullptr;                  // (keep the bootstrap backend alive)
    delete pb;
    return boot;
  }
  return pb;
}

int pcd_comm_init(pcd_handle h, int rank, int nranks, const void* id) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (nranks < 1 || rank < 0 || rank >= nranks || !id) return fail(PCD_ERR_ARG, "comm_init: bad rank/size/id");
  // one rank: nothing to partition - unless PCD_FORCE_COMM=1 asks for the
  // communicator anyway (single-rank RCCL smoke test of the multi-rank code)
  { const char* e = getenv("PCD_FORCE_COMM");
    if (nranks == 1 && !(e && e[0] == '1')) return 0; }
  HIPCHK(hipSetDevice(h->device));
  std::string err;
  if (!rccl_api().load(err)) return fail(PCD_ERR_COMM, "%s", err.c_str());
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  RcclBackend* b = new RcclBackend();
  ncclResult_t r = rccl_api().CommInitRank(&b->comm, nranks, uid, rank);
  if (r != ncclSuccess) { delete b; return fail(PCD_ERR_COMM, "ncclCommInitRank: %s", rccl_api().GetErrorString(r)); }
  for (auto& m : h->mat) if (m.set) { delete b; return fail(PCD_ERR_STATE, "comm_init: call before any operator is handed over"); }
  return comm_attach(h, wrap_peer(h, b, rank, nranks, nullptr, true), rank, nranks);
} PCD_ABI_CATCH(pcd_comm_init)

int pcd_comm_init_host(pcd_handle h, int rank, int nranks, pcd_host_allreduce_fn allreduce,
                       pcd_host_exchange_fn exchange, void* ctx) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (nranks < 2 || rank < 0 || rank >= nranks || !allreduce || !exchange)
    return fail(PCD_ERR_ARG, "comm_init_host: bad rank / size / callbacks");
  for (auto& m : h->mat) if (m.set) return fail(PCD_ERR_STATE, "comm_init: call before any operator is handed over");
  HIPCHK(hipSetDevice(h->device));
  HostBackend* b = new HostBackend();
  b->ar = allreduce; b->ex = exchange; b->ctx = ctx;
  return comm_attach(h, wrap_peer(h, b, rank, nranks, nullptr, true), rank, nranks);
} PCD_ABI_CATCH(pcd_comm_init_host)

// test-only backend: `nranks` engines of ONE process (one thread each) on one
// GPU exchange through device copies; *group is created by the first caller
int pcd_comm_init_threads(pcd_handle h, int rank, int nranks, void** group) try {
  if (!h || !group) return fail(PCD_ERR_ARG, "comm_init_threads: null argument");
  if (nranks < 2 || rank < 0 || rank >= nranks) return fail(PCD_ERR_ARG, "comm_init_threads: bad rank/size");
  {
    static std::mutex mu;                 // the ranks may arrive concurrently
    std::lock_guard<std::mutex> lk(mu);
    if (!*group) *group = new ThreadGroup(nranks);
  }
  ThreadBackend* b = new ThreadBackend();
  b->g = static_cast<ThreadGroup*>(*group);
  for (auto& m : h->mat) if (m.set) { delete b; return fail(PCD_ERR_STATE, "comm_init: call before any operator is handed over"); }
  HIPCHK(hipSetDevice(h->device));
  return comm_attach(h, wrap_peer(h, b, rank, nranks, b->g, false), rank, nranks);
} PCD_ABI_CATCH(pcd_comm_init_threads)

// Host-only view of the partitioning (no device call): the row block, the
// localised columns and the halo plan rank `rank` of `nranks` derives from a
// global CSR.  Lets multi-process CPU tests drive the same C++ that the GPU
// path uses.  Output arrays must hold nrows+1 / nnz / ncols / nranks+1 entries.
int pcd_dist_probe(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                   const int32_t* colidx, const double* vals, int rank,
                   int nranks, int even_rows, int even_cols, int64_t* counts,
                   int32_t* out_rowptr, int32_t* out_col, double* out_val,
                   int32_t* send_peers, int32_t* send_off, int32_t* send_idx,
                   int32_t* recv_peers, int32_t* recv_off) try {
  if (!rowptr || !colidx || !vals || !counts || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(PCD_ERR_ARG, "dist_probe: bad arguments");
  const Space rs = Space::field(nrows, nranks, even_rows ? (even_rows > 1 ? even_rows : 2) : 1);
  const Space cs = Space::field(ncols, nranks, even_cols ? (even_cols > 1 ? even_cols : 2) : 1);
  std::vector<int32_t> orp, oc; std::vector<double> ov; std::vector<int64_t> osrc;
  HaloPlan plan;
  localize(rs, cs, rank, nranks, rowptr, colidx, vals, nullptr, orp, oc, ov, osrc, plan);
  counts[0] = rs.nloc(rank); counts[1] = cs.nloc(rank); counts[2] = plan.nghost;
  counts[3] = (int64_t)plan.peers_send.size(); counts[4] = (int64_t)plan.peers_recv.size();
  counts[5] = rs.bounds[0][rank]; counts[6] = cs.bounds[0][rank];
  std::copy(orp.begin(), orp.end(), out_rowptr);
  std::copy(oc.begin(), oc.end(), out_col);
  std::copy(ov.begin(), ov.end(), out_val);
  std::copy(plan.peers_send.begin(), plan.peers_send.end(), send_peers);
  std::copy(plan.send_off.begin(), plan.send_off.end(), send_off);
  std::copy(plan.send_idx.begin(), plan.send_idx.end(), send_idx);
  std::copy(plan.peers_recv.begin(), plan.peers_recv.end(), recv_peers);
  std::copy(plan.recv_off.begin(), plan.recv_off.end(), recv_off);
  return 0;
} PCD_ABI_CATCH(pcd_dist_probe)

