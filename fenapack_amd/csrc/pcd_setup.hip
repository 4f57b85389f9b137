// pcd_setup.hip - operators cross the ABI: hand-over, structure detection, tile layouts, hierarchies
// (one of the engine's translation units; shared declarations: pcd_internal.hpp)
#include "pcd_internal.hpp"


int io_begin(Engine* h, IoMap& io, const double* x, size_t nx, double* y,
                    size_t ny, int mem, bool y_in) {
  io.h = h; io.mem = mem; io.ny = ny;
  if (mem == PCD_MEM_DEVICE) { io.dx = x; io.dy = y; return 0; }
  if (mem != PCD_MEM_HOST) return fail(PCD_ERR_ARG, "bad mem flag %d", mem);
  if (x) {
    CHK(h->io_x.ensure(nx));
    HIPCHK(hipMemcpyAsync(h->io_x.p, x, nx * sizeof(double), hipMemcpyHostToDevice, h->stream));
    io.dx = h->io_x.p;
  }
  if (y) {
    CHK(h->io_y.ensure(ny));
    if (y_in)
      HIPCHK(hipMemcpyAsync(h->io_y.p, y, ny * sizeof(double), hipMemcpyHostToDevice, h->stream));
    io.dy = h->io_y.p; io.hy = y;
  }
  return 0;
}

int io_end(IoMap& io) {
  if (io.mem == PCD_MEM_DEVICE) return 0;
  Engine* h = io.h;
  if (io.hy)
    HIPCHK(hipMemcpyAsync(io.hy, io.dy, io.ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}


// After new values arrived: refresh F's values and verify that all
// components still carry the same numbers; otherwise drop to the general path.
int refresh_kron(Engine* h, DCsr& A) {
  if (A.rk && A.rk_nnz)          // row-blocked values follow `val`
    hipLaunchKernelGGL(k_lm_values, dim3(grid1d(A.rk_nnz * A.rk, 4)), dim3(kBlock), 0, h->stream,
                       A.rk_nnz * A.rk, A.rk_pos.p, A.val.p, A.rk_val.p);
  if (!A.kron_pat || !A.nnz2) return 0;
  HIPCHK(hipMemsetAsync(A.kron_flag.p, 0, sizeof(int), h->stream));
  hipLaunchKernelGGL(k_kron_gather, dim3(grid1d(A.nnz2, 4)), dim3(kBlock), 0,
                     h->stream, (int)A.nnz2, A.kron_pat, A.kron_pos.p, A.val.p,
                     A.val2.p, A.kron_flag.p);
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, A.kron_flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  // values differ between components: general path - until a later refresh
  // brings equal components back.  Either way the set of kernels a PCApply
  // launches changes, so a captured graph is stale.
  if (A.vt && A.vt_lm)
    hipLaunchKernelGGL(k_lm_values, dim3(grid1d(A.vt_slots, 4)), dim3(kBlock), 0, h->stream,
                       A.vt_slots, A.vt_pos.p, A.val2.p, A.vt_val.p);
  const int now = flag ? 0 : A.kron_pat;
  if (now != A.kron) { A.kron = now; ++h->gen; }
  return 0;
}

// pattern test for A = F (x) I_nc on interleaved dofs
bool kron_pattern(int nc, int64_t nrows, int64_t ncols, const int32_t* rowptr,
                         const int32_t* col) {
  if (nrows < nc || nrows % nc || ncols % nc) return false;
  std::atomic<bool> ok{true};
  parallel_chunks(nrows / nc, [&](int64_t s0, int64_t s1) {
    for (int64_t s = s0; s < s1 && ok.load(std::memory_order_relaxed); ++s) {
      const int32_t a = rowptr[nc * s], len = rowptr[nc * s + 1] - a;
      bool good = true;
      for (int c = 1; c < nc && good; ++c)
        if (rowptr[nc * s + c + 1] - rowptr[nc * s + c] != len) good = false;
      for (int32_t k = 0; k < len && good; ++k) {
        if (col[a + k] % nc) good = false;
        for (int c = 1; c < nc && good; ++c)
          if (col[rowptr[nc * s + c] + k] != col[a + k] + c) good = false;
      }
      if (!good) ok.store(false, std::memory_order_relaxed);
    }
  });
  return ok.load();
}

// LDS-staged vector tiles (pcd_kernels.hpp): row blocks of the scalar stencil
// F chosen greedily - rows are added while the block's distinct columns fit
// the tile, its entries the lanes' registers (lane-major form) and its rows
// the workgroup -, per block its distinct columns in ascending order
// (the tile's sources) and per entry the offset of its column in the tile.  PCD_VEC_TILE: 0 off, 1 three-component operators
// of at least PCD_VEC_TILE_ROWS node rows (default), 2 every F (x) I operator.
static int g_vec_tile = 1;
static long long g_vec_tile_rows = 80000;
// rows per block of the direct form: 64 (cavity level 6, cache-resident:
// gather kernel 18.9 us, tile 15.5 with 64 rows, 17.7 with 128 -
// profiles/r04_r_vt_sweep_2d.txt); the lane-major form: lm_rows()
int build_vec_tile(Engine* h, DCsr& A, int nc, int64_t nn, int64_t nloc,
                          const std::vector<int32_t>& rpc, const std::vector<int32_t>& cc) {
  A.vt = false; A.vt_blocks = 0;
  // (read per operator, defaults restored when a variable is gone: the A/B
  // tests of one process must not leak their switches into later engines)
  { const char* e = getenv("PCD_VT_NT"); g_vt_nt = e ? atoi(e) : 1; }
  // operators streamed from HBM: lane-major entries, straight to registers
  // (the form is fixed with the layout)
  A.vt_lm = A.nt2 && g_vt_nt;
  // Operators with LONG rows (the coarse levels of an algebraic hierarchy: ~180
  // entries per node row on config 5's first level).  The direct form gives a
  // row 256 / ROWS lanes whatever the block holds, and a block of such rows is
  // cut by its tile nodes after 8-16 rows: with ROWS = 64 only 32-64 of the
  // 256 lanes work (N = 73, first coarse level, counter pass of round 6:
  // 112 MB in 58-78 us = 0.18-0.25 of the HBM peak).  Such operators take
  // blocks of <= 16 rows, 16 lanes each (the ROWS = 16 instantiation of the
  // same kernels).  PCD_VT_LONG_ROW: mean entries per node row from which
  // (default 64; 0: never).  (The lane-major form was tried for them first:
  // its 512 tile nodes do not hold every single row of such a level, and an
  // operator whose tiles cannot be built falls to the scalar kernels - 3 x
  // the bytes: 2.24 -> 2.40 ms per PCApply, profiles/r06_c_*.)
  int long_rows16 = 0;
  if (!A.vt_lm) {
    const char* e = getenv("PCD_VT_LONG_ROW");
    const long long thr = e ? atoll(e) : 64;
    if (thr > 0 && nn > 0 && (long long)(rpc[nn] / nn) >= thr) long_rows16 = 1;
  }
  const int kVtRows = A.vt_lm ? lm_rows(nc) : (long_rows16 ? 16 : 64);
  const int kEntries = A.vt_lm ? kLmEntries : INT32_MAX;       // (direct form: no entry buffer)
  const int kNodes = A.vt_lm ? lm_nodes(nc) : kVtNodes;
  const int kVtRowOff = vt_rowoff(kVtRows);
  A.vt_rows = kVtRows;
  { const char* e = getenv("PCD_VEC_TILE"); g_vec_tile = e ? atoi(e) : 1; }
  { const char* e = getenv("PCD_VEC_TILE_ROWS"); g_vec_tile_rows = e ? atoll(e) : 80000; }
  // default: F (x) I operators from 80 000 node rows.  Timed alone (launched
  // back to back, operator warm in L2) cavity level 5 - 103 k node rows - runs
  // 5.9 us with both kernels, the latency floor; INSIDE the cycle, where the
  // finest level has pushed it out of L2, its five launches take 9-11 us with
  // the gather kernels and the tile kernels' 10 instead of 12 bytes per entry
  // count: level-6 PCApply 0.3093 -> 0.2989 ms with tiles on level 5 as well,
  // 0.3014 with level 4 (26 k node rows) too (profiles/r04_k_*).
  // Measured, k_cheb_step on the finest A00 of the unit cube, us per launch
  // (profiles/r04_q_vt_sweep*.txt, r04_e_*, r05_a_* ... r05_i_*): N = 32 (90 MB
  // per launch, cache-resident) gather kernel 34.9, tile kernel (direct form)
  // 27.3; N = 48 (428 MB per launch, HBM-bound) 110.8 against 68.5 in the
  // lane-major form (85.4 with the entries staged in LDS, round 4; 123.9 in the
  // direct form).
  if (!g_vec_tile || (g_vec_tile == 1 && nn < g_vec_tile_rows)) return 0;
  if (nn < 1 || A.dense2 || A.long_rows || A.wave_rows) return 0;
  // independent super-blocks of rows: block boundaries restart at multiples of
  // kSuper rows, so the host threads need no hand-over and the result does not
  // depend on their number
  constexpr int64_t kSuper = 8192;
  const int64_t nsup = (nn + kSuper - 1) / kSuper;
  struct Blk { int32_t r0, nr, k0, tn; };
  std::vector<std::vector<Blk>> b_desc(nsup);
  std::vector<std::vector<int32_t>> b_src(nsup);             // tile sources, block after block
  std::vector<unsigned short> loc(cc.size());
  std::atomic<bool> ok{true};
  int32_t ncol_all = 0;
  for (int32_t c : cc) ncol_all = std::max(ncol_all, c + 1);
  parallel_chunks(nsup, [&](int64_t s0, int64_t s1) {
    std::vector<int32_t> uniq;
    // distinct columns of the growing block are counted with a stamp per
    // column (one pass over the entries; the columns are sorted once per
    // block, not once per row)
    std::vector<int32_t> stamp(ncol_all, -1), slot(ncol_all, 0);
    int32_t tick = 0;
    for (int64_t sb = s0; sb < s1 && ok.load(std::memory_order_relaxed); ++sb) {
      const int64_t ra = sb * kSuper, rz = std::min<int64_t>(nn, ra + kSuper);
      int64_t r = ra;
      while (r < rz) {
        ++tick;
        int64_t r1 = r;
        int32_t nuniq = 0;
        uniq.clear();
        while (r1 < rz && r1 - r < kVtRows) {
          if (rpc[r1 + 1] - rpc[r] > kEntries) break;                 // (lane-major form: 8 per lane)
          int32_t add = 0;
          const size_t before = uniq.size();
          for (int32_t k = rpc[r1]; k < rpc[r1 + 1]; ++k)
            if (stamp[cc[k]] != tick) { stamp[cc[k]] = tick; uniq.push_back(cc[k]); ++add; }
          if (nuniq + add > kNodes) {
            for (size_t q = before; q < uniq.size(); ++q) stamp[uniq[q]] = -1;     // undo the row
            uniq.resize(before);
            break;
          }
          nuniq += add;
          ++r1;
        }
        if (r1 == r) { ok.store(false); return; }       // one row alone does not fit
        std::sort(uniq.begin(), uniq.end());
        b_src[sb].insert(b_src[sb].end(), uniq.begin(), uniq.end());
        b_desc[sb].push_back(Blk{(int32_t)r, (int32_t)(r1 - r), rpc[r], (int32_t)uniq.size()});
        for (size_t q = 0; q < uniq.size(); ++q) slot[uniq[q]] = (int32_t)q;
        for (int32_t k = rpc[r]; k < rpc[r1]; ++k) loc[k] = (unsigned short)slot[cc[k]];
        r = r1;
      }
    }
  });
  if (!ok.load()) return 0;
  std::vector<int4> desc;
  std::vector<int32_t> tsrc;
  std::vector<unsigned short> rowoff;
  int64_t lm_lanes = 0;
  std::vector<int32_t> lm_k0;
  bool lm_empty_row = false;
  for (int64_t sb = 0; sb < nsup; ++sb) {
    for (const Blk& b : b_desc[sb]) {
      if (tsrc.size() + b.tn > (size_t)INT32_MAX) return 0;
      if (A.vt_lm) {
        // y = lanes of all blocks before this one, w = rows | tile nodes << 9 | lanes << 20
        const int ne = rpc[b.r0 + b.nr] - b.k0, L = (ne + kLmE - 1) / kLmE;
        desc.push_back(int4{b.r0, (int)lm_lanes, (int)tsrc.size(), b.nr | (b.tn << 9) | (L << 20)});
        lm_k0.push_back(b.k0);
        lm_lanes += L;
        for (int i = 0; i < b.nr; ++i) lm_empty_row |= rpc[b.r0 + i + 1] == rpc[b.r0 + i];
      } else
      desc.push_back(int4{b.r0, b.k0, (int)tsrc.size(), b.nr | (b.tn << 8)});
      tsrc.resize(tsrc.size() + b.tn);
      const size_t at = rowoff.size();
      rowoff.resize(at + kVtRowOff, 0);
      for (int i = 0; i <= b.nr; ++i) rowoff[at + i] = (unsigned short)(rpc[b.r0 + i] - b.k0);
      for (int i = b.nr + 1; i < kVtRowOff; ++i) rowoff[at + i] = rowoff[at + b.nr];
    }
  }
  {
    size_t at = 0;
    for (int64_t sb = 0; sb < nsup; ++sb) {
      std::copy(b_src[sb].begin(), b_src[sb].end(), tsrc.begin() + at);
      at += b_src[sb].size();
    }
  }
  if (A.vt_lm) {
    // tile sources at a fixed stride per block (padded with -1): their address
    // depends on the block index alone
    const int TN = lm_nodes(nc);
    std::vector<int32_t> fixed(desc.size() * (size_t)TN, -1);
    for (size_t j = 0; j < desc.size(); ++j) {
      const int tn = (desc[j].w >> 9) & 0x7ff;
      std::copy(tsrc.begin() + desc[j].z, tsrc.begin() + desc[j].z + tn, fixed.begin() + j * (size_t)TN);
      desc[j].z = (int)(j * (size_t)TN);
    }
    if (fixed.size() > (size_t)INT32_MAX) return 0;
    tsrc.swap(fixed);
  }
  // rows of hundreds of entries leave a handful of rows per block: most lanes
  // of the row-sum phase idle and the tile is loaded for nothing
  if ((double)nn < 8.0 * (double)desc.size()) return 0;
  A.vt_blocks = (int)desc.size();
  A.vt_nsrc = (int64_t)tsrc.size();
  if (const char* e = getenv("PCD_VEC_TILE_STATS")) if (e[0] == '1') {
    int full = 0;
    for (const int4& d : desc) full += (d.w & 0xff) == kVtRows;
    fprintf(stderr, "[pcd vec tile] %lld node rows x %d comps: %zu blocks, %.1f rows, %.0f tile nodes "
                    "per block on average; %.0f %% of the blocks full (%d rows)\n",
            (long long)nn, nc, desc.size(), (double)nn / desc.size(), (double)tsrc.size() / desc.size(),
            100.0 * full / desc.size(), kVtRows);
  }
  if (A.vt_lm) {
    // (the lanes find their rows by counting row ends: no empty rows; 32-bit slots)
    if (lm_empty_row || lm_lanes * kLmE > (int64_t)INT32_MAX - kLmE) { A.vt_blocks = 0; A.vt_lm = false; return 0; }
    A.vt_slots = lm_lanes * kLmE;
    std::vector<unsigned short> lloc((size_t)A.vt_slots, 0);
    std::vector<int32_t> lpos((size_t)A.vt_slots, -1);
    parallel_chunks((int64_t)desc.size(), [&](int64_t j0, int64_t j1) {
      for (int64_t j = j0; j < j1; ++j) {
        const int4 d = desc[j];
        const int nr = d.w & 0x1ff, L = (d.w >> 20) & 0x1ff, k0 = lm_k0[j];
        const int ne = rpc[d.x + nr] - k0;
        const size_t base = (size_t)d.y * kLmE;
        int row = 0;                                   // block row of entry e
        for (int e = 0; e < ne; ++e) {
          while (rpc[d.x + row + 1] - k0 <= e) ++row;
          const int t = e / kLmE, u = e % kLmE;
          unsigned short w = loc[k0 + e];              // (< 2048: 11 bits)
          if (e + 1 == rpc[d.x + row + 1] - k0) w |= 0x8000u;
          if (u == 0) w |= (unsigned short)((row & 0xf) << 11);
          lloc[base + (size_t)t * kLmE + u] = w;
          lpos[base + ((size_t)(u / 2) * L + t) * 2 + (u % 2)] = k0 + e;
        }
        // (the high nibble of a lane's first row travels with its entry 1,
        // which may be padding)
        for (int t = 0; t < L; ++t) {
          const int e = t * kLmE;
          int r = 0;
          while (rpc[d.x + r + 1] - k0 <= e) ++r;
          lloc[base + (size_t)t * kLmE + 1] |= (unsigned short)(((r >> 4) & 0xf) << 11);
        }
      }
    });
    loc.swap(lloc);
    CHK(A.vt_pos.ensure(lpos.size())); CHK(A.vt_val.ensure(lpos.size()));
    HIPCHK(hipMemcpy(A.vt_pos.p, lpos.data(), lpos.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  {
    // blocks that read a ghost column (tile source >= the owned nodes) last
    std::vector<int> inner, outer;
    for (size_t j = 0; j < desc.size(); ++j) {
      const int tn = A.vt_lm ? (desc[j].w >> 9) & 0x7ff : desc[j].w >> 8;
      // (tile sources ascend: the last one tells)
      const bool bnd = tn > 0 && tsrc[(size_t)desc[j].z + tn - 1] >= nloc;
      (bnd ? outer : inner).push_back((int)j);
    }
    A.vt_nint = (int)inner.size(); A.vt_nbnd = (int)outer.size();
    if (A.vt_nbnd) {
      inner.insert(inner.end(), outer.begin(), outer.end());
      CHK(A.vt_list.ensure(inner.size()));
      HIPCHK(hipMemcpy(A.vt_list.p, inner.data(), inner.size() * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  CHK(A.vt_desc.ensure(desc.size())); CHK(A.vt_rowoff.ensure(rowoff.size()));
  CHK(A.vt_tsrc.ensure(tsrc.size())); CHK(A.vt_loc.ensure(loc.size() + 8));
  HIPCHK(hipMemcpy(A.vt_desc.p, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(A.vt_rowoff.p, rowoff.data(), rowoff.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(A.vt_tsrc.p, tsrc.data(), tsrc.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(A.vt_loc.p, loc.data(), loc.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  A.vt = true;
  return 0;
}

// detect the structure (the velocity block size first) + compressed arrays
int detect_kron(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                       const int32_t* rowptr, const int32_t* col, bool have_vals) {
  A.kron = 0; A.kron_pat = 0; A.rb2 = 0; A.nnz2 = 0; A.dense2 = false;
  if (g_no_kron || rowptr[nrows] == 0) return 0;
  int nc = 0;
  const int first = h->vel_block == 3 ? 3 : 2;
  for (int cand : {first, 5 - first})
    if (kron_pattern(cand, nrows, ncols, rowptr, col)) { nc = cand; break; }
  if (!nc) return 0;
  const int64_t nn = nrows / nc;
  std::vector<int32_t> rpc(nn + 1, 0);
  for (int64_t s = 0; s < nn; ++s) rpc[s + 1] = rpc[s] + (rowptr[nc * s + 1] - rowptr[nc * s]);
  std::vector<int32_t> cc(rpc[nn]);
  std::vector<std::vector<int32_t>> pos(nc, std::vector<int32_t>(rpc[nn]));
  parallel_chunks(nn, [&](int64_t s0, int64_t s1) {
    for (int64_t s = s0; s < s1; ++s) {
      const int32_t a = rowptr[nc * s], len = rowptr[nc * s + 1] - a, o = rpc[s];
      for (int32_t k = 0; k < len; ++k) {
        cc[o + k] = col[a + k] / nc;
        for (int c = 0; c < nc; ++c) pos[c][o + k] = rowptr[nc * s + c] + k;
      }
    }
  });
  const int rb2 = rb_for(nn, rpc.data(), nc == 3 ? tile_c<3>() : tile_c<2>());
  const bool dense2 = nn >= 64 && (int64_t)cc.size() == nn * (ncols / nc) &&
                      full_sorted_rows(nn, ncols / nc, rpc.data(), cc.data());
  // rb2 == 0: some row block of F does not fit the LDS tile of the gather
  // kernels (a few fat rows are enough: the first smoothed-aggregation level
  // of cube N = 73, 41 entries per row on average, fell back to the scalar
  // kernels - three times the bytes - for that).  The wave-per-row / dense
  // kernels need no tile, and the tile kernels choose their blocks greedily
  // by entries and distinct columns: tried below, kept if they took it
  A.dense2 = dense2;
  A.nnz2 = (int64_t)cc.size();
  CHK(A.rowptr2.ensure(nn + 1)); CHK(A.col2.ensure(A.nnz2)); CHK(A.val2.ensure(A.nnz2 + 2));
  CHK(A.kron_pos.ensure(nc * A.nnz2)); CHK(A.kron_flag.ensure(1));
  HIPCHK(hipMemcpy(A.rowptr2.p, rpc.data(), (nn + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(A.col2.p, cc.data(), A.nnz2 * sizeof(int), hipMemcpyHostToDevice));
  for (int c = 0; c < nc; ++c)
    HIPCHK(hipMemcpy(A.kron_pos.p + c * A.nnz2, pos[c].data(), A.nnz2 * sizeof(int),
                     hipMemcpyHostToDevice));
  A.kron = A.kron_pat = nc; A.rb2 = rb2;
  // bytes one fused step moves: F (12 B / entry) + five vector streams
  A.nt2 = g_nt_bytes >= 0 && 12.0 * (double)A.nnz2 + 40.0 * (double)nrows > (double)g_nt_bytes;
  CHK(build_vec_tile(h, A, nc, nn, ncols / nc, rpc, cc));
  // no row block fits the gather kernels' tile and the tile kernels declined
  // (a handful of rows per block): rows of a hundred entries and more go to
  // the wave-per-row kernel, F still read once for all components (the
  // restriction of a smoothed-aggregation level in space, ~180 entries per
  // row: cube N = 73 183 us with the scalar stream kernel)
  if (!rb2 && !A.vt && !dense2 && !A.long_rows && !A.wave_rows && nn > 0 &&
      (int64_t)cc.size() / nn >= 96)
    A.wave_rows = true;
  if (!rb2 && !A.vt && !g_want_wave && !dense2 && !A.wave_rows && !A.long_rows) {
    // nobody can run it as F (x) I: the scalar kernels take the operator
    A.kron = 0; A.kron_pat = 0; A.rb2 = 0; A.nnz2 = 0; A.dense2 = false;
    A.rowptr2.release(); A.col2.release(); A.val2.release(); A.kron_pos.release();
    return 0;
  }
  if (have_vals) CHK(refresh_kron(h, A));
  return 0;
}

// ROW-blocked structure: the `nc` rows of every node (nc = velocity components)
// carry the same columns - the discrete gradient A01 and nothing else on this
// path.  PCD_NO_ROWKRON=1: A/B switch.
int detect_rowkron(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                          const int32_t* rowptr, const int32_t* col, bool have_vals) {
  static const bool off = [] { const char* e = getenv("PCD_NO_ROWKRON"); return e && e[0] == '1'; }();
  A.rk = 0; A.rk_rb = 0; A.rk_nnz = 0;
  const int nc = h->vel_block == 3 ? 3 : 2;
  // (rectangular operators from a few thousand rows; long rows / dense / F (x) I
  // operators have kernels of their own)
  if (off || A.kron || A.kron_pat || A.dense || A.long_rows || A.wave_rows || nrows == ncols ||
      nrows < 3 * 1024 || nrows % nc || rowptr[nrows] % nc)
    return 0;
  const int64_t nn = nrows / nc;
  std::atomic<bool> ok{true};
  parallel_chunks(nn, [&](int64_t s0, int64_t s1) {
    for (int64_t s = s0; s < s1 && ok.load(std::memory_order_relaxed); ++s) {
      const int32_t a = rowptr[nc * s], len = rowptr[nc * s + 1] - a;
      bool good = true;
      for (int c = 1; c < nc && good; ++c)
        good = rowptr[nc * s + c + 1] - rowptr[nc * s + c] == len &&
               !memcmp(col + a, col + rowptr[nc * s + c], (size_t)len * sizeof(int32_t));
      if (!good) ok.store(false, std::memory_order_relaxed);
    }
  });
  if (!ok.load()) return 0;
  std::vector<int32_t> rpc(nn + 1, 0);
  for (int64_t s = 0; s < nn; ++s) rpc[s + 1] = rpc[s] + (rowptr[nc * s + 1] - rowptr[nc * s]);
  const int rb = rb_for(nn, rpc.data(), nc == 3 ? tile_c<3>() : tile_c<2>());
  if (!rb) return 0;
  std::vector<int32_t> cc(rpc[nn]), pos((size_t)rpc[nn] * nc);
  parallel_chunks(nn, [&](int64_t s0, int64_t s1) {
    for (int64_t s = s0; s < s1; ++s) {
      const int32_t len = rpc[s + 1] - rpc[s], o = rpc[s];
      for (int32_t k = 0; k < len; ++k) {
        cc[o + k] = col[rowptr[nc * s] + k];
        for (int c = 0; c < nc; ++c) pos[(size_t)(o + k) * nc + c] = rowptr[nc * s + c] + k;
      }
    }
  });
  A.rk_nnz = rpc[nn];
  CHK(A.rk_rowptr.ensure(nn + 1)); CHK(A.rk_col.ensure(A.rk_nnz)); CHK(A.rk_pos.ensure(pos.size()));
  CHK(A.rk_val.ensure(pos.size() + 2));
  HIPCHK(hipMemcpy(A.rk_rowptr.p, rpc.data(), (nn + 1) * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(A.rk_col.p, cc.data(), cc.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(A.rk_pos.p, pos.data(), pos.size() * sizeof(int), hipMemcpyHostToDevice));
  A.rk = nc; A.rk_rb = rb;
  // bytes one launch moves: (4 + 8 nc) per node-entry + the vectors
  A.rk_nt = g_nt_bytes >= 0 &&
            (4.0 + 8.0 * nc) * (double)A.rk_nnz + 16.0 * (double)nrows > (double)g_nt_bytes;
  if (have_vals)
    hipLaunchKernelGGL(k_lm_values, dim3(grid1d(A.rk_nnz * nc, 4)), dim3(kBlock), 0, h->stream,
                       A.rk_nnz * nc, A.rk_pos.p, A.val.p, A.rk_val.p);
  HIPCHK(hipGetLastError());
  return 0;
}

int upload_csr(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                      const int32_t* rowptr, const int32_t* col,
                      const double* val, const int64_t* src) {
  const int64_t nnz = rowptr[nrows];
  // a one-launch Chebyshev patch belongs to the PATTERN it was cut from: a
  // second hand-over on the slot drops it (inner_prepare builds the new one)
  A.cp.release();
  A.nrows = nrows; A.ncols = ncols; A.nnz = nnz;
  CHK(A.rowptr.ensure(nrows + 1)); CHK(A.col.ensure(nnz)); CHK(A.val.ensure(nnz));
  HIPCHK(hipMemcpyAsync(A.rowptr.p, rowptr, (nrows + 1) * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (nnz) HIPCHK(hipMemcpyAsync(A.col.p, col, nnz * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (val && nnz) HIPCHK(hipMemcpyAsync(A.val.p, val, nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
  A.has_src = false;
  if (src) {
    CHK(A.src.ensure(nnz));
    if (nnz) HIPCHK(hipMemcpyAsync(A.src.p, src, nnz * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    A.has_src = true;
  }
  HIPCHK(hipStreamSynchronize(h->stream));   // host arrays may be freed by the caller
  A.set = true;
  A.lpr = choose_lpr(A);
  A.rb = g_force_vector ? 0 : choose_rb(nrows, rowptr);
  A.small_tile = false;
  if (A.rb && !g_no_small_tile) {
    bool fits = true;
    for (int64_t r = 0; r < nrows && fits; r += A.rb)
      if (rowptr[std::min<int64_t>(r + A.rb, nrows)] - rowptr[r] > kTileSmall) fits = false;
    A.small_tile = fits && nrows >= (int64_t)A.rb * 1280;   // (only when workgroups queue for CUs)
  }
  A.long_rows = A.rb == 0 && nrows > 0 && nnz / nrows >= 256;
  // the dense kernels read `val` as a row-major matrix and ignore `col`: only
  // valid when every row stores columns 0..ncols-1 in ascending order
  A.dense = nrows >= 64 && nnz == nrows * ncols && full_sorted_rows(nrows, ncols, rowptr, col);
  // measured (profiles/r02_f_timeline.txt): a wave per row wins on the few,
  // very long rows of a residual-restriction product (150-300 entries: 7 us
  // against 13-38 us for the stream kernel's serialised tile passes) and
  // loses on the many 30-80-entry rows of an up-sweep product (29 against
  // 13 us at 10^5 rows): half of its lanes idle and every row pays its own
  // dependent chain
  // ... and on many rows (the up-sweep product of a 26 000-node level, 150
  // entries per row: 15 against 12 us), so: long rows AND few of them
  // ... and rows of thousands of entries (3-D: a coarse hat function covers
  // thousands of fine nodes) want a whole workgroup each
  if (g_want_wave && nrows > 0 && nnz / nrows >= 1024) A.long_rows = true;
  A.wave_rows = g_want_wave && !A.long_rows && nrows > 0 && nnz / nrows >= 96 &&
                (nrows <= 3 * 8192 || nnz / nrows >= 300);
  CHK(detect_kron(h, A, nrows, ncols, rowptr, col, val != nullptr));
  CHK(detect_rowkron(h, A, nrows, ncols, rowptr, col, val != nullptr));
  return 0;
}

// field spaces are fixed by the first operator that shows their size
int ensure_space(Engine* h, Space& sp, int64_t n, bool velocity, const char* what) {
  if (!h->comm) return 0;
  if (sp.nf == 0) { sp = Space::field(n, h->nranks, velocity ? h->vel_block : 1); return 0; }
  if (sp.total() != n)
    return fail(PCD_ERR_ARG, "%s: size %lld does not match the partitioned space (%lld)",
                what, (long long)n, (long long)sp.total());
  return 0;
}

template <class Span>
static int upload_owned(Engine* h, DCsr& A, const Space* rs, const Space* cs,
                        int64_t nrow_loc, Span span, const int32_t* col,
                        const double* val, const int64_t* src);

// Hand over a GLOBAL CSR.  One GPU: uploaded as is.  Several ranks: this
// rank's row block with localised columns and the halo plan (pcd_dist.hpp);
// the provenance array then maps local entries to the caller's value array.
int upload_global(Engine* h, DCsr& A, const Space* rs, const Space* cs,
                         int64_t nrows, int64_t ncols, const int32_t* rowptr,
                         const int32_t* col, const double* val,
                         const int64_t* src) {
  A.gnnz = rowptr[nrows];
  A.val_src = false;
  if (!h->comm) {
    A.plan = HaloPlan();
    return upload_csr(h, A, nrows, ncols, rowptr, col, val, src);
  }
  // this rank's rows only (a view of the global arrays); what the others need
  // from it arrives in the set-up handshake (pcd_dist.hpp: localize_owned)
  const int me = h->rank;
  auto span = [&](int64_t i) {
    const int64_t g = rs->global(i, me);
    return std::pair<int64_t, int64_t>(rowptr[g], rowptr[g + 1]);
  };
  return upload_owned(h, A, rs, cs, rs->nloc(me), span, col, val, src);
}

// hand-over of this rank's rows (`span(i)`: entries of local row i, GLOBAL
// column ids): localisation, halo plan by handshake, upload
template <class Span>
static int upload_owned(Engine* h, DCsr& A, const Space* rs, const Space* cs,
                        int64_t nrow_loc, Span span, const int32_t* col,
                        const double* val, const int64_t* src) {
  std::vector<int32_t> orp, oc;
  std::vector<double> ov;
  std::vector<int64_t> osrc;
  HaloPlan plan;
  std::string err;
  HIPCHK(hipSetDevice(h->device));
  if (localize_owned(*rs, *cs, h->rank, h->nranks, nrow_loc, span, col, val, src,
                     h->comm, h->stream, orp, oc, ov, osrc, plan, err))
    return fail(PCD_ERR_COMM, "set-up handshake: %s", err.c_str());
  CHK(upload_csr(h, A, nrow_loc, cs->nloc(h->rank), orp.data(), oc.data(),
                 val ? ov.data() : nullptr, osrc.data()));
  A.val_src = true;
  A.plan = plan;
  CHK(A.ghost.ensure(plan.nghost));
  CHK(A.sendbuf.ensure(plan.send_idx.size()));
  CHK(A.send_idx.ensure(plan.send_idx.size()));
  if (!plan.send_idx.empty())
    HIPCHK(hipMemcpy(A.send_idx.p, plan.send_idx.data(), plan.send_idx.size() * sizeof(int),
                     hipMemcpyHostToDevice));
  if (A.ph.dev.seq) { (void)hipFree(A.ph.dev.seq); }
  if (A.ph.owner) {
    std::lock_guard<std::mutex> lk(peer_live_mu());
    if (peer_live().count(A.ph.owner)) A.ph.owner->give_back(A.ph);
  }
  A.ph = PeerHalo();
  if (h->comm->peer()) {
    // collective: landing buffers and flags of this halo in the peers' arenas
    PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
    if (pb->register_halo(A.plan, A.send_idx.p, A.ghost.p, A.ph, h->stream))
      return fail(PCD_ERR_COMM, "peer halo registration: %s", pb->err.c_str());
  }
  return 0;
}

// Hand over a GLOBAL CSR in the caller's numbering with the engine renumbering
// of its row / column space applied first (either may be the identity)
int upload_global_r(Engine* h, DCsr& A, const Space* rs, const Space* cs,
                           int64_t nrows, int64_t ncols, const int32_t* rowptr,
                           const int32_t* col, const double* val, const int64_t* src,
                           const Reorder* rr, const Reorder* rc) {
  const bool pr = rr && rr->active(), pc = rc && rc->active();
  if (!pr && !pc) return upload_global(h, A, rs, cs, nrows, ncols, rowptr, col, val, src);
  if ((pr && (int64_t)rr->n2o.size() != nrows) || (pc && (int64_t)rc->o2n.size() != ncols))
    return fail(PCD_ERR_ARG, "operator %lld x %lld does not match the renumbered space",
                (long long)nrows, (long long)ncols);
  PermCsr B;
  permute_csr(nrows, rowptr, col, pr ? rr->rows() : nullptr, pc ? rc->cols() : nullptr, B);
  std::vector<double> bv;
  if (val) { bv.resize(B.src.size()); for (size_t k = 0; k < bv.size(); ++k) bv[k] = val[B.src[k]]; }
  if (src) for (auto& q : B.src) q = src[q];
  const int64_t gnnz = rowptr[nrows];
  CHK(upload_global(h, A, rs, cs, nrows, ncols, B.rp.data(), B.ci.data(),
                    val ? bv.data() : nullptr, B.src.data()));
  A.gnnz = gnnz;
  A.val_src = true;                      // refreshes arrive in the caller's entry order
  return 0;
}

int upload_perm(Engine* h, Reorder& r) {
  if (!r.active()) return 0;
  CHK(r.d_n2o.ensure(r.n2o.size()));
  HIPCHK(hipMemcpy(r.d_n2o.p, r.n2o.data(), r.n2o.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

// new values of a handed-over operator: one GPU copies, several ranks stage
// the caller's global array and gather their entries
int refresh_values(Engine* h, DCsr& A, const double* vals, int mem) {
  if (!A.val_src) {
    HIPCHK(hipMemcpyAsync(A.val.p, vals, A.nnz * sizeof(double),
                          mem == PCD_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          h->stream));
    return 0;
  }
  const double* dv = vals;
  if (mem == PCD_MEM_HOST) {
    // (its own buffer: `sysvals` keeps the staged system values, which the
    // device producer refreshes in place)
    CHK(h->valstage.ensure(A.gnnz));
    HIPCHK(hipMemcpyAsync(h->valstage.p, vals, A.gnnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    dv = h->valstage.p;
  }
  if (A.nnz)
    hipLaunchKernelGGL(k_gather_vals, dim3(grid1d(A.nnz, 4)), dim3(kBlock), 0,
                       h->stream, A.nnz, A.src.p, dv, A.val.p);
  HIPCHK(hipGetLastError());
  return 0;
}

// copy between a GLOBAL device vector of a space and this rank's local one
int slice_in(Engine* h, const Space& sp, const double* full, double* loc) {
  int64_t off = 0;
  for (int f = 0; f < sp.nf; ++f) {
    const int64_t b0 = sp.bounds[f][h->rank], len = sp.bounds[f][h->rank + 1] - b0;
    if (len) HIPCHK(hipMemcpyAsync(loc + off, full + sp.goff[f] + b0, len * sizeof(double),
                                   hipMemcpyDeviceToDevice, h->stream));
    off += len;
  }
  return 0;
}
int slice_out(Engine* h, const Space& sp, const double* loc, double* full) {
  const int64_t n = sp.total();
  HIPCHK(hipMemsetAsync(full, 0, n * sizeof(double), h->stream));
  int64_t off = 0;
  for (int f = 0; f < sp.nf; ++f) {
    const int64_t b0 = sp.bounds[f][h->rank], len = sp.bounds[f][h->rank + 1] - b0;
    if (len) HIPCHK(hipMemcpyAsync(full + sp.goff[f] + b0, loc + off, len * sizeof(double),
                                   hipMemcpyDeviceToDevice, h->stream));
    off += len;
  }
  if (h->comm->allreduce(full, n, h->stream))
    return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
  return 0;
}

// new values of operator `which`: everything composed from the old ones (a
// fused finest multigrid level, explicit factors) is stale
void values_changed(Engine* h, int which) {
  for (int slot = 0; slot < PCD_KSP_COUNT; ++slot) {
    if (kSlotMat[slot] != which) continue;
    Inner& s = h->inner[slot];
    if (!s.mg.empty() && s.mg.back().fused) { s.mg.back().fused = false; ++h->gen; }
    if (!s.chain.empty()) s.chain_stale = true;
  }
}


// ---- C ABI (linkage from the declarations of include/pcd_engine.h) ----

// several ranks, peer protocol: did a wait for a neighbour give up since the
// last check?  (the kernels never hang: they set an error word and go on)
int peer_check(Engine* h) {
  if (!h->comm || !h->comm->peer()) return 0;
  PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
  if (pb->take_error(h->stream)) return fail(PCD_ERR_COMM, "%s", pb->err.c_str());
  return 0;
}

int pcd_synchronize(pcd_handle h) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  HIPCHK(hipStreamSynchronize(h->stream));
  return peer_check(h);
} PCD_ABI_CATCH(pcd_synchronize)

int pcd_set_csr(pcd_handle h, int which, int64_t nrows, int64_t ncols,
                const int32_t* rowptr, const int32_t* colidx,
                const double* vals) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (which < 0 || which >= PCD_MAT_A)
    return fail(PCD_ERR_ARG, "set_csr: operator %d cannot be set directly", which);
  if (!rowptr || (!colidx && rowptr[nrows]) || nrows < 0 || ncols < 0)
    return fail(PCD_ERR_ARG, "set_csr: bad arrays");
  if (nrows >= INT32_MAX || ncols >= INT32_MAX)
    return fail(PCD_ERR_ARG, "set_csr: dimensions exceed int32 indexing");
  HIPCHK(hipSetDevice(h->device));
  DCsr& A = h->mat[which];
  const Space *rs = nullptr, *cs = nullptr;
  if (h->comm) {
    if (which == PCD_MAT_A00) {
      CHK(ensure_space(h, h->sp_u, nrows, true, "set_csr"));
      rs = cs = &h->sp_u;
    } else if (which == PCD_MAT_A01) {
      CHK(ensure_space(h, h->sp_u, nrows, true, "set_csr"));
      CHK(ensure_space(h, h->sp_p, ncols, false, "set_csr"));
      rs = &h->sp_u; cs = &h->sp_p;
    } else {
      CHK(ensure_space(h, h->sp_p, nrows, false, "set_csr"));
      rs = cs = &h->sp_p;
    }
  }
  // operators handed over one by one arrive in the caller's FIELD numbering:
  // the engine renumbering decided at pcd_set_system applies to them as well
  const Reorder* rr = (which == PCD_MAT_A00 || which == PCD_MAT_A01) ? &h->ru : &h->rp;
  const Reorder* rc = (which == PCD_MAT_A00) ? &h->ru : &h->rp;
  CHK(upload_global_r(h, A, rs, cs, nrows, ncols, rowptr, colidx, vals, nullptr, rr, rc));
  CHK(refresh_dinv(h, A));
  values_changed(h, which);
  h->ready = false; ++h->gen;
  return 0;
} PCD_ABI_CATCH(pcd_set_csr)

int pcd_row_range(pcd_handle h, int velocity, int64_t n_global, int64_t* r0, int64_t* r1) try {
  if (!h || !r0 || !r1 || n_global < 0) return fail(PCD_ERR_ARG, "row_range: bad arguments");
  if (!h->comm) { *r0 = 0; *r1 = n_global; return 0; }
  // the same rule for every space of the engine - the fields and the levels of
  // their multigrid hierarchies: even cuts, on node boundaries for velocities
  const std::vector<int64_t> b = Space::cut(n_global, h->nranks, velocity ? h->vel_block : 1);
  *r0 = b[h->rank]; *r1 = b[h->rank + 1];
  return 0;
} PCD_ABI_CATCH(pcd_row_range)

int pcd_set_csr_local(pcd_handle h, int which, int64_t nrows_global, int64_t ncols_global,
                      int64_t nrows_local, const int32_t* rowptr, const int32_t* colidx,
                      const double* vals) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (which < 0 || which >= PCD_MAT_A)
    return fail(PCD_ERR_ARG, "set_csr_local: operator %d cannot be set directly", which);
  if (!rowptr || nrows_local < 0 || (!colidx && rowptr[nrows_local]) || nrows_global < 0 || ncols_global < 0)
    return fail(PCD_ERR_ARG, "set_csr_local: bad arrays");
  if (nrows_global >= INT32_MAX || ncols_global >= INT32_MAX)
    return fail(PCD_ERR_ARG, "set_csr_local: dimensions exceed int32 indexing");
  if (h->ru.active() || h->rp.active())
    return fail(PCD_ERR_STATE, "set_csr_local: the engine renumbered the dofs at pcd_set_system "
                               "(PCD_REORDER); rank-local hand-over needs PCD_REORDER=none");
  if (!h->comm) {
    if (nrows_local != nrows_global)
      return fail(PCD_ERR_ARG, "set_csr_local: one rank owns every row (%lld), got %lld",
                  (long long)nrows_global, (long long)nrows_local);
    return pcd_set_csr(h, which, nrows_global, ncols_global, rowptr, colidx, vals);
  }
  HIPCHK(hipSetDevice(h->device));
  DCsr& A = h->mat[which];
  const Space *rs = nullptr, *cs = nullptr;
  if (which == PCD_MAT_A00) {
    CHK(ensure_space(h, h->sp_u, nrows_global, true, "set_csr_local"));
    rs = cs = &h->sp_u;
  } else if (which == PCD_MAT_A01) {
    CHK(ensure_space(h, h->sp_u, nrows_global, true, "set_csr_local"));
    CHK(ensure_space(h, h->sp_p, ncols_global, false, "set_csr_local"));
    rs = &h->sp_u; cs = &h->sp_p;
  } else {
    CHK(ensure_space(h, h->sp_p, nrows_global, false, "set_csr_local"));
    rs = cs = &h->sp_p;
  }
  if (cs->total() != ncols_global) return fail(PCD_ERR_ARG, "set_csr_local: column count does not match the partitioned space");
  if (rs->nloc(h->rank) != nrows_local)
    return fail(PCD_ERR_ARG, "set_csr_local: this rank owns %lld rows (pcd_row_range), got %lld",
                (long long)rs->nloc(h->rank), (long long)nrows_local);
  for (int64_t k = 0; k < rowptr[nrows_local]; ++k)
    if (colidx[k] < 0 || colidx[k] >= ncols_global)
      return fail(PCD_ERR_ARG, "set_csr_local: column id %d outside [0, %lld)", colidx[k], (long long)ncols_global);
  auto span = [&](int64_t i) { return std::pair<int64_t, int64_t>(rowptr[i], rowptr[i + 1]); };
  A.gnnz = rowptr[nrows_local];            // value updates carry this rank's entries
  CHK(upload_owned(h, A, rs, cs, nrows_local, span, colidx, vals, nullptr));
  CHK(refresh_dinv(h, A));
  values_changed(h, which);
  h->ready = false; ++h->gen;
  return 0;
} PCD_ABI_CATCH(pcd_set_csr_local)

int pcd_update_values(pcd_handle h, int which, const double* vals, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (which < 0 || which >= PCD_MAT_A || !h->mat[which].set)
    return fail(PCD_ERR_STATE, "update_values: operator %d not set", which);
  if (!vals) return fail(PCD_ERR_ARG, "update_values: null vals");
  if (mem != PCD_MEM_HOST && mem != PCD_MEM_DEVICE) return fail(PCD_ERR_ARG, "update_values: bad mem flag %d", mem);
  DCsr& A = h->mat[which];
  CHK(refresh_values(h, A, vals, mem));
  CHK(refresh_dinv(h, A));
  values_changed(h, which);
  if (mem == PCD_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_update_values)

// host-side MatCreateSubMatrix with value provenance
void extract_block(int64_t nr, const int32_t* rows, const int32_t* rowptr,
                          const int32_t* col, const std::vector<int32_t>& colmap,
                          std::vector<int32_t>& orp, std::vector<int32_t>& oc,
                          std::vector<int64_t>& osrc) {
  orp.assign(nr + 1, 0);
  parallel_chunks(nr, [&](int64_t i0, int64_t i1) {
    for (int64_t i = i0; i < i1; ++i) {
      int32_t c = 0;
      for (int32_t k = rowptr[rows[i]]; k < rowptr[rows[i] + 1]; ++k) c += colmap[col[k]] >= 0;
      orp[i + 1] = c;
    }
  });
  for (int64_t i = 0; i < nr; ++i) orp[i + 1] += orp[i];
  oc.resize(orp[nr]); osrc.resize(orp[nr]);
  parallel_chunks(nr, [&](int64_t i0, int64_t i1) {
    std::vector<std::pair<int32_t, int64_t>> tmp;
    for (int64_t i = i0; i < i1; ++i) {
      tmp.clear();
      bool sorted = true;
      for (int32_t k = rowptr[rows[i]]; k < rowptr[rows[i] + 1]; ++k) {
        const int32_t c = colmap[col[k]];
        if (c < 0) continue;
        if (!tmp.empty() && c < tmp.back().first) sorted = false;
        tmp.emplace_back(c, (int64_t)k);
      }
      if (!sorted) std::sort(tmp.begin(), tmp.end());
      int64_t q = orp[i];
      for (auto& t : tmp) { oc[q] = t.first; osrc[q] = t.second; ++q; }
    }
  });
}

int gather_block_values(Engine* h, DCsr& A, const double* dvals) {
  if (!A.nnz) return 0;
  hipLaunchKernelGGL(k_gather_vals, dim3(grid1d(A.nnz, 4)), dim3(kBlock), 0,
                     h->stream, A.nnz, A.src.p, dvals, A.val.p);
  if (A.rk && A.rk_nnz)
    hipLaunchKernelGGL(k_lm_values, dim3(grid1d(A.rk_nnz * A.rk, 4)), dim3(kBlock), 0, h->stream,
                       A.rk_nnz * A.rk, A.rk_pos.p, A.val.p, A.rk_val.p);
  HIPCHK(hipGetLastError());
  return 0;
}

// Engine renumbering (pcd_reorder.hpp), decided once per pattern at
// pcd_set_system: the velocity nodes by reverse Cuthill-McKee on the node graph
// of A00 when the caller's numbering is not local ("auto": mean |row - col| / n
// above 0.1; a geometric numbering gives a few per cent, a random one 0.33),
// the pressure dofs by the first velocity node they couple to.  Outputs the
// renumbered index sets (empty: unchanged).  Operators handed over BEFORE this
// call stay in the numbering they came in: then only the velocity is renumbered.
int decide_reordering(Engine* h, int64_t n, const int32_t* rowptr, const int32_t* colidx,
                             int64_t n_u, const int32_t* is_u, int64_t n_p, const int32_t* is_p,
                             std::vector<int32_t>& isu_r, std::vector<int32_t>& isp_r) {
  { const char* e = getenv("PCD_REORDER");
    if (e) h->reorder_mode = !strcmp(e, "none") ? 0 : !strcmp(e, "always") ? 2 : !strcmp(e, "cluster") ? 3 : 1; }
  for (int64_t i = 0; i < n_u; ++i)
    if (is_u[i] < 0 || is_u[i] >= n) return fail(PCD_ERR_ARG, "set_system: index sets do not partition 0..n-1");
  for (int64_t i = 0; i < n_p; ++i)
    if (is_p[i] < 0 || is_p[i] >= n) return fail(PCD_ERR_ARG, "set_system: index sets do not partition 0..n-1");
  if ((int64_t)h->ru.n2o.size() == n_u && (!h->rp.active() || (int64_t)h->rp.n2o.size() == n_p)) {
    // same spaces as before (new pattern of the same problem): keep the numbering
  } else {
    h->ru.clear(); h->rp.clear(); h->rs.clear();
    bool u_ops = h->mat[PCD_MAT_A00].set || h->mat[PCD_MAT_A01].set;
    // a velocity hierarchy pushed before the system (the finest level carries
    // no operator, so A00 need not be set for it) is in the caller's numbering
    if (!h->inner[PCD_KSP_A00].mg.empty()) u_ops = true;
    if (h->reorder_mode && !u_ops && n_u > 0) {
      std::vector<int32_t> mu(n, -1), rp, cc; std::vector<int64_t> src;
      for (int64_t i = 0; i < n_u; ++i) mu[is_u[i]] = (int32_t)i;
      extract_block(n_u, is_u, rowptr, colidx, mu, rp, cc, src);
      // node graph when the block has the interleaved-component pattern
      int nc = 1;
      for (int cand : {h->vel_block, 2, 3})
        if (cand > 1 && kron_pattern(cand, n_u, n_u, rp.data(), cc.data())) { nc = cand; break; }
      std::vector<int32_t> grp, gcc;
      const int32_t *gp = rp.data(), *gc = cc.data();
      const int64_t nn = n_u / nc;
      if (nc > 1) {
        grp.assign(nn + 1, 0);
        for (int64_t s2 = 0; s2 < nn; ++s2) grp[s2 + 1] = grp[s2] + (rp[nc * s2 + 1] - rp[nc * s2]);
        gcc.resize(grp[nn]);
        for (int64_t s2 = 0; s2 < nn; ++s2)
          for (int32_t k = 0; k < grp[s2 + 1] - grp[s2]; ++k) gcc[grp[s2] + k] = cc[rp[nc * s2] + k] / nc;
        gp = grp.data(); gc = gcc.data();
      }
      const double m = locality_metric(nn, gp, gc);
      if (h->reorder_mode >= 2 || m > 0.1) {
        // ("cluster": graph balls of PCD_CLUSTER_NODES nodes in the caller's
        // order instead of reverse Cuthill-McKee - for the vector-tile kernels)
        static const int kc = [] { const char* e = getenv("PCD_CLUSTER_NODES"); return e && atoi(e) > 0 ? atoi(e) : 56; }();
        std::vector<int32_t> nodes = h->reorder_mode == 3 && m <= 0.1 ? cluster_order(nn, gp, gc, kc)
                                                                      : rcm_order(nn, gp, gc);
        h->ru.n2o = nc > 1 ? expand_nodes(nodes, nc) : nodes;
        h->ru.o2n = invert_perm(h->ru.n2o);
        // pressure: by the first (renumbered) velocity dof it couples to - only
        // if no pressure operator was handed over in the caller's numbering yet
        bool p_ops = h->mat[PCD_MAT_AP].set || h->mat[PCD_MAT_MP].set || h->mat[PCD_MAT_KP].set ||
                     h->mat[PCD_MAT_RP].set;
        for (int sl : {PCD_KSP_AP, PCD_KSP_MP, PCD_KSP_RP}) if (!h->inner[sl].mg.empty()) p_ops = true;
        if (!p_ops && n_p > 0) {
          std::vector<int32_t> mp(n, -1);
          for (int64_t i = 0; i < n_p; ++i) mp[is_p[i]] = (int32_t)i;
          extract_block(n_u, is_u, rowptr, colidx, mp, rp, cc, src);     // A01 pattern
          h->rp.n2o = induced_order(n_u, n_p, rp.data(), cc.data(), h->ru.o2n.data());
          h->rp.o2n = invert_perm(h->rp.n2o);
        }
        h->rs.n2o.resize(n);
        for (int64_t i = 0; i < n_u; ++i) h->rs.n2o[i] = h->ru.n2o[i];
        for (int64_t j = 0; j < n_p; ++j)
          h->rs.n2o[n_u + j] = (int32_t)(n_u + (h->rp.active() ? h->rp.n2o[j] : j));
        h->rs.o2n = invert_perm(h->rs.n2o);
        CHK(upload_perm(h, h->ru)); CHK(upload_perm(h, h->rp)); CHK(upload_perm(h, h->rs));
      }
    }
  }
  if (h->ru.active()) { isu_r.resize(n_u); for (int64_t i = 0; i < n_u; ++i) isu_r[i] = is_u[h->ru.n2o[i]]; }
  if (h->rp.active()) { isp_r.resize(n_p); for (int64_t i = 0; i < n_p; ++i) isp_r[i] = is_p[h->rp.n2o[i]]; }
  return 0;
}

int pcd_set_reorder(pcd_handle h, int mode) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (mode < 0 || mode > 3) return fail(PCD_ERR_ARG, "set_reorder: mode 0 (never), 1 (auto), 2 (always) or 3 (cluster)");
  h->reorder_mode = mode;
  return 0;
} PCD_ABI_CATCH(pcd_set_reorder)

int pcd_update_system(pcd_handle h, const double* vals, const double* pvals,
                      int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!h->mat[PCD_MAT_A].set) return fail(PCD_ERR_STATE, "update_system: no system set");
  if (!vals) return fail(PCD_ERR_ARG, "update_system: null vals");
  const double *dv = vals, *dp = pvals;
  if (mem == PCD_MEM_HOST) {
    CHK(h->sysvals.ensure(h->sys_nnz));
    HIPCHK(hipMemcpyAsync(h->sysvals.p, vals, h->sys_nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
    dv = h->sysvals.p;
    if (pvals) {
      CHK(h->psysvals.ensure(h->sys_nnz));
      HIPCHK(hipMemcpyAsync(h->psysvals.p, pvals, h->sys_nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
      dp = h->psysvals.p;
    }
  }
  if (!dp) dp = dv;                      // P = A  (nonlinear_solvers.py:75)
  h->p_is_a = (dp == dv);
  if (h->a10.set) {
    CHK(gather_block_values(h, h->a10, dv));
    CHK(gather_block_values(h, h->a11, dv));
    if (mem == PCD_MEM_HOST) {            // (device-side updates leave A11 alone)
      h->a11_zero = true;
      for (int64_t k : h->a11_src_host)
        if (vals[k] != 0.0) { h->a11_zero = false; break; }
      if (h->sys_local && h->comm) {
        // every rank saw its own rows only; the ranks must agree on whether
        // the (1,1) block takes part in the grouped halo exchange
        double f = h->a11_zero ? 0.0 : 1.0;
        CHK(h->flagbuf.ensure(1));
        HIPCHK(hipMemcpyAsync(h->flagbuf.p, &f, sizeof f, hipMemcpyHostToDevice, h->stream));
        if (h->comm->allreduce(h->flagbuf.p, 1, h->stream))
          return fail(PCD_ERR_COMM, "allreduce: %s", h->comm->err.c_str());
        HIPCHK(hipMemcpyAsync(&f, h->flagbuf.p, sizeof f, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->a11_zero = f == 0.0;
      }
    }
  }
  CHK(gather_block_values(h, h->mat[PCD_MAT_A], dv));
  CHK(gather_block_values(h, h->mat[PCD_MAT_A00], dp));
  CHK(gather_block_values(h, h->mat[PCD_MAT_A01], dp));
  CHK(refresh_dinv(h, h->mat[PCD_MAT_A00]));
  values_changed(h, PCD_MAT_A00);
  if (mem == PCD_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
} PCD_ABI_CATCH(pcd_update_system)

int pcd_set_system(pcd_handle h, int64_t n, const int32_t* rowptr,
                   const int32_t* colidx, const double* vals,
                   const double* pvals, int64_t n_u, const int32_t* is_u,
                   int64_t n_p, const int32_t* is_p) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!rowptr || !colidx || !vals || !is_u || !is_p)
    return fail(PCD_ERR_ARG, "set_system: null argument");
  if (n_u + n_p != n) return fail(PCD_ERR_ARG, "set_system: n_u + n_p != n");
  if (n >= INT32_MAX) return fail(PCD_ERR_ARG, "set_system: n exceeds int32 indexing");
  HIPCHK(hipSetDevice(h->device));
  PhaseTimer pt;
  std::vector<int32_t> isu_r, isp_r;       // index sets in engine numbering
  CHK(decide_reordering(h, n, rowptr, colidx, n_u, is_u, n_p, is_p, isu_r, isp_r));
  pt.lap("set_system: reordering");
  if (!isu_r.empty()) is_u = isu_r.data();
  if (!isp_r.empty()) is_p = isp_r.data();
  std::vector<int32_t> perm(n), mu(n, -1), mp(n, -1), ma(n, -1);
  for (int64_t i = 0; i < n_u; ++i) perm[i] = is_u[i];
  for (int64_t i = 0; i < n_p; ++i) perm[n_u + i] = is_p[i];
  for (int64_t i = 0; i < n; ++i) {
    if (perm[i] < 0 || perm[i] >= n || ma[perm[i]] >= 0)
      return fail(PCD_ERR_ARG, "set_system: index sets do not partition 0..n-1");
    ma[perm[i]] = (int32_t)i;
  }
  for (int64_t i = 0; i < n_u; ++i) mu[is_u[i]] = (int32_t)i;
  for (int64_t i = 0; i < n_p; ++i) mp[is_p[i]] = (int32_t)i;
  h->n_u = n_u; h->n_p = n_p; h->sys_nnz = rowptr[n];
  h->sys_local = false;
  h->perm_glob = perm;
  CHK(ensure_space(h, h->sp_u, n_u, true, "set_system"));
  CHK(ensure_space(h, h->sp_p, n_p, false, "set_system"));
  if (h->comm) h->sp_sys = Space::system(h->sp_u, h->sp_p);
  {  // local split position -> caller's index
    const int64_t nloc = h->comm ? h->sp_sys.nloc(h->rank) : n;
    std::vector<int32_t> pl(nloc);
    for (int64_t i = 0; i < nloc; ++i)
      pl[i] = perm[h->comm ? h->sp_sys.global(i, h->rank) : i];
    CHK(h->perm.ensure(nloc));
    if (nloc) HIPCHK(hipMemcpy(h->perm.p, pl.data(), nloc * sizeof(int), hipMemcpyHostToDevice));
  }
  pt.lap("set_system: index maps");
  std::vector<int32_t> rp, cc; std::vector<int64_t> src;
  extract_block(n_u, is_u, rowptr, colidx, mu, rp, cc, src);
  pt.lap("set_system: extract A00");
  CHK(upload_global(h, h->mat[PCD_MAT_A00], &h->sp_u, &h->sp_u, n_u, n_u, rp.data(), cc.data(), nullptr, src.data()));
  pt.lap("set_system: upload A00");
  extract_block(n_u, is_u, rowptr, colidx, mp, rp, cc, src);
  CHK(upload_global(h, h->mat[PCD_MAT_A01], &h->sp_u, &h->sp_p, n_u, n_p, rp.data(), cc.data(), nullptr, src.data()));
  pt.lap("set_system: A01");
  h->a10.release(); h->a11.release(); h->a11_src_host.clear();
  {
    // (1,0) and (1,1) blocks: w = A z is applied block-wise (velocity block
    // through its F x I fast path) - with several ranks too, the halos of the
    // blocks travelling in one grouped exchange (apply_system)
    extract_block(n_p, is_p, rowptr, colidx, mu, rp, cc, src);
    CHK(upload_global(h, h->a10, &h->sp_p, &h->sp_u, n_p, n_u, rp.data(), cc.data(), nullptr, src.data()));
    extract_block(n_p, is_p, rowptr, colidx, mp, rp, cc, src);
    CHK(upload_global(h, h->a11, &h->sp_p, &h->sp_p, n_p, n_p, rp.data(), cc.data(), nullptr, src.data()));
    h->a11_src_host = src;               // (positions in the caller's values, all rows)
  }
  pt.lap("set_system: A10, A11");
  extract_block(n, perm.data(), rowptr, colidx, ma, rp, cc, src);
  pt.lap("set_system: extract A");
  CHK(upload_global(h, h->mat[PCD_MAT_A], &h->sp_sys, &h->sp_sys, n, n, rp.data(), cc.data(), nullptr, src.data()));
  pt.lap("set_system: upload A");
  h->ready = false; ++h->gen;
  const int rc = pcd_update_system(h, vals, pvals, PCD_MEM_HOST);
  pt.lap("set_system: values");
  return rc;
} PCD_ABI_CATCH(pcd_set_system)

// Rank-local form of pcd_set_system: this rank's rows of the monolithic matrix
// only.  `rows[i]` is the caller's (global) index of local row i: the rank's
// velocity rows is_u[u0 .. u1) followed by its pressure rows is_p[p0 .. p1),
// the ranges being pcd_row_range's; columns carry the caller's global indices.
// The index sets are handed over whole (O(n) integers per rank - the matrix,
// O(nnz / R), is what matters); who needs which of this rank's entries is
// found in the set-up handshake (pcd_dist.hpp: localize_owned), so no rank
// ever looks at a row it does not own.  The caller's dof order is kept
// (renumbering needs the whole graph).  pcd_update_system then takes the
// values of these rows, in this order.
int pcd_set_system_local(pcd_handle h, int64_t n, int64_t n_u, const int32_t* is_u,
                         int64_t n_p, const int32_t* is_p, int64_t nrows_local,
                         const int32_t* rows, const int32_t* rowptr, const int32_t* colidx,
                         const double* vals, const double* pvals) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (nrows_local < 0 || nrows_local > n)
    return fail(PCD_ERR_ARG, "set_system_local: nrows_local outside [0, n]");
  if (!rows || !rowptr || !vals || !is_u || !is_p || (!colidx && rowptr[nrows_local]))
    return fail(PCD_ERR_ARG, "set_system_local: null argument");
  if (n_u + n_p != n) return fail(PCD_ERR_ARG, "set_system_local: n_u + n_p != n");
  if (n >= INT32_MAX) return fail(PCD_ERR_ARG, "set_system_local: n exceeds int32 indexing");
  HIPCHK(hipSetDevice(h->device));
  h->ru.clear(); h->rp.clear(); h->rs.clear();
  std::vector<int32_t> perm(n), mu(n, -1), mp(n, -1), ma(n, -1);
  for (int64_t i = 0; i < n_u; ++i) perm[i] = is_u[i];
  for (int64_t i = 0; i < n_p; ++i) perm[n_u + i] = is_p[i];
  for (int64_t i = 0; i < n; ++i) {
    if (perm[i] < 0 || perm[i] >= n || ma[perm[i]] >= 0)
      return fail(PCD_ERR_ARG, "set_system_local: index sets do not partition 0..n-1");
    ma[perm[i]] = (int32_t)i;
  }
  for (int64_t i = 0; i < n_u; ++i) mu[is_u[i]] = (int32_t)i;
  for (int64_t i = 0; i < n_p; ++i) mp[is_p[i]] = (int32_t)i;
  h->n_u = n_u; h->n_p = n_p; h->sys_nnz = rowptr[nrows_local];
  h->sys_local = true;
  h->perm_glob = perm;
  CHK(ensure_space(h, h->sp_u, n_u, true, "set_system_local"));
  CHK(ensure_space(h, h->sp_p, n_p, false, "set_system_local"));
  if (h->comm) h->sp_sys = Space::system(h->sp_u, h->sp_p);
  const int me = h->rank;
  const int64_t nul = h->comm ? h->sp_u.nloc(me) : n_u, npl = h->comm ? h->sp_p.nloc(me) : n_p;
  if (nrows_local != nul + npl)
    return fail(PCD_ERR_ARG, "set_system_local: this rank owns %lld + %lld rows (pcd_row_range), got %lld",
                (long long)nul, (long long)npl, (long long)nrows_local);
  std::vector<int32_t> pl(nrows_local);
  for (int64_t i = 0; i < nrows_local; ++i) {
    pl[i] = perm[h->comm ? h->sp_sys.global(i, me) : i];
    if (rows[i] != pl[i])
      return fail(PCD_ERR_ARG, "set_system_local: local row %lld is the caller's row %d, expected %d "
                               "(velocity rows of pcd_row_range first, then the pressure rows)",
                  (long long)i, rows[i], pl[i]);
  }
  for (int64_t k = 0; k < rowptr[nrows_local]; ++k)
    if (colidx[k] < 0 || colidx[k] >= n)
      return fail(PCD_ERR_ARG, "set_system_local: column id %d outside [0, %lld)", colidx[k], (long long)n);
  CHK(h->perm.ensure(nrows_local));
  if (nrows_local) HIPCHK(hipMemcpy(h->perm.p, pl.data(), nrows_local * sizeof(int), hipMemcpyHostToDevice));
  // local row ids of the velocity / pressure rows in the arrays handed over
  std::vector<int32_t> lu(nul), lp(npl), la(nrows_local);
  for (int64_t i = 0; i < nul; ++i) lu[i] = (int32_t)i;
  for (int64_t i = 0; i < npl; ++i) lp[i] = (int32_t)(nul + i);
  for (int64_t i = 0; i < nrows_local; ++i) la[i] = (int32_t)i;
  std::vector<int32_t> rp, cc; std::vector<int64_t> src;
  auto hand_over = [&](DCsr& A, const Space* rs, const Space* cs, int64_t nr, const std::vector<int32_t>& lrows,
                       const std::vector<int32_t>& colmap, int64_t ncols_glob) -> int {
    extract_block(nr, lrows.data(), rowptr, colidx, colmap, rp, cc, src);
    A.gnnz = rp[nr];
    if (!h->comm) {
      A.plan = HaloPlan();
      A.val_src = false;
      return upload_csr(h, A, nr, ncols_glob, rp.data(), cc.data(), nullptr, src.data());
    }
    auto span = [&](int64_t i) { return std::pair<int64_t, int64_t>(rp[i], rp[i + 1]); };
    return upload_owned(h, A, rs, cs, nr, span, cc.data(), nullptr, src.data());
  };
  CHK(hand_over(h->mat[PCD_MAT_A00], &h->sp_u, &h->sp_u, nul, lu, mu, n_u));
  CHK(hand_over(h->mat[PCD_MAT_A01], &h->sp_u, &h->sp_p, nul, lu, mp, n_p));
  h->a10.release(); h->a11.release(); h->a11_src_host.clear();
  CHK(hand_over(h->a10, &h->sp_p, &h->sp_u, npl, lp, mu, n_u));
  CHK(hand_over(h->a11, &h->sp_p, &h->sp_p, npl, lp, mp, n_p));
  h->a11_src_host = src;                 // (positions in this rank's values)
  CHK(hand_over(h->mat[PCD_MAT_A], &h->sp_sys, &h->sp_sys, nrows_local, la, ma, n));
  h->ready = false; ++h->gen;
  return pcd_update_system(h, vals, pvals, PCD_MEM_HOST);
} PCD_ABI_CATCH(pcd_set_system_local)

int pcd_set_bc(pcd_handle h, int64_t n_bc, const int32_t* idx, const double* vals) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (n_bc < 0 || (n_bc && (!idx || !vals))) return fail(PCD_ERR_ARG, "set_bc: bad arrays");
  h->bc_host.assign(idx, idx + n_bc);
  h->bc_val_host.assign(vals, vals + n_bc);
  h->ready = false; ++h->gen;             // filtered / uploaded by pcd_setup
  return 0;
} PCD_ABI_CATCH(pcd_set_bc)

// ---- multigrid hierarchy ------------------------------------------------
int pcd_mg_begin(pcd_handle h, int slot, int nlevels, int nu_pre, int nu_post) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "mg_begin: bad slot %d", slot);
  if (nlevels < 1 || nlevels > 32 || nu_pre < 0 || nu_post < 0)
    return fail(PCD_ERR_ARG, "mg_begin: bad level / smoothing counts");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  Inner& s = h->inner[slot];
  for (auto& l : s.mg) l.release();
  s.mg.clear();
  s.mg.resize(nlevels);
  s.mg_r.clear(); s.mg_r.resize(nlevels);
  s.mg_r_known.assign(nlevels, 0);
  s.mg_space.clear(); s.mg_space.resize(nlevels);     // (row cuts belong to a hierarchy)
  s.nu_pre = nu_pre; s.nu_post = nu_post;
  ++h->gen;
  return 0;
} PCD_ABI_CATCH(pcd_mg_begin)

int pcd_mg_set_level_cuts(pcd_handle h, int slot, int level, int64_t n, const int64_t* bounds) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "mg_set_level_cuts: bad slot %d", slot);
  Inner& s = h->inner[slot];
  const int L = (int)s.mg.size();
  if (level < 0 || level >= L - 1)
    return fail(PCD_ERR_ARG, "mg_set_level_cuts: level %d outside [0,%d) (the finest level has the field's cuts)",
                level, L - 1);
  if (!bounds || n < 0) return fail(PCD_ERR_ARG, "mg_set_level_cuts: bad arguments");
  if (!h->comm) return fail(PCD_ERR_STATE, "mg_set_level_cuts: no communicator");
  const int even = slot == PCD_KSP_A00 ? h->vel_block : 1;
  if (bounds[0] != 0 || bounds[h->nranks] != n)
    return fail(PCD_ERR_ARG, "mg_set_level_cuts: cuts must run from 0 to n");
  for (int r = 0; r < h->nranks; ++r)
    if (bounds[r + 1] < bounds[r] || (even > 1 && bounds[r + 1] % even))
      return fail(PCD_ERR_ARG, "mg_set_level_cuts: cuts must ascend%s", even > 1 ? " on node boundaries" : "");
  if (s.mg[level].A.set || s.mg[level].P.set || (level + 1 < L && s.mg[level + 1].P.set))
    return fail(PCD_ERR_STATE, "mg_set_level_cuts: level %d or its prolongation is set already", level);
  if ((int)s.mg_space.size() != L) s.mg_space.resize(L);
  Space sp; sp.nf = 1; sp.bounds[0].assign(bounds, bounds + h->nranks + 1);
  s.mg_space[level] = sp;
  return 0;
} PCD_ABI_CATCH(pcd_mg_set_level_cuts)

int pcd_mg_set_level(pcd_handle h, int slot, int level, int64_t n,
                     const int32_t* rowptr, const int32_t* colidx,
                     const double* vals, int64_t p_rows, int64_t p_cols,
                     const int32_t* prowptr, const int32_t* pcolidx,
                     const double* pvals, double emin, double emax) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "mg_set_level: bad slot %d", slot);
  Inner& s = h->inner[slot];
  const int L = (int)s.mg.size();
  if (level < 0 || level >= L) return fail(PCD_ERR_ARG, "mg_set_level: level %d outside [0,%d)", level, L);
  if (level > 0 && !(emax > emin && emin > 0.0))
    return fail(PCD_ERR_ARG, "mg_set_level: smoother needs 0 < emin < emax");
  if (level > 0 && (!prowptr || !pcolidx || !pvals))
    return fail(PCD_ERR_ARG, "mg_set_level: prolongation missing");
  if (!rowptr && level != L - 1)
    return fail(PCD_ERR_ARG, "mg_set_level: coarse levels need an operator");
  if (rowptr && (!colidx || !vals)) return fail(PCD_ERR_ARG, "mg_set_level: bad operator arrays");
  MgLevel& M = s.mg[level];
  // Engine renumbering of the levels (pcd_reorder.hpp): the finest level takes
  // the renumbering of its field; a coarser level inherits its order through
  // the prolongation of the level above it - when the levels arrive finest
  // first - and keeps the caller's numbering otherwise (the coarsest level,
  // an explicit inverse, always does).  The arrays handed over are permuted
  // here, once; everything below sees the engine numbering only.
  PermCsr PA, PP;
  std::vector<double> va_p, vp_p;
  const int64_t* a_src = nullptr;
  {
    const int64_t nl0 = rowptr ? n : p_rows;
    if ((int)s.mg_r.size() != L) { s.mg_r.clear(); s.mg_r.resize(L); s.mg_r_known.assign(L, 0); }
    if (!s.mg_r_known[level]) {
      if (level == L - 1) {
        const Reorder& fr = slot == PCD_KSP_A00 ? h->ru : h->rp;
        if (fr.active() && (int64_t)fr.n2o.size() == nl0) { s.mg_r[level].n2o = fr.n2o; s.mg_r[level].o2n = fr.o2n; }
      }
      s.mg_r_known[level] = 1;
    }
    Reorder& rl = s.mg_r[level];
    if (level > 0 && !s.mg_r_known[level - 1]) {
      if (level - 1 > 0 && rl.active() && prowptr && pcolidx) {
        // velocity levels: order the NODES, keep a node's components together
        const int nc = slot == PCD_KSP_A00 ? h->vel_block : 1;
        if (nc > 1 && p_rows % nc == 0 && p_cols % nc == 0 && kron_pattern(nc, p_rows, p_cols, prowptr, pcolidx)) {
          const int64_t nf = p_rows / nc, ncs = p_cols / nc;
          std::vector<int32_t> grp(nf + 1, 0), gcc, fo2n(nf);
          for (int64_t q = 0; q < nf; ++q) grp[q + 1] = grp[q] + (prowptr[nc * q + 1] - prowptr[nc * q]);
          gcc.resize(grp[nf]);
          for (int64_t q = 0; q < nf; ++q)
            for (int32_t k = 0; k < grp[q + 1] - grp[q]; ++k) gcc[grp[q] + k] = pcolidx[prowptr[nc * q] + k] / nc;
          for (int64_t q = 0; q < nf; ++q) fo2n[q] = rl.o2n[nc * q] / nc;
          s.mg_r[level - 1].n2o = expand_nodes(induced_order(nf, ncs, grp.data(), gcc.data(), fo2n.data()), nc);
        } else {
          s.mg_r[level - 1].n2o = induced_order(p_rows, p_cols, prowptr, pcolidx, rl.o2n.data());
        }
        s.mg_r[level - 1].o2n = invert_perm(s.mg_r[level - 1].n2o);
      }
      s.mg_r_known[level - 1] = 1;
    }
    const Reorder* rc0 = level > 0 ? &s.mg_r[level - 1] : nullptr;
    if (rowptr && rl.active()) {
      if ((int64_t)rl.n2o.size() != n) return fail(PCD_ERR_ARG, "mg_set_level: level %d size does not match its renumbering", level);
      permute_csr(n, rowptr, colidx, rl.rows(), rl.cols(), PA);
      va_p.resize(PA.src.size());
      for (size_t k = 0; k < va_p.size(); ++k) va_p[k] = vals[PA.src[k]];
      rowptr = PA.rp.data(); colidx = PA.ci.data(); vals = va_p.data(); a_src = PA.src.data();
    }
    if (level > 0 && (rl.active() || (rc0 && rc0->active()))) {
      permute_csr(p_rows, prowptr, pcolidx, rl.rows(), rc0 ? rc0->cols() : nullptr, PP);
      vp_p.resize(PP.src.size());
      for (size_t k = 0; k < vp_p.size(); ++k) vp_p[k] = pvals[PP.src[k]];
      prowptr = PP.rp.data(); pcolidx = PP.ci.data(); pvals = vp_p.data();
    }
  }
  // multi-GPU: large levels are cut into contiguous row blocks like the finest
  // one (velocity levels keep the two components of a node together); levels
  // of at most PCD_REPLICATE_BELOW rows (default 60000) are replicated
  const int even = slot == PCD_KSP_A00 ? h->vel_block : 1;   // cut granule
  const Space *sl = nullptr, *sc = nullptr;
  const int64_t nl = rowptr ? n : p_rows;
  bool rep_l = false, rep_c = false;
  if (h->comm) {
    const char* lim_env = getenv("PCD_REPLICATE_BELOW");
    const int64_t limit = lim_env ? atoll(lim_env) : 60000LL;
    rep_l = level < L - 1 && nl <= limit;
    rep_c = level > 0 && p_cols <= limit;
    s.mg_space.resize(L);
    if (!rep_l) {
      if (s.mg_space[level].nf == 0) s.mg_space[level] = Space::field(nl, h->nranks, even);
      if (s.mg_space[level].total() != nl) return fail(PCD_ERR_ARG, "mg_set_level: level %d size mismatch", level);
      sl = &s.mg_space[level];
    }
    if (level > 0 && !rep_c) {
      if (s.mg_space[level - 1].nf == 0) s.mg_space[level - 1] = Space::field(p_cols, h->nranks, even);
      if (s.mg_space[level - 1].total() != p_cols) return fail(PCD_ERR_ARG, "mg_set_level: level %d prolongation width mismatch", level);
      sc = &s.mg_space[level - 1];
    }
  }
  M.replicated = rep_l; M.transition = h->comm && !rep_l && rep_c; M.n_coarse = p_cols;
  if (rowptr) {
    if (rep_l) { CHK(upload_csr(h, M.A, n, n, rowptr, colidx, vals, a_src)); M.A.replicated = true; M.A.gnnz = rowptr[n]; M.A.val_src = a_src != nullptr; }
    else { CHK(upload_global(h, M.A, sl, sl, n, n, rowptr, colidx, vals, a_src)); if (a_src) M.A.val_src = true; }
    CHK(refresh_dinv(h, M.A));
  }
  if (level > 0) {
    // restriction = transpose, built on the host (counting sort by column)
    auto transpose = [](int64_t nr, int64_t nc, const int32_t* rp, const int32_t* ci, const double* va,
                        std::vector<int32_t>& trp, std::vector<int32_t>& tc, std::vector<double>& tv) {
      const int64_t nnz = rp[nr];
      trp.assign(nc + 1, 0); tc.resize(nnz); tv.resize(nnz);
      for (int64_t k = 0; k < nnz; ++k) ++trp[ci[k] + 1];
      for (int64_t c = 0; c < nc; ++c) trp[c + 1] += trp[c];
      std::vector<int32_t> fill(trp.begin(), trp.end() - 1);
      for (int64_t i = 0; i < nr; ++i)
        for (int32_t k = rp[i]; k < rp[i + 1]; ++k) {
          const int32_t q = fill[ci[k]]++;
          tc[q] = (int32_t)i; tv[q] = va[k];
        }
    };
    std::vector<int32_t> trp, tc; std::vector<double> tv;
    if (!h->comm || rep_l) {                       // both levels on every rank
      CHK(upload_csr(h, M.P, p_rows, p_cols, prowptr, pcolidx, pvals, nullptr));
      transpose(p_rows, p_cols, prowptr, pcolidx, pvals, trp, tc, tv);
      CHK(upload_csr(h, M.R, p_cols, p_rows, trp.data(), tc.data(), tv.data(), nullptr));
      M.P.replicated = M.R.replicated = h->comm != nullptr;
    } else if (M.transition) {
      // my fine rows x ALL coarse columns; its transpose sums my contribution
      // to every coarse row, the all-reduce in the cycle completes it
      const int64_t r0 = sl->bounds[0][h->rank], r1 = sl->bounds[0][h->rank + 1];
      std::vector<int32_t> lrp(r1 - r0 + 1);
      for (int64_t i = r0; i <= r1; ++i) lrp[i - r0] = prowptr[i] - prowptr[r0];
      const int32_t* lc = pcolidx + prowptr[r0];
      const double* lv = pvals + prowptr[r0];
      CHK(upload_csr(h, M.P, r1 - r0, p_cols, lrp.data(), lc, lv, nullptr));
      transpose(r1 - r0, p_cols, lrp.data(), lc, lv, trp, tc, tv);
      CHK(upload_csr(h, M.R, p_cols, r1 - r0, trp.data(), tc.data(), tv.data(), nullptr));
      M.P.replicated = M.R.replicated = true;     // no halo on either
    } else {
      CHK(upload_global(h, M.P, sl, sc, p_rows, p_cols, prowptr, pcolidx, pvals, nullptr));
      transpose(p_rows, p_cols, prowptr, pcolidx, pvals, trp, tc, tv);
      CHK(upload_global(h, M.R, sc, sl, p_cols, p_rows, trp.data(), tc.data(), tv.data(), nullptr));
    }
  }
  M.emin = emin; M.emax = emax;
  M.fused = false;                       // composed from other values
  ++h->gen;
  if (h->ready) CHK(inner_prepare(h, slot));
  return 0;
} PCD_ABI_CATCH(pcd_mg_set_level)

// Rank-local form of pcd_mg_set_level for a PARTITIONED level (more rows than
// PCD_REPLICATE_BELOW; replicated levels are small by definition and keep the
// global form): this rank's rows of the level operator (NULL on the finest
// level, which is the field's own operator), its rows of the prolongation
// (fine rows owned x global coarse columns) and - when the level below is
// partitioned too - its rows of the restriction P^T (coarse rows owned x global
// fine columns; [ext PETSc] MatTranspose of the distributed P).  Row ranges are
// pcd_row_range's for a field of that many rows (velocity levels: whole nodes).
int pcd_mg_set_level_local(pcd_handle h, int slot, int level, int64_t n, int64_t nrows_local,
                           const int32_t* rowptr, const int32_t* colidx, const double* vals,
                           int64_t p_cols, const int32_t* prowptr, const int32_t* pcolidx,
                           const double* pvals, int64_t r_rows_local, const int32_t* rrowptr,
                           const int32_t* rcolidx, const double* rvals, double emin, double emax) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "mg_set_level_local: bad slot %d", slot);
  Inner& s = h->inner[slot];
  const int L = (int)s.mg.size();
  if (level < 1 || level >= L)
    return fail(PCD_ERR_ARG, "mg_set_level_local: level %d outside [1,%d) (the coarsest level is an explicit "
                             "inverse: pcd_mg_set_level)", level, L);
  if (!(emax > emin && emin > 0.0)) return fail(PCD_ERR_ARG, "mg_set_level_local: smoother needs 0 < emin < emax");
  if (!prowptr || !pcolidx || !pvals) return fail(PCD_ERR_ARG, "mg_set_level_local: prolongation missing");
  if (!rowptr && level != L - 1) return fail(PCD_ERR_ARG, "mg_set_level_local: coarse levels need an operator");
  if (rowptr && (!colidx || !vals)) return fail(PCD_ERR_ARG, "mg_set_level_local: bad operator arrays");
  if (!h->comm) return fail(PCD_ERR_STATE, "mg_set_level_local: no communicator (one GPU: pcd_mg_set_level)");
  if (h->ru.active() || h->rp.active())
    return fail(PCD_ERR_STATE, "mg_set_level_local: the engine renumbered the dofs (PCD_REORDER); "
                               "rank-local hand-over needs PCD_REORDER=none");
  HIPCHK(hipSetDevice(h->device));
  MgLevel& M = s.mg[level];
  if ((int)s.mg_r.size() != L) { s.mg_r.clear(); s.mg_r.resize(L); s.mg_r_known.assign(L, 0); }
  s.mg_r_known[level] = s.mg_r_known[level - 1] = 1;      // the caller's numbering, as handed over
  const int even = slot == PCD_KSP_A00 ? h->vel_block : 1;
  const char* lim_env = getenv("PCD_REPLICATE_BELOW");
  const int64_t limit = lim_env ? atoll(lim_env) : 60000LL;
  if (level < L - 1 && n <= limit)
    return fail(PCD_ERR_ARG, "mg_set_level_local: level %d (%lld rows) is replicated (PCD_REPLICATE_BELOW %lld): "
                             "hand it over whole with pcd_mg_set_level", level, (long long)n, (long long)limit);
  const bool rep_c = p_cols <= limit;
  s.mg_space.resize(L);
  if (s.mg_space[level].nf == 0) s.mg_space[level] = Space::field(n, h->nranks, even);
  if (s.mg_space[level].total() != n) return fail(PCD_ERR_ARG, "mg_set_level_local: level %d size mismatch", level);
  const Space* sl = &s.mg_space[level];
  const Space* sc = nullptr;
  if (!rep_c) {
    if (s.mg_space[level - 1].nf == 0) s.mg_space[level - 1] = Space::field(p_cols, h->nranks, even);
    if (s.mg_space[level - 1].total() != p_cols)
      return fail(PCD_ERR_ARG, "mg_set_level_local: level %d prolongation width mismatch", level);
    sc = &s.mg_space[level - 1];
  }
  const int me = h->rank;
  if (sl->nloc(me) != nrows_local)
    return fail(PCD_ERR_ARG, "mg_set_level_local: this rank owns %lld rows of level %d (pcd_row_range), got %lld",
                (long long)sl->nloc(me), level, (long long)nrows_local);
  auto check_cols = [&](const int32_t* rp, const int32_t* ci, int64_t nr, int64_t ncols, const char* what) -> int {
    for (int64_t k = 0; k < rp[nr]; ++k)
      if (ci[k] < 0 || ci[k] >= ncols)
        return fail(PCD_ERR_ARG, "mg_set_level_local: %s column id %d outside [0, %lld)", what, ci[k], (long long)ncols);
    return 0;
  };
  // (every refusal before anything is exchanged: the hand-over is collective)
  if (!rep_c) {
    if (!rrowptr || !rcolidx || !rvals)
      return fail(PCD_ERR_ARG, "mg_set_level_local: the level below is partitioned too: this rank's rows of the "
                               "restriction P^T are needed");
    if (sc->nloc(me) != r_rows_local)
      return fail(PCD_ERR_ARG, "mg_set_level_local: this rank owns %lld rows of level %d, the restriction has %lld",
                  (long long)sc->nloc(me), level - 1, (long long)r_rows_local);
    CHK(check_cols(rrowptr, rcolidx, r_rows_local, n, "restriction"));
  }
  CHK(check_cols(prowptr, pcolidx, nrows_local, p_cols, "prolongation"));
  M.replicated = false; M.transition = rep_c; M.n_coarse = p_cols;
  if (rowptr) {
    CHK(check_cols(rowptr, colidx, nrows_local, n, "operator"));
    auto span = [&](int64_t i) { return std::pair<int64_t, int64_t>(rowptr[i], rowptr[i + 1]); };
    M.A.gnnz = rowptr[nrows_local];        // value updates carry this rank's entries
    CHK(upload_owned(h, M.A, sl, sl, nrows_local, span, colidx, vals, nullptr));
    CHK(refresh_dinv(h, M.A));
  }
  if (rep_c) {
    // my fine rows x ALL coarse columns; the transpose sums my contribution to
    // every coarse row, the all-reduce in the cycle completes it
    CHK(upload_csr(h, M.P, nrows_local, p_cols, prowptr, pcolidx, pvals, nullptr));
    const int64_t nnz = prowptr[nrows_local];
    std::vector<int32_t> trp(p_cols + 1, 0), tc(nnz); std::vector<double> tv(nnz);
    for (int64_t k = 0; k < nnz; ++k) ++trp[pcolidx[k] + 1];
    for (int64_t c = 0; c < p_cols; ++c) trp[c + 1] += trp[c];
    std::vector<int32_t> fill(trp.begin(), trp.end() - 1);
    for (int64_t i = 0; i < nrows_local; ++i)
      for (int32_t k = prowptr[i]; k < prowptr[i + 1]; ++k) {
        const int32_t q = fill[pcolidx[k]]++;
        tc[q] = (int32_t)i; tv[q] = pvals[k];
      }
    CHK(upload_csr(h, M.R, p_cols, nrows_local, trp.data(), tc.data(), tv.data(), nullptr));
    M.P.replicated = M.R.replicated = true;       // no halo on either
  } else {
    auto pspan = [&](int64_t i) { return std::pair<int64_t, int64_t>(prowptr[i], prowptr[i + 1]); };
    auto rspan = [&](int64_t i) { return std::pair<int64_t, int64_t>(rrowptr[i], rrowptr[i + 1]); };
    CHK(upload_owned(h, M.P, sl, sc, nrows_local, pspan, pcolidx, pvals, nullptr));
    CHK(upload_owned(h, M.R, sc, sl, r_rows_local, rspan, rcolidx, rvals, nullptr));
  }
  M.emin = emin; M.emax = emax;
  M.fused = false;
  ++h->gen;
  if (h->ready) CHK(inner_prepare(h, slot));
  return 0;
} PCD_ABI_CATCH(pcd_mg_set_level_local)

// Pre-composed form of one level (see MgLevel): Wd is n_c x n, Wu is
// n x (2 n + 2 n_c) over [x1 | r_c | e_c | b].  wd_rowptr == NULL drops it.
// Partitioned levels of a multi-GPU run keep the step-by-step cycle (their
// kernels exchange halos); replicated ones may be fused.
int pcd_mg_set_fused(pcd_handle h, int slot, int level,
                     int64_t wd_rows, int64_t wd_cols, const int32_t* wd_rowptr,
                     const int32_t* wd_col, const double* wd_val,
                     int64_t wu_rows, int64_t wu_cols, const int32_t* wu_rowptr,
                     const int32_t* wu_col, const double* wu_val) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "mg_set_fused: bad slot %d", slot);
  Inner& s = h->inner[slot];
  const int L = (int)s.mg.size();
  if (level < 1 || level >= L) return fail(PCD_ERR_ARG, "mg_set_fused: level %d outside [1,%d)", level, L);
  MgLevel& M = s.mg[level];
  HIPCHK(hipSetDevice(h->device));
  ++h->gen;
  if (!wd_rowptr) { M.fused = false; return 0; }
  if (!M.P.set) return fail(PCD_ERR_STATE, "mg_set_fused: level %d has no prolongation yet", level);
  if (h->comm && !(M.P.replicated && !M.transition && (level == L - 1 || M.replicated))) {
    M.fused = false;                     // partitioned level: not fused
    return 0;
  }
  if (!wd_col || !wd_val || !wu_rowptr || !wu_col || !wu_val)
    return fail(PCD_ERR_ARG, "mg_set_fused: null arrays");
  const int64_t n = M.P.nrows, nc = M.P.ncols;
  if (wd_rows != nc || wd_cols != n || wu_rows != n || wu_cols != 2 * n + 2 * nc)
    return fail(PCD_ERR_ARG, "mg_set_fused: level %d is %lld -> %lld, got Wd %lld x %lld, Wu %lld x %lld",
                level, (long long)nc, (long long)n, (long long)wd_rows, (long long)wd_cols,
                (long long)wu_rows, (long long)wu_cols);
  if (s.nu_pre < 1 || s.nu_post < 1)
    return fail(PCD_ERR_STATE, "mg_set_fused: needs at least one pre- and one post-smoothing step");
  HIPCHK(hipStreamSynchronize(h->stream));
  // engine renumbering of the two levels: Wd is (level-1) x (level); the
  // columns of Wu run over [x1 (level) | r_c | e_c (level-1) | b (level)]
  PermCsr PD, PU;
  std::vector<double> vd_p, vu_p;
  if ((int)s.mg_r.size() == L && (s.mg_r[level].active() || s.mg_r[level - 1].active())) {
    const Reorder &rl = s.mg_r[level], &rcs = s.mg_r[level - 1];
    permute_csr(wd_rows, wd_rowptr, wd_col, rcs.rows(), rl.cols(), PD);
    vd_p.resize(PD.src.size());
    for (size_t k = 0; k < vd_p.size(); ++k) vd_p[k] = wd_val[PD.src[k]];
    std::vector<int32_t> cmap(2 * n + 2 * nc);
    for (int64_t i = 0; i < n; ++i) {
      const int32_t q = rl.active() ? rl.o2n[i] : (int32_t)i;
      cmap[i] = q; cmap[n + 2 * nc + i] = (int32_t)(n + 2 * nc + q);
    }
    for (int64_t j = 0; j < nc; ++j) {
      const int32_t q = rcs.active() ? rcs.o2n[j] : (int32_t)j;
      cmap[n + j] = (int32_t)(n + q); cmap[n + nc + j] = (int32_t)(n + nc + q);
    }
    permute_csr(wu_rows, wu_rowptr, wu_col, rl.rows(), cmap.data(), PU);
    vu_p.resize(PU.src.size());
    for (size_t k = 0; k < vu_p.size(); ++k) vu_p[k] = wu_val[PU.src[k]];
    wd_rowptr = PD.rp.data(); wd_col = PD.ci.data(); wd_val = vd_p.data();
    wu_rowptr = PU.rp.data(); wu_col = PU.ci.data(); wu_val = vu_p.data();
  }
  g_chunks_override = 64; g_want_wave = true;
  int rc_up = upload_csr(h, M.Wd, wd_rows, wd_cols, wd_rowptr, wd_col, wd_val, nullptr);
  if (!rc_up) rc_up = upload_csr(h, M.Wu, wu_rows, wu_cols, wu_rowptr, wu_col, wu_val, nullptr);
  g_chunks_override = 0; g_want_wave = false;
  CHK(rc_up);
  M.Wd.replicated = M.Wu.replicated = h->comm != nullptr;
  M.fused = true;
  if (h->ready) CHK(inner_prepare(h, slot));
  return 0;
} PCD_ABI_CATCH(pcd_mg_set_fused)

// Factor k of nfactors of an explicitly composed inner solve
// (pc_type = PCD_PC_EXPLICIT): x = W_{nfactors-1} ... W_0 b.  Factors are
// square operators on the slot's space; k == 0 starts a new chain.
int pcd_set_inner_factor(pcd_handle h, int slot, int k, int nfactors, int64_t n,
                         const int32_t* rowptr, const int32_t* colidx,
                         const double* vals) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "set_inner_factor: bad slot %d", slot);
  if (nfactors < 1 || nfactors > 8 || k < 0 || k >= nfactors)
    return fail(PCD_ERR_ARG, "set_inner_factor: factor %d of %d", k, nfactors);
  if (!rowptr || !colidx || !vals || n < 0 || n >= INT32_MAX)
    return fail(PCD_ERR_ARG, "set_inner_factor: bad arrays");
  const DCsr& A = h->mat[kSlotMat[slot]];
  if (!A.set) return fail(PCD_ERR_STATE, "set_inner_factor: operator of slot %d not set", slot);
  HIPCHK(hipSetDevice(h->device));
  Inner& s = h->inner[slot];
  HIPCHK(hipStreamSynchronize(h->stream));
  if (k == 0 || (int)s.chain.size() != nfactors) {
    for (auto& f : s.chain) f.release();
    s.chain.clear();
    s.chain.resize(nfactors);
    s.chain_stale = false;
  }
  const Space* sp = nullptr;
  if (h->comm) sp = (slot == PCD_KSP_A00) ? &h->sp_u : &h->sp_p;
  if (h->comm && sp->total() != n)
    return fail(PCD_ERR_ARG, "set_inner_factor: size %lld does not match the partitioned space", (long long)n);
  g_chunks_override = 64; g_want_wave = true;
  const Reorder* fr = slot == PCD_KSP_A00 ? &h->ru : &h->rp;
  const int rc_up = upload_global_r(h, s.chain[k], sp, sp, n, n, rowptr, colidx, vals, nullptr, fr, fr);
  g_chunks_override = 0; g_want_wave = false;
  CHK(rc_up);
  if (s.chain[k].nrows != A.nrows)
    return fail(PCD_ERR_ARG, "set_inner_factor: factor has %lld rows, the operator %lld",
                (long long)s.chain[k].nrows, (long long)A.nrows);
  ++h->gen;
  return 0;
} PCD_ABI_CATCH(pcd_set_inner_factor)

int pcd_mg_update_values(pcd_handle h, int slot, int level, const double* vals,
                         double emin, double emax, int mem) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "mg_update_values: bad slot %d", slot);
  Inner& s = h->inner[slot];
  if (level < 0 || level >= (int)s.mg.size()) return fail(PCD_ERR_STATE, "mg_update_values: level %d not set", level);
  MgLevel& M = s.mg[level];
  M.fused = false;                       // composed from the old values / bounds
  ++h->gen;
  if (vals) {
    if (!M.A.set) return fail(PCD_ERR_STATE, "mg_update_values: level %d has no operator", level);
    CHK(refresh_values(h, M.A, vals, mem));
    CHK(refresh_dinv(h, M.A));
    if (mem == PCD_MEM_HOST) HIPCHK(hipStreamSynchronize(h->stream));
  }
  if (level > 0) {
    if (!(emax > emin && emin > 0.0)) return fail(PCD_ERR_ARG, "mg_update_values: smoother needs 0 < emin < emax");
    M.emin = emin; M.emax = emax;
    ++h->gen;                        // Chebyshev coefficients are baked in
  }
  return 0;
} PCD_ABI_CATCH(pcd_mg_update_values)

int pcd_set_inner(pcd_handle h, int slot, int ksp_type, int pc_type, int max_it,
                  double rtol, double emin, double emax) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (slot < 0 || slot >= PCD_KSP_COUNT) return fail(PCD_ERR_ARG, "set_inner: bad slot %d", slot);
  if (ksp_type < PCD_KSP_PREONLY || ksp_type > PCD_KSP_CG_SR)
    return fail(PCD_ERR_ARG, "set_inner: unsupported ksp type %d", ksp_type);
  if (pc_type != PCD_PC_NONE && pc_type != PCD_PC_JACOBI && pc_type != PCD_PC_MG &&
      pc_type != PCD_PC_EXPLICIT)
    return fail(PCD_ERR_ARG, "set_inner: unsupported pc type %d", pc_type);
  if (pc_type == PCD_PC_EXPLICIT && ksp_type != PCD_KSP_PREONLY)
    return fail(PCD_ERR_ARG, "set_inner: pc explicit runs under preonly (the factors ARE the solve)");
  if (pc_type == PCD_PC_MG && ksp_type != PCD_KSP_PREONLY && ksp_type != PCD_KSP_RICHARDSON)
    return fail(PCD_ERR_ARG, "set_inner: pc mg is supported under preonly / richardson only");
  if (max_it < 0) return fail(PCD_ERR_ARG, "set_inner: negative max_it");
  if (ksp_type == PCD_KSP_CHEBYSHEV && !(emax > emin && emin > 0.0))
    return fail(PCD_ERR_ARG, "set_inner: chebyshev needs 0 < emin < emax");
  Inner& s = h->inner[slot];
  if (pc_type != PCD_PC_EXPLICIT && !s.chain.empty()) {
    HIPCHK(hipStreamSynchronize(h->stream));
    for (auto& f : s.chain) f.release();
    s.chain.clear();
  }
  ++h->gen;
  s.ksp = ksp_type; s.pc = pc_type; s.max_it = max_it; s.rtol = rtol;
  s.emin = emin; s.emax = emax;
  if (h->ready) CHK(inner_prepare(h, slot));
  return 0;
} PCD_ABI_CATCH(pcd_set_inner)

int pcd_setup(pcd_handle h) try {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(h->device));
  if (!h->mat[PCD_MAT_AP].set || !h->mat[PCD_MAT_MP].set || !h->mat[PCD_MAT_KP].set)
    return fail(PCD_ERR_STATE, "setup: Ap, Mp and Kp are required");
  if ((h->variant == PCDR_BRM1 || h->variant == PCDR_BRM2) && !h->mat[PCD_MAT_RP].set)
    return fail(PCD_ERR_STATE, "setup: PCDR variants require Rp");
  const int64_t np = h->mat[PCD_MAT_AP].nrows;            // rows of this rank
  const int64_t np_glob = h->comm ? h->sp_p.total() : np;
  for (int m : {PCD_MAT_AP, PCD_MAT_MP, PCD_MAT_KP, PCD_MAT_RP}) {
    const DCsr& A = h->mat[m];
    if (A.set && (A.nrows != np || A.ncols != np))
      return fail(PCD_ERR_ARG, "setup: operator %d is %lld x %lld, expected %lld^2", m,
                  (long long)A.nrows, (long long)A.ncols, (long long)np);
  }
  if (h->mat[PCD_MAT_A00].set) {
    if (h->n_p && h->n_p != np_glob) return fail(PCD_ERR_ARG, "setup: n_p of the split (%lld) != size of Ap (%lld)", (long long)h->n_p, (long long)np_glob);
    if (h->mat[PCD_MAT_A01].ncols != np) return fail(PCD_ERR_ARG, "setup: A01 has %lld columns, expected %lld", (long long)h->mat[PCD_MAT_A01].ncols, (long long)np);
    h->nu_loc = h->mat[PCD_MAT_A00].nrows;
    if (!h->comm) h->n_u = h->nu_loc;
    CHK(h->wu.ensure(h->nu_loc));
    CHK(h->xs.ensure(h->nu_loc + np)); CHK(h->ys.ensure(h->nu_loc + np));
  }
  h->n_p = np_glob; h->np_loc = np;
  // SubfieldBC::compute_subfield_bc keeps the owned indices and shifts them by
  // the rank offset (SubfieldBC.h:138-155); here: global -> local
  {
    const int64_t p0 = h->comm ? h->sp_p.bounds[0][h->rank] : 0;
    std::vector<int32_t> li; std::vector<double> lv;
    for (size_t k = 0; k < h->bc_host.size(); ++k) {
      int64_t g = h->bc_host[k];
      if (g < 0 || g >= np_glob) return fail(PCD_ERR_ARG, "setup: bc index %lld outside [0,%lld)", (long long)g, (long long)np_glob);
      if (h->rp.active()) g = h->rp.o2n[g];                     // engine numbering
      if (g >= p0 && g < p0 + np) { li.push_back((int32_t)(g - p0)); lv.push_back(h->bc_val_host[k]); }
    }
    h->n_bc = (int64_t)li.size();
    {
      std::vector<int32_t> slot(np, -1);
      for (size_t k = 0; k < li.size(); ++k) slot[li[k]] = (int32_t)k;   // last one wins
      CHK(h->bc_slot.ensure(np));
      if (np) HIPCHK(hipMemcpy(h->bc_slot.p, slot.data(), np * sizeof(int), hipMemcpyHostToDevice));
    }
    if (h->n_bc) {
      CHK(h->bc_idx.ensure(h->n_bc)); CHK(h->bc_val.ensure(h->n_bc));
      HIPCHK(hipMemcpy(h->bc_idx.p, li.data(), h->n_bc * sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(h->bc_val.p, lv.data(), h->n_bc * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  CHK(h->w[0].ensure(np)); CHK(h->w[1].ensure(np));
  for (int s = 0; s < PCD_KSP_COUNT; ++s) CHK(inner_prepare(h, s));
  h->ready = true; ++h->gen;
  return 0;
} PCD_ABI_CATCH(pcd_setup)

