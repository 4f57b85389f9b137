// pcd_engine.hip - host side of the MI355X PCD engine and its C ABI
// (include/pcd_engine.h).  One handle drives one GPU; all work of the hot path
// is enqueued on one HIP stream without host synchronisation unless an inner
// solver was given a relative tolerance (then the host peeks at a device flag
// every few iterations) or the caller hands over host pointers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pcd_engine.h"
#include "pcd_launch.hpp"
#include "pcd_kernels.hpp"
#include "pcd_fe.hpp"
#include "pcd_dist.hpp"
#include "pcd_peer.hpp"
#include "pcd_reorder.hpp"

using namespace pcd;

// Host-side set-up loops (sub-matrix extraction, structure detection) run on
// a few threads: chunks [begin, end) of 0..n, one std::thread each.
#include <atomic>
#include <thread>
template <class F>
static void parallel_chunks(int64_t n, F f) {
  int T = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
  if (const char* e = getenv("PCD_SETUP_THREADS")) T = std::max(1, atoi(e));
  T = (int)std::min<int64_t>(T, std::max<int64_t>(1, n / 4096));
  if (T <= 1) { f((int64_t)0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t) th.emplace_back(f, n * t / T, n * (t + 1) / T);
  for (auto& x : th) x.join();
}

// PCD_SETUP_TIMING=1: wall time of the set-up phases on stderr (diagnostics)
struct PhaseTimer {
  bool on;
  std::chrono::steady_clock::time_point t;
  PhaseTimer() : on([] { const char* e = getenv("PCD_SETUP_TIMING"); return e && e[0] == '1'; }()),
                 t(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[pcd set-up] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// ------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(expr)                                                        \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess)                                                   \
      return fail(PCD_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, \
                  hipGetErrorString(e_));                                   \
  } while (0)

#define CHK(expr)              \
  do {                         \
    int rc_ = (expr);          \
    if (rc_) return rc_;       \
  } while (0)

// ------------------------------------------------------------ device data
template <class T>
struct DBuf {
  T* p = nullptr;
  size_t n = 0;
  int ensure(size_t count) {
    if (count <= n && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess)
      return fail(PCD_ERR_NOMEM, "hipMalloc(%zu B): %s", count * sizeof(T),
                  hipGetErrorString(e));
    n = count;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
  }
};

struct DCsr {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  DBuf<int> rowptr, col;
  DBuf<double> val, dinv;
  DBuf<double> vals, val2s;   // column-scaled copies val .* dinv[col] (zero-guess first step)
  DBuf<double> dghost;        // tile kernels: reciprocal diagonal of the ghost columns (no val2s then)
  DBuf<int64_t> src;      // provenance in the caller's monolithic values
  bool has_src = false;
  // value refreshes arrive in the CALLER's entry order and go through `src`
  // (localised row block and / or engine renumbering): staged gather
  bool val_src = false;
  bool set = false;
  int lpr = 8;
  int rb = 0;             // rows per workgroup of the CSR-stream kernels (0: n/a)
  bool small_tile = false; // every rb-row block fits the half LDS tile (short rows): SpMV takes it
  bool long_rows = false; // >= 256 nonzeros per row on average: workgroup per row
  bool wave_rows = false; // composed operator with 24..255 entries per row: wave per row
  bool dense = false;     // every entry stored (explicit coarse inverse): val is row-major
  bool dense2 = false;    // the same for the scalar stencil F of F (x) I
  // multi-component structure A = F (x) I_kron (kron = 2, 3; 0: none): F
  // stored once
  int kron = 0;
  int kron_pat = 0;       // components the PATTERN admits (kron: values agree too)
  int rb2 = 0;
  bool nt2 = false;       // F (x) I launches move more than the Infinity Cache holds: stream val/col non-temporally
  int64_t nnz2 = 0;
  DBuf<int> rowptr2, col2, kron_pos;
  DBuf<double> val2;
  DBuf<int> kron_flag;
  // ROW-blocked structure (the discrete gradient A01): the rk rows of a node
  // share one column pattern - kept once per node, values node-entry-major
  // (pcd_kernels.hpp k_spmv_rk); refreshed from `val` through rk_pos
  int rk = 0, rk_rb = 0;
  bool rk_nt = false;
  int64_t rk_nnz = 0;                 // node-entries
  DBuf<int> rk_rowptr, rk_col, rk_pos;
  DBuf<double> rk_val;
  // LDS-staged vector tiles of the F (x) I kernels (pcd_kernels.hpp k_*_tc):
  // greedy row blocks, their column segments, 16-bit tile offsets per entry
  bool vt = false;
  int vt_blocks = 0, vt_rows = 0;     // rows per block (template parameter of the kernels)
  int64_t vt_nsrc = 0;                // tile slots of all blocks
  // lane-major form (operators streamed from HBM; pcd_kernels.hpp k_*_lm):
  // values in lane-major order (refreshed from val2 through vt_pos), 8 per lane
  bool vt_lm = false;
  int64_t vt_slots = 0;               // entries incl. the padding to whole lanes
  DBuf<double> vt_val;
  DBuf<int> vt_pos;                   // entry of F (row-major) behind every slot, -1: padding
  // several ranks, PCD_OVERLAP=1: the blocks that read no ghost column
  // (vt_nint of them, first in vt_list) run while the halo travels, the
  // vt_nbnd others after it has landed
  DBuf<int> vt_list;
  int vt_nint = 0, vt_nbnd = 0;
  DBuf<int4> vt_desc;
  DBuf<int> vt_tsrc;
  DBuf<unsigned short> vt_loc, vt_rowoff;
  // multi-GPU: nrows / ncols are LOCAL counts (ncols = owned columns); ghost
  // columns are numbered ncols .. ncols + nghost and live in `ghost`
  HaloPlan plan;
  bool replicated = false;   // multi-GPU: whole operator on every rank, no halo
  int64_t gnnz = 0;       // nonzeros of the GLOBAL matrix (value updates)
  DBuf<double> ghost, sendbuf;
  DBuf<int> send_idx;
  PeerHalo ph;            // one-shot peer-write channel of this operator's halo (pcd_peer.hpp)
  void release() {
    rowptr.release(); col.release(); val.release(); dinv.release();
    vals.release(); val2s.release(); dghost.release();
    src.release(); ghost.release(); sendbuf.release(); send_idx.release();
    rowptr2.release(); col2.release(); kron_pos.release();
    val2.release(); kron_flag.release(); kron = 0; kron_pat = 0; rb2 = 0; nnz2 = 0;
    vt = false; vt_blocks = 0; vt_desc.release(); vt_rowoff.release(); vt_tsrc.release(); vt_loc.release();
    vt_lm = false; vt_slots = 0; vt_val.release(); vt_pos.release();
    vt_list.release(); vt_nint = vt_nbnd = 0;
    rk = 0; rk_rb = 0; rk_nnz = 0; rk_rowptr.release(); rk_col.release(); rk_pos.release(); rk_val.release();
    plan = HaloPlan(); replicated = false;
    if (ph.dev.seq) (void)hipFree(ph.dev.seq);
    if (ph.owner) {
      // the channel's landing buffers and flags go back to the arena's free
      // list (a gamg hierarchy pushed again every Picard step would otherwise
      // fill the arena and fall back to the bootstrap path without a word)
      std::lock_guard<std::mutex> lk(peer_live_mu());
      if (peer_live().count(ph.owner)) ph.owner->give_back(ph);
    }
    ph = PeerHalo();
    set = false; nrows = ncols = nnz = 0; has_src = false; val_src = false;
  }
};

// Engine renumbering of one index space (pcd_reorder.hpp): n2o[new] = caller's
// index, o2n its inverse; empty = identity.  d_n2o: device copy for the
// gather / scatter of field vectors that cross the ABI in caller numbering.
struct Reorder {
  std::vector<int32_t> n2o, o2n;
  DBuf<int> d_n2o;
  bool active() const { return !n2o.empty(); }
  const int32_t* rows() const { return n2o.empty() ? nullptr : n2o.data(); }
  const int32_t* cols() const { return o2n.empty() ? nullptr : o2n.data(); }
  void clear() { n2o.clear(); o2n.clear(); d_n2o.release(); }
};

// one level of the geometric multigrid hierarchy (level 0 = coarsest)
struct MgLevel {
  DCsr A;                 // operator; level 0: explicit inverse; finest: unused
  DCsr P, R;              // prolongation level-1 -> level and its transpose
  double emin = 0.0, emax = 0.0;
  // multi-GPU: small levels are computed redundantly by every rank
  // (replicated); `transition` marks the finest replicated level's parent,
  // whose restriction ends in one all-reduce and whose prolongation reads the
  // full coarse vector - no halo on any coarse kernel
  bool replicated = false, transition = false;
  int64_t n_coarse = 0;
  DBuf<double> x, t0, t1, r, b;
  // Pre-composed form of this level (pcd_mg_set_fused): three launches around
  // the coarse solve instead of nu_pre + nu_post + 3,
  //   x1  = pre-smoothing of b (the ordinary kernels)
  //   r_c = Wd b    (Wd = R (I - A H1): residual + restriction in one product)
  //   x   = Wu [T | b],  T = [x1 (n) | r_c (n_c) | e_c (n_c)]
  //         (prolongation + correction + all post-smoothing steps)
  // valid for the values / smoother bounds it was composed from: any update
  // of those drops it (the step-by-step cycle takes over) until it is set again
  DCsr Wd, Wu;
  DBuf<double> T;
  bool fused = false;
  void release() {
    A.release(); P.release(); R.release();
    x.release(); t0.release(); t1.release(); r.release(); b.release();
    Wd.release(); Wu.release(); T.release(); fused = false;
  }
};

struct Inner {
  int ksp = PCD_KSP_CG, pc = PCD_PC_JACOBI, max_it = 10000;
  double rtol = 1e-12, emin = 0.5, emax = 2.0;
  int nu_pre = 2, nu_post = 2;
  std::vector<MgLevel> mg;
  // PCD_PC_EXPLICIT: the solve is x = W_{m-1} ... W_0 b with sparse factors
  // the caller composed (pcd_set_inner_factor)
  std::vector<DCsr> chain;
  bool chain_stale = false;          // the operator changed after composition
  // device scratch, sized at setup
  DBuf<double> t0, t1, t2, t3, t4;   // r,z,p,q,p'  or the Chebyshev ring
  DBuf<double> parts;                // 3 * kMaxParts
  DBuf<double> slots;                // rank-reduced scalars (multi-GPU)
  DBuf<CgState> state;
  std::vector<Space> mg_space;       // multi-GPU: row space of every level
  std::vector<Reorder> mg_r;         // engine renumbering of every level
  std::vector<char> mg_r_known;      // ... decided (identity counts)
  int last_its = 0;
  bool its_on_device = false;
  int state_idx = 0;                 // which of the two state records is final
  void release() {
    t0.release(); t1.release(); t2.release(); t3.release(); t4.release();
    parts.release(); slots.release(); state.release();
    for (auto& l : mg) l.release();
    mg.clear();
    for (auto& f : chain) f.release();
    chain.clear();
  }
};

struct FeState;                        // device operator producer (pcd_fe_host.hpp)

struct pcd_engine_s {
  FeState* fe = nullptr;
  int variant = PCD_BRM1;
  int device = 0;
  hipStream_t stream = nullptr;
  DCsr mat[PCD_MAT_COUNT];
  Inner inner[PCD_KSP_COUNT];
  int64_t n_bc = 0;
  DBuf<int> bc_idx;
  DBuf<int> bc_slot;                  // row -> position in bc_val, or -1
  std::vector<int32_t> bc_host;
  DBuf<double> bc_val;
  int64_t n_u = 0, n_p = 0, sys_nnz = 0;   // GLOBAL sizes
  int64_t nu_loc = 0, np_loc = 0;         // rows of this rank (= global on 1 GPU)
  // multi-GPU: contiguous row blocks per rank
  CommBackend* comm = nullptr;
  int rank = 0, nranks = 1;
  int vel_block = 2;                  // velocity components per node
  // engine renumbering of the velocity / pressure dofs (decided at
  // pcd_set_system; PCD_REORDER = none | auto | always, default auto)
  Reorder ru, rp, rs;                  // velocity, pressure, [u; p] system vectors
  int reorder_mode = 1;
  DBuf<double> px_s, py_s;            // staging of renumbered field vectors
  Space sp_u, sp_p, sp_sys;
  DBuf<double> loc_x, loc_y;          // local slices for host-pointer calls
  std::vector<double> bc_val_host;
  DBuf<int> perm;                     // LOCAL split position -> caller's index
  DBuf<double> sysvals, psysvals;     // staging of the caller's value arrays
  DBuf<double> valstage;              // staging of one operator's global values (several ranks)
  std::vector<int32_t> perm_glob;     // split position -> caller's index, all rows
  // (1,0) and (1,1) blocks of the system (one GPU): w = A z is then applied
  // block-wise, so the velocity block goes through its F x I fast path
  DCsr a10, a11;
  std::vector<int64_t> a11_src_host;
  bool sys_local = false;             // pcd_set_system_local: value arrays hold this rank's rows
  DBuf<double> flagbuf;
  bool a11_zero = true, p_is_a = true;
  bool ready = false;
  DBuf<double> w[2];                  // pressure work vectors (get_work_vecs)
  DBuf<double> wu;                    // velocity work vector
  DBuf<double> xs, ys;                // split-ordered in/out
  DBuf<double> io_x, io_y;            // staging for host-pointer calls
  // GMRES
  DBuf<double> V, gz, gw, gparts, gh, gy, gxs, gbs;
  DBuf<double> gH, gcs, gsn, gg;      // Hessenberg, rotations, rotated rhs (device)
  DBuf<GmresStatus> gstat;
  hipEvent_t gev[2] = {nullptr, nullptr};
  int64_t V_ld = 0;
  int V_m = 0;
  double* pinned = nullptr;           // host-pinned scratch
  size_t pinned_n = 0;
  long num_pcd = 0, num_fs = 0;
  // hipGraph replay of the fixed-iteration fieldsplit apply
  bool graph_on = false;
  hipGraphExec_t gexec = nullptr;
  hipStream_t cap_stream = nullptr;
  uint64_t gen = 1, ggen = 0;        // configuration generation / captured one
  // several ranks: a PCApply is captured only when every exchange and
  // reduction in it is a kernel of this stream (peer protocol, pcd_peer.hpp);
  // the first apply of a configuration runs eagerly and counts the others
  long boot_exchanges = 0;
  uint64_t gcheck_gen = 0;
  bool g_ok = false;
  int gmres_its = 0;
  double gmres_rnorm = 0.0;
  // pcd_probe_a00_step: event pairs around every fused Chebyshev step on the
  // finest velocity operator of an EAGER fieldsplit apply
  bool probe_on = false;
  std::vector<hipEvent_t> probe_ev;
  // interior / boundary split of a tile-kernel launch (PCD_OVERLAP=1):
  // 0 all blocks, 1 the blocks without ghost columns, 2 the others
  int ov_phase = 0;
};

typedef pcd_engine_s Engine;

static void fe_release(pcd_engine_s* h);

static const int kSlotMat[PCD_KSP_COUNT] = {PCD_MAT_AP, PCD_MAT_MP, PCD_MAT_RP,
                                            PCD_MAT_A00};

static inline int grid1d(int64_t n, int per_thread = 1, int cap = 8192) {
  int64_t g = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}
static inline int grid_rows(int64_t nrows, int lpr, int cap = 16384) {
  int rpb = kBlock / lpr;
  int64_t g = (nrows + rpb - 1) / rpb;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

static int choose_lpr(const DCsr& A) {
  double avg = A.nrows ? (double)A.nnz / (double)A.nrows : 1.0;
  int l = 4;
  while (l < 32 && l < avg) l *= 2;
  return l;
}

// every row holds exactly columns 0, 1, ..., ncols-1 in this order
static bool full_sorted_rows(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                             const int32_t* col) {
  for (int64_t r = 0; r < nrows; ++r) {
    if (rowptr[r + 1] - rowptr[r] != ncols) return false;
    const int32_t* c = col + rowptr[r];
    for (int64_t k = 0; k < ncols; ++k) if (c[k] != k) return false;
  }
  return true;
}

// rows per workgroup for the CSR-stream kernels: the largest of 256/128/64
// whose every row block fits the LDS tile; 0 = some row block is too long
static int g_max_rb = 256;              // PCD_MAX_RB: A/B switch
static int g_min_wgs = 512;             // PCD_MIN_WGS: A/B switch (see rb_for)
static int g_max_chunks = 2;            // PCD_MAX_CHUNKS: LDS-tile passes per row block
// Pre-composed operators (fused multigrid levels, explicit factors) have long
// and very uneven rows (a stacked restriction row holds hundreds of entries):
// while they are handed over, small row blocks may take this many passes
// through the tile instead of falling back to the CSR-vector kernels.
static thread_local int g_chunks_override = 0;
static thread_local bool g_want_wave = false;   // the operator being handed over is a composed one
// rows per workgroup: the largest RB whose every row block fits the LDS tile
// and that still yields `g_min_wgs` workgroups (small operators then take
// smaller row blocks: more, shorter workgroups).  Blocks of 64 rows and
// fewer may take two passes through the tile - the way out of 32-row blocks
// for operators with long rows: on the 3-D velocity block (28 entries per
// row) 64 rows in two passes beat 32 in one by 16 %; larger blocks in two
// passes lose ~1 % to smaller ones in one pass on the 2-D operators.
static int rb_for(int64_t nrows, const int32_t* rowptr, int tile) {
  int fit = 0;
  for (int rb : {256, 128, 64, 32}) {
    if (rb > g_max_rb) continue;
    bool ok = true;
    for (int64_t r = 0; r < nrows && ok; r += rb) {
      const int64_t r1 = std::min<int64_t>(r + rb, nrows);
      const int chunks = rb > 64 ? 1 : (g_chunks_override ? g_chunks_override : g_max_chunks);
      if (rowptr[r1] - rowptr[r] > (int64_t)tile * chunks) ok = false;
    }
    if (!ok) continue;
    fit = rb;                            // smaller ones fit as well
    if ((nrows + rb - 1) / rb >= g_min_wgs) return rb;
  }
  return fit;                            // 0: nothing fits; else the smallest
}
static int choose_rb(int64_t nrows, const int32_t* rowptr) {
  return rb_for(nrows, rowptr, kTile);
}
// workgroups for a stream kernel: one per row block (capped), multiple of 8
static inline int grid_stream(int64_t nrows, int rb, int cap = 1 << 20) {
  if (rb <= 0) rb = 32;                  // (callers check; never divide by zero)
  int64_t nrb = (nrows + rb - 1) / rb;
  int64_t g = std::min<int64_t>(std::max<int64_t>(nrb, 1), cap);
  return (int)((g + 7) / 8 * 8);
}
static bool g_force_vector = false;   // PCD_FORCE_CSR_VECTOR=1: A/B switch
static bool g_no_small_tile = false;  // PCD_NO_SMALL_TILE=1: A/B switch
// operators whose launches move more than this stream their matrix arrays with
// non-temporal loads (PCD_NT_BYTES; -1: never): beyond the 256 MiB Infinity Cache
static long long g_nt_bytes = 256ll << 20;
// the tile kernels of operators streamed from HBM (nt2) take the LANE-MAJOR
// form (k_*_lm: entries straight to registers with coalesced non-temporal
// loads); PCD_VT_NT=0 keeps the direct, default-policy form at every size
static int g_vt_nt = 1;
static int g_num_cus = 256;
static int ensure_pinned(Engine* h, size_t n) {
  if (n <= h->pinned_n) return 0;
  if (h->pinned) (void)hipHostFree(h->pinned);
  h->pinned = nullptr; h->pinned_n = 0;
  HIPCHK(hipHostMalloc((void**)&h->pinned, n * sizeof(double)));
  h->pinned_n = n;
  return 0;
}

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
static void halo_collect(Engine* h, const DCsr& A, const double* x,
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
static int halo_exchange(Engine* h, const DCsr& A, const double* x) {
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
static bool overlap_ok(Engine* h, const DCsr& A) {
  return g_overlap && h->comm && !A.replicated && A.vt && A.vt_nbnd > 0 && A.ph.ready &&
         static_cast<PeerBackend*>(h->comm)->usable(h->stream);
}
static int halo_send(Engine* h, const DCsr& A, const double* x) {
  PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
  if (pb->halo_send(A.ph, x, h->stream)) return fail(PCD_ERR_COMM, "halo exchange: %s", pb->err.c_str());
  return 0;
}
static int halo_wait(Engine* h, const DCsr& A) {
  PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
  if (pb->halo_wait(A.ph, h->stream)) return fail(PCD_ERR_COMM, "halo exchange: %s", pb->err.c_str());
  return 0;
}
// the blocks of a tile-kernel launch in the current phase
struct VtBlocks { int n; const int* list; };
static inline VtBlocks vt_blocks_now(const Engine* h, const DCsr& A) {
  if (h->ov_phase == 1) return VtBlocks{A.vt_nint, A.vt_list.p};
  if (h->ov_phase == 2) return VtBlocks{A.vt_nbnd, A.vt_list.p + A.vt_nint};
  return VtBlocks{A.vt_blocks, nullptr};
}

// the halos of several operators in ONE grouped exchange (one latency instead
// of one per operator); each operator keeps its own ghost buffer
struct HaloItem { const DCsr* A; const double* x; };
static int halo_exchange_group(Engine* h, std::initializer_list<HaloItem> items) {
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

// rank-local partials -> (pointer, count) the consumer kernels reduce; with
// several ranks the partials are summed into one slot and all-reduced first
struct PartsRef { const double* p; int n; };
static int reduce_global(Engine* h, double* parts, int nparts, double* slot,
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

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// the two-component kernels move 16-byte pairs; three components need no
// more than the 8-byte alignment every double* has
static inline bool kron_ok(const DCsr& A, const void* a, const void* b = nullptr,
                           const void* c = nullptr, const void* d = nullptr,
                           bool need_tile = false) {
  if (!A.kron) return false;
  // rb2 == 0: no row block of F fits the LDS tile - only the kernels that need
  // no tile (dense, wave-per-row, workgroup-per-row SpMV) may take the
  // multi-component path; the Chebyshev / first-step stream kernels may not
  // (the tile kernels choose their blocks greedily and need no such fit)
  if (!A.rb2 && !A.vt && (need_tile || !(A.dense2 || A.wave_rows || A.long_rows))) return false;
  if (A.kron != 2) return true;
  return aligned16(a) && aligned16(b) && aligned16(c) && aligned16(d);
}

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
static int spmv(Engine* h, const DCsr& A, const double* x, double* y,
                int mode = 0, const double* add = nullptr,
                const double* x2 = nullptr, int64_t n1 = 0,
                bool halo_done = false) {
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
static int spmv_other_values(Engine* h, DCsr& A, double*& other, const double* x, double* y) {
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

static int refresh_kron(Engine* h, DCsr& A);
static int refresh_dinv(Engine* h, DCsr& A) {
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
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------ inner KSPs
static int inner_prepare(Engine* h, int slot) {
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
      break;
    case PCD_KSP_RICHARDSON:
      CHK(s.t0.ensure(n));
      break;
    default: break;
  }
  return 0;
}

static int launch_cheb_step(Engine* h, const DCsr& A, const double* dinv,
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
                       A.vt_loc.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p, nloc, epi_nt)
    // (the step's five vectors beyond the Infinity Cache: streamed past it)
    const int epi_nt = g_nt_bytes >= 0 && 40.0 * (double)A.nrows > (double)g_nt_bytes;
    if (A.kron == 2) PCD_CHEB_LM(2); else PCD_CHEB_LM(3);
#undef PCD_CHEB_LM
  } else if (tile) {
    const VtBlocks vb = vt_blocks_now(h, A);
    if (!vb.n) return;
    const int gt = grid_stream(vb.n, 1);
    const int nloc = (int)(A.ncols / A.kron);
#define PCD_CHEB_TC(NC)                                                                        \
    hipLaunchKernelGGL((k_cheb_step_tc<NC, 64>), dim3(gt), dim3(kBlock), 0, h->stream,        \
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.val2.p,       \
                       A.vt_loc.p, dinv, b, pm, pk, pn, c0, c1, c2, A.ghost.p, nloc)
    if (A.kron == 2) PCD_CHEB_TC(2); else PCD_CHEB_TC(3);
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
static bool can_fuse_first(const Engine* h, const DCsr& A, const double* dinv) {
  (void)h;
  // (a multi-component operator none of whose kernels takes the step falls
  // through to the scalar stream kernel and its column-scaled values)
  return A.rb && dinv != nullptr;
}
static int launch_cheb_first(Engine* h, const DCsr& A, const double* dinv,
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
#define PCD_FIRST_TC(NC)                                                                       \
    hipLaunchKernelGGL((k_cheb_first_tc<NC, 64>), dim3(gt), dim3(kBlock), 0, h->stream,       \
                       vb.n, vb.list, A.vt_desc.p, A.vt_rowoff.p, A.vt_tsrc.p, A.val2.p,       \
                       A.vt_loc.p, dinv, b, p0, pn, s, c1, c2, ghost, (int)(A.ncols / A.kron), \
                       A.dghost.p ? A.dghost.p : dinv)
    if (A.kron == 2) PCD_FIRST_TC(2); else PCD_FIRST_TC(3);
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
static int solve_cg_stream(Engine* h, const DCsr& A, Inner& s, const double* b,
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

static int solve_cg(Engine* h, const DCsr& A, Inner& s, const double* b,
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
static int solve_cg_sr(Engine* h, const DCsr& A, Inner& s, const double* b,
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
static int solve_cheb(Engine* h, const DCsr& A, Inner& s, const double* b,
                      double* x, double out_scale = 1.0) {
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

static int solve_rich(Engine* h, const DCsr& A, Inner& s, const double* b,
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
static int mg_smooth(Engine* h, const DCsr& A, double emin, double emax, int nu,
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
static int mg_vcycle(Engine* h, const DCsr& Afine, Inner& s, int l,
                     const double* b, double** out, double* target = nullptr) {
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
static int solve_mg(Engine* h, const DCsr& A, Inner& s, const double* b,
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
static int inner_solve(Engine* h, int slot, const double* b, double* x,
                       double out_scale = 1.0, bool* scaled = nullptr) {
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
static int apply_bc_dev(Engine* h, double* x) {
  if (h->n_bc == 0) return 0;
  hipLaunchKernelGGL(k_bc_set, dim3(grid1d(h->n_bc, 1, 1 << 30)), dim3(kBlock), 0,
                     h->stream, (int)h->n_bc, h->bc_idx.p, h->bc_val.p, x);
  HIPCHK(hipGetLastError());
  return 0;
}

// The four PCPYTHON apply bodies on device pointers (x, y distinct, n_p long)
static int pcd_apply_dev(Engine* h, const double* x, double* y) {
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
static int fs_apply_eager(Engine* h, const double* x, double* y) {
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
static bool graph_capturable(const Engine* h) {
  const bool reaction = h->variant == PCDR_BRM1 || h->variant == PCDR_BRM2;
  for (int slot : {PCD_KSP_AP, PCD_KSP_MP, PCD_KSP_RP, PCD_KSP_A00}) {
    if (slot == PCD_KSP_RP && !reaction) continue;
    const Inner& s = h->inner[slot];
    if (s.pc != PCD_PC_MG && (s.ksp == PCD_KSP_CG || s.ksp == PCD_KSP_CG_SR) && s.rtol > 0.0) return false;
  }
  return true;
}

static int fs_apply_split(Engine* h, const double* x, double* y) {
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

// ------------------------------------------------------------ host <-> dev
struct IoMap {
  Engine* h;
  const double* dx = nullptr;
  double* dy = nullptr;
  double* hy = nullptr;
  size_t ny = 0;
  int mem;
};

static int io_begin(Engine* h, IoMap& io, const double* x, size_t nx, double* y,
                    size_t ny, int mem, bool y_in = false) {
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

static int io_end(IoMap& io) {
  if (io.mem == PCD_MEM_DEVICE) return 0;
  Engine* h = io.h;
  if (io.hy)
    HIPCHK(hipMemcpyAsync(io.hy, io.dy, io.ny * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

static bool g_no_kron = false;         // PCD_NO_KRON2=1: A/B switch

// After new values arrived: refresh F's values and verify that all
// components still carry the same numbers; otherwise drop to the general path.
static int refresh_kron(Engine* h, DCsr& A) {
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
static bool kron_pattern(int nc, int64_t nrows, int64_t ncols, const int32_t* rowptr,
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
static int build_vec_tile(Engine* h, DCsr& A, int nc, int64_t nn, int64_t nloc,
                          const std::vector<int32_t>& rpc, const std::vector<int32_t>& cc) {
  A.vt = false; A.vt_blocks = 0;
  // (read per operator, defaults restored when a variable is gone: the A/B
  // tests of one process must not leak their switches into later engines)
  { const char* e = getenv("PCD_VT_NT"); g_vt_nt = e ? atoi(e) : 1; }
  // operators streamed from HBM: lane-major entries, straight to registers
  // (the form is fixed with the layout)
  A.vt_lm = A.nt2 && g_vt_nt;
  const int kVtRows = A.vt_lm ? lm_rows(nc) : 64;
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
  // (profiles/r04_q_vt_sweep*.txt, r04_e_*, r04_f_*): N = 32 (90 MB per
  // launch, cache-resident) gather kernel 34.9, tile kernel 27.1-29.6;
  // N = 48 (428 MB per launch, HBM-bound) 110.8 against 91.3 in the staged
  // non-temporal form (123.9 with default-policy loads, 157 with
  // non-temporal loads read by rows).
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
static int detect_kron(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
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
static int detect_rowkron(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
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

static int upload_csr(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                      const int32_t* rowptr, const int32_t* col,
                      const double* val, const int64_t* src) {
  const int64_t nnz = rowptr[nrows];
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
static int ensure_space(Engine* h, Space& sp, int64_t n, bool velocity, const char* what) {
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
static int upload_global(Engine* h, DCsr& A, const Space* rs, const Space* cs,
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
static int upload_global_r(Engine* h, DCsr& A, const Space* rs, const Space* cs,
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

static int upload_perm(Engine* h, Reorder& r) {
  if (!r.active()) return 0;
  CHK(r.d_n2o.ensure(r.n2o.size()));
  HIPCHK(hipMemcpy(r.d_n2o.p, r.n2o.data(), r.n2o.size() * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

// new values of a handed-over operator: one GPU copies, several ranks stage
// the caller's global array and gather their entries
static int refresh_values(Engine* h, DCsr& A, const double* vals, int mem) {
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
static int slice_in(Engine* h, const Space& sp, const double* full, double* loc) {
  int64_t off = 0;
  for (int f = 0; f < sp.nf; ++f) {
    const int64_t b0 = sp.bounds[f][h->rank], len = sp.bounds[f][h->rank + 1] - b0;
    if (len) HIPCHK(hipMemcpyAsync(loc + off, full + sp.goff[f] + b0, len * sizeof(double),
                                   hipMemcpyDeviceToDevice, h->stream));
    off += len;
  }
  return 0;
}
static int slice_out(Engine* h, const Space& sp, const double* loc, double* full) {
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
static void values_changed(Engine* h, int which) {
  for (int slot = 0; slot < PCD_KSP_COUNT; ++slot) {
    if (kSlotMat[slot] != which) continue;
    Inner& s = h->inner[slot];
    if (!s.mg.empty() && s.mg.back().fused) { s.mg.back().fused = false; ++h->gen; }
    if (!s.chain.empty()) s.chain_stale = true;
  }
}

// ================================================================= C ABI
extern "C" {

const char* pcd_last_error(void) { return g_err; }

int pcd_create(pcd_handle* out, int variant, int device) {
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
  Engine* h = new (std::nothrow) Engine();
  if (!h) return fail(PCD_ERR_NOMEM, "create: out of host memory");
  h->variant = variant; h->device = device;
  *out = h;
  return 0;
}

int pcd_destroy(pcd_handle h) {
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
}

int pcd_set_stream(pcd_handle h, void* hip_stream) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  HIPCHK(hipStreamSynchronize(h->stream));
  h->stream = reinterpret_cast<hipStream_t>(hip_stream);
  ++h->gen;
  return 0;
}

// several ranks, peer protocol: did a wait for a neighbour give up since the
// last check?  (the kernels never hang: they set an error word and go on)
static int peer_check(Engine* h) {
  if (!h->comm || !h->comm->peer()) return 0;
  PeerBackend* pb = static_cast<PeerBackend*>(h->comm);
  if (pb->take_error(h->stream)) return fail(PCD_ERR_COMM, "%s", pb->err.c_str());
  return 0;
}

int pcd_synchronize(pcd_handle h) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  HIPCHK(hipStreamSynchronize(h->stream));
  return peer_check(h);
}

int pcd_set_csr(pcd_handle h, int which, int64_t nrows, int64_t ncols,
                const int32_t* rowptr, const int32_t* colidx,
                const double* vals) {
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
}

int pcd_row_range(pcd_handle h, int velocity, int64_t n_global, int64_t* r0, int64_t* r1) {
  if (!h || !r0 || !r1 || n_global < 0) return fail(PCD_ERR_ARG, "row_range: bad arguments");
  if (!h->comm) { *r0 = 0; *r1 = n_global; return 0; }
  // the same rule for every space of the engine - the fields and the levels of
  // their multigrid hierarchies: even cuts, on node boundaries for velocities
  const std::vector<int64_t> b = Space::cut(n_global, h->nranks, velocity ? h->vel_block : 1);
  *r0 = b[h->rank]; *r1 = b[h->rank + 1];
  return 0;
}

int pcd_set_csr_local(pcd_handle h, int which, int64_t nrows_global, int64_t ncols_global,
                      int64_t nrows_local, const int32_t* rowptr, const int32_t* colidx,
                      const double* vals) {
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
}

int pcd_update_values(pcd_handle h, int which, const double* vals, int mem) {
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
}

// host-side MatCreateSubMatrix with value provenance
static void extract_block(int64_t nr, const int32_t* rows, const int32_t* rowptr,
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

static int gather_block_values(Engine* h, DCsr& A, const double* dvals) {
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
static int decide_reordering(Engine* h, int64_t n, const int32_t* rowptr, const int32_t* colidx,
                             int64_t n_u, const int32_t* is_u, int64_t n_p, const int32_t* is_p,
                             std::vector<int32_t>& isu_r, std::vector<int32_t>& isp_r) {
  { const char* e = getenv("PCD_REORDER");
    if (e) h->reorder_mode = !strcmp(e, "none") ? 0 : !strcmp(e, "always") ? 2 : 1; }
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
      if (h->reorder_mode == 2 || m > 0.1) {
        std::vector<int32_t> nodes = rcm_order(nn, gp, gc);
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

int pcd_set_reorder(pcd_handle h, int mode) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (mode < 0 || mode > 2) return fail(PCD_ERR_ARG, "set_reorder: mode 0 (never), 1 (auto) or 2 (always)");
  h->reorder_mode = mode;
  return 0;
}

int pcd_update_system(pcd_handle h, const double* vals, const double* pvals,
                      int mem) {
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
}

int pcd_set_system(pcd_handle h, int64_t n, const int32_t* rowptr,
                   const int32_t* colidx, const double* vals,
                   const double* pvals, int64_t n_u, const int32_t* is_u,
                   int64_t n_p, const int32_t* is_p) {
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
}

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
                         const double* vals, const double* pvals) {
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
}

int pcd_set_bc(pcd_handle h, int64_t n_bc, const int32_t* idx, const double* vals) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (n_bc < 0 || (n_bc && (!idx || !vals))) return fail(PCD_ERR_ARG, "set_bc: bad arrays");
  h->bc_host.assign(idx, idx + n_bc);
  h->bc_val_host.assign(vals, vals + n_bc);
  h->ready = false; ++h->gen;             // filtered / uploaded by pcd_setup
  return 0;
}

// ---- multigrid hierarchy ------------------------------------------------
int pcd_mg_begin(pcd_handle h, int slot, int nlevels, int nu_pre, int nu_post) {
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
}

int pcd_mg_set_level_cuts(pcd_handle h, int slot, int level, int64_t n, const int64_t* bounds) {
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
}

int pcd_mg_set_level(pcd_handle h, int slot, int level, int64_t n,
                     const int32_t* rowptr, const int32_t* colidx,
                     const double* vals, int64_t p_rows, int64_t p_cols,
                     const int32_t* prowptr, const int32_t* pcolidx,
                     const double* pvals, double emin, double emax) {
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
}

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
                           const int32_t* rcolidx, const double* rvals, double emin, double emax) {
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
}

// Pre-composed form of one level (see MgLevel): Wd is n_c x n, Wu is
// n x (2 n + 2 n_c) over [x1 | r_c | e_c | b].  wd_rowptr == NULL drops it.
// Partitioned levels of a multi-GPU run keep the step-by-step cycle (their
// kernels exchange halos); replicated ones may be fused.
int pcd_mg_set_fused(pcd_handle h, int slot, int level,
                     int64_t wd_rows, int64_t wd_cols, const int32_t* wd_rowptr,
                     const int32_t* wd_col, const double* wd_val,
                     int64_t wu_rows, int64_t wu_cols, const int32_t* wu_rowptr,
                     const int32_t* wu_col, const double* wu_val) {
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
}

// Factor k of nfactors of an explicitly composed inner solve
// (pc_type = PCD_PC_EXPLICIT): x = W_{nfactors-1} ... W_0 b.  Factors are
// square operators on the slot's space; k == 0 starts a new chain.
int pcd_set_inner_factor(pcd_handle h, int slot, int k, int nfactors, int64_t n,
                         const int32_t* rowptr, const int32_t* colidx,
                         const double* vals) {
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
}

int pcd_mg_update_values(pcd_handle h, int slot, int level, const double* vals,
                         double emin, double emax, int mem) {
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
}

int pcd_set_inner(pcd_handle h, int slot, int ksp_type, int pc_type, int max_it,
                  double rtol, double emin, double emax) {
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
}

int pcd_setup(pcd_handle h) {
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
}

// Host-pointer calls always carry GLOBAL vectors; with several ranks each rank
// works on its slice and the result is summed back into a full vector.
// Device-pointer calls carry the rank's LOCAL slice (split ordering for
// system vectors) when several ranks are active.
struct FieldIo {
  Engine* h; IoMap io; const Space* sp; int64_t nglob, nloc;
  const double* lx = nullptr; double* ly = nullptr;
  const Reorder* ry = nullptr;         // renumbering of the output space (or null)
  double* y_caller = nullptr;          // where the caller-numbered result goes
  double* y_engine = nullptr;          // the global vector in engine numbering
};

// Field vectors cross the ABI in the CALLER's numbering (global vectors); the
// engine renumbering of their space (rx / ry, may be null) is applied here.
// Device-pointer calls of a partitioned engine carry the rank's slice in the
// engine's own order and pass through untouched.
static int fio_begin(FieldIo& f, Engine* h, const Space* spx, int64_t nx_glob, int64_t nx_loc,
                     const Space* spy, int64_t ny_glob, int64_t ny_loc,
                     const double* x, double* y, int mem, bool y_in = false,
                     const Reorder* rx = nullptr, const Reorder* ry = nullptr) {
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

static int fio_end(FieldIo& f) {
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

int pcd_apply(pcd_handle h, const double* x, double* y, int mem) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!h->ready) return fail(PCD_ERR_STATE, "apply: call pcd_setup first");
  if (!x || !y || x == y) return fail(PCD_ERR_ARG, "apply: x and y must be distinct non-null vectors");
  FieldIo f;
  CHK(fio_begin(f, h, &h->sp_p, h->n_p, h->np_loc, &h->sp_p, h->n_p, h->np_loc, x, y, mem, false,
                &h->rp, &h->rp));
  CHK(pcd_apply_dev(h, f.lx, f.ly));
  return fio_end(f);
}

int pcd_fieldsplit_apply(pcd_handle h, const double* x, double* y, int mem) {
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
}

// The dominant kernel where it runs: `reps` EAGER fieldsplit applies (device
// vectors) with an event pair around every fused Chebyshev step on the finest
// velocity operator - the caches in the state the multigrid cycle leaves them
// in, where a back-to-back loop on one operator keeps them warm.  An event
// pair adds about a microsecond of its own.
int pcd_probe_a00_step(pcd_handle h, const double* x, double* y, int reps, double* us, int* launches) {
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
}

// sqrt(v.v) over all ranks; synchronises
static int dev_norm(Engine* h, int64_t n, const double* v, double* out) {
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
static int apply_system(Engine* h, const double* z, double* w) {
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
                    double* rnorm) {
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
}

// which -> (row space, column space) of a stored operator
static void mat_spaces(Engine* h, int which, const Space** rs, const Space** cs) {
  switch (which) {
    case PCD_MAT_A00: *rs = *cs = &h->sp_u; break;
    case PCD_MAT_A01: *rs = &h->sp_u; *cs = &h->sp_p; break;
    case PCD_MAT_A: *rs = *cs = &h->sp_sys; break;
    default: *rs = *cs = &h->sp_p; break;
  }
}

int pcd_spmv(pcd_handle h, int which, const double* x, double* y, int mem) {
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
}

int pcd_inner_solve(pcd_handle h, int slot, const double* b, double* x, int mem) {
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
}

int pcd_apply_bc(pcd_handle h, double* x, int mem) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (!x) return fail(PCD_ERR_ARG, "apply_bc: null vector");
  if (!h->ready) return fail(PCD_ERR_STATE, "apply_bc: call pcd_setup first");
  FieldIo f;
  CHK(fio_begin(f, h, &h->sp_p, h->n_p, h->np_loc, &h->sp_p, h->n_p, h->np_loc, nullptr, x, mem, true,
                nullptr, &h->rp));
  CHK(apply_bc_dev(h, f.ly));
  return fio_end(f);
}

int pcd_get_info(pcd_handle h, int key, double* out) {
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
}

int pcd_set_velocity_block(pcd_handle h, int ncomp) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (ncomp < 1 || ncomp > 3) return fail(PCD_ERR_ARG, "set_velocity_block: 1..3 components");
  for (auto& m : h->mat) if (m.set) return fail(PCD_ERR_STATE, "set_velocity_block: call before any operator is handed over");
  h->vel_block = ncomp;
  return 0;
}

// Streaming bandwidth of this GPU as a kernel of this library sees it:
// kind 0 copy, 1 triad, 2 read-only, 3 read-mostly (6 % writes), 4 read-only
// with non-temporal loads, on arrays of
// `bytes` each (>= 256 MiB: beyond the
// Infinity Cache), best of `reps` launches, timed with events on the engine's
// stream.  *gbs = bytes moved (reads + writes) per second / 1e9.
int pcd_bandwidth_probe(pcd_handle h, int kind, int64_t bytes, int reps, double* gbs) {
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
}

int pcd_graph_enable(pcd_handle h, int on) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  h->graph_on = on != 0;
  return 0;
}

// ---- multi-GPU bootstrap ---------------------------------------------------
int pcd_comm_unique_id(void* out128) {
  if (!out128) return fail(PCD_ERR_ARG, "comm_unique_id: null buffer");
  std::string err;
  if (!rccl_api().load(err)) return fail(PCD_ERR_COMM, "%s", err.c_str());
  ncclUniqueId id;
  ncclResult_t r = rccl_api().GetUniqueId(&id);
  if (r != ncclSuccess) return fail(PCD_ERR_COMM, "ncclGetUniqueId: %s", rccl_api().GetErrorString(r));
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(out128, &id, sizeof id);
  return 0;
}

static int comm_attach(Engine* h, CommBackend* c, int rank, int nranks) {
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
static CommBackend* wrap_peer(Engine* h, CommBackend* boot, int rank, int nranks, ThreadGroup* tg,
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

int pcd_comm_init(pcd_handle h, int rank, int nranks, const void* id) {
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
}

int pcd_comm_init_host(pcd_handle h, int rank, int nranks, pcd_host_allreduce_fn allreduce,
                       pcd_host_exchange_fn exchange, void* ctx) {
  if (!h) return fail(PCD_ERR_ARG, "null handle");
  if (nranks < 2 || rank < 0 || rank >= nranks || !allreduce || !exchange)
    return fail(PCD_ERR_ARG, "comm_init_host: bad rank / size / callbacks");
  for (auto& m : h->mat) if (m.set) return fail(PCD_ERR_STATE, "comm_init: call before any operator is handed over");
  HIPCHK(hipSetDevice(h->device));
  HostBackend* b = new HostBackend();
  b->ar = allreduce; b->ex = exchange; b->ctx = ctx;
  return comm_attach(h, wrap_peer(h, b, rank, nranks, nullptr, true), rank, nranks);
}

// test-only backend: `nranks` engines of ONE process (one thread each) on one
// GPU exchange through device copies; *group is created by the first caller
int pcd_comm_init_threads(pcd_handle h, int rank, int nranks, void** group) {
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
}

// Host-only view of the partitioning (no device call): the row block, the
// localised columns and the halo plan rank `rank` of `nranks` derives from a
// global CSR.  Lets multi-process CPU tests drive the same C++ that the GPU
// path uses.  Output arrays must hold nrows+1 / nnz / ncols / nranks+1 entries.
int pcd_dist_probe(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                   const int32_t* colidx, const double* vals, int rank,
                   int nranks, int even_rows, int even_cols, int64_t* counts,
                   int32_t* out_rowptr, int32_t* out_col, double* out_val,
                   int32_t* send_peers, int32_t* send_off, int32_t* send_idx,
                   int32_t* recv_peers, int32_t* recv_off) {
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
}

}  // extern "C"

#include "pcd_fe_host.hpp"
