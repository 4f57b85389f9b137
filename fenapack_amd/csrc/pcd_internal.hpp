#pragma once
// pcd_internal.hpp - what the translation units of the MI355X PCD engine share
// (data structures, set-up switches, prototypes); the host side behind the C ABI
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

#pragma GCC visibility push(default)
#include "../../include/pcd_engine.h"
#pragma GCC visibility pop
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
#include <exception>
#include <new>
#include <thread>
template <class F>
inline void parallel_chunks(int64_t n, F f) {
  int T = (int)std::min<int64_t>(std::max(1u, std::thread::hardware_concurrency()), 32);
  if (const char* e = getenv("PCD_SETUP_THREADS")) T = std::max(1, atoi(e));
  T = (int)std::min<int64_t>(T, std::max<int64_t>(1, n / 4096));
  if (T <= 1) { f((int64_t)0, n); return; }
  // an exception of a worker (bad_alloc of its scratch) must not end the
  // process: it is kept, the others finish, and it is thrown again on the
  // caller's thread - where the export's PCD_ABI_CATCH turns it into a status
  std::vector<std::exception_ptr> ex((size_t)T);
  std::vector<std::thread> th;
  th.reserve((size_t)T);
  auto guarded = [&f, &ex](int t, int64_t b, int64_t e) {
    try { f(b, e); } catch (...) { ex[(size_t)t] = std::current_exception(); }
  };
  struct Joiner {                        // (a failed thread start joins those already running)
    std::vector<std::thread>& th;
    ~Joiner() { for (auto& x : th) if (x.joinable()) x.join(); }
  };
  {
    Joiner j{th};
    for (int t = 0; t < T; ++t) th.emplace_back(guarded, t, n * t / T, n * (t + 1) / T);
  }
  for (auto& e : ex) if (e) std::rethrow_exception(e);
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
inline thread_local char g_err[1024] = "";

inline int fail(int code, const char* fmt, ...) {
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

// Nothing throws across the ABI (SURVEY 8(b) "Errors": "never throws across
// the ABI, never aborts"; the reference turns every Python exception of its
// PCPYTHON context into a PETSc error code: fenapack/field_split.py:135-140).
// Every extern "C" entry point is a function-try-block closed by this macro:
// std::bad_alloc of the host-side set-up containers -> PCD_ERR_NOMEM, anything
// else -> PCD_ERR_INTERNAL; the message is in pcd_last_error(), the handle
// stays destroyable (pcd_destroy only releases what is there).
#define PCD_ABI_CATCH(name)                                                                  \
  catch (const std::bad_alloc&) { return fail(PCD_ERR_NOMEM, #name ": out of host memory"); } \
  catch (const std::exception& e_) { return fail(PCD_ERR_INTERNAL, #name ": %s", e_.what()); } \
  catch (...) { return fail(PCD_ERR_INTERNAL, #name ": unknown exception"); }

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

// Chebyshev-Jacobi solves of m steps in ONE launch (k_cheb_patch): the rows
// are cut into graph clusters, every cluster carries the nodes within m edges
// (its PATCH, ordered by distance) and the matrix rows of the nodes within
// m - 1 edges in ELL form with 16-bit patch-local columns; a workgroup runs the
// m steps of its patch in LDS - step k on the nodes within m - k edges - and
// writes its own rows.  For operators at the launch-latency floor (a dependent
// launch on <= 10^5 rows costs 4-6 us whatever it computes).
constexpr int kChebPatchMaxM = 8;
constexpr int kChebPatchNodes = 1536;        // patch nodes a workgroup holds (5 x 8 B each in LDS)
struct ChebPatch {
  bool ready = false;
  int tried_m = 0;                       // a build for this m was declined on the pattern in force: not retried
  int m = 0, nblocks = 0, wmax = 0;      // wmax: the widest ELL row of all patches
  int64_t nslots = 0;
  DBuf<int> node;                 // patch node lists, block after block
  DBuf<int4> desc;                // x node offset, y ELL offset, z patch nodes, w padded rows | width << 16
  DBuf<int> cnt;                  // per block kChebPatchMaxM + 1: nodes within k edges
  DBuf<unsigned short> col;       // ELL columns (patch-local)
  DBuf<double> val;               // ELL values
  DBuf<int> pos;                  // ... their positions in the operator's values (-1: padding)
  void release() {
    ready = false; tried_m = 0; m = 0; nblocks = 0; wmax = 0; nslots = 0;
    node.release(); desc.release(); cnt.release(); col.release(); val.release(); pos.release();
  }
};

struct DCsr {
  int64_t nrows = 0, ncols = 0, nnz = 0;
  ChebPatch cp;
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
    cp.release();
    rk = 0; rk_rb = 0; rk_nnz = 0; rk_rowptr.release(); rk_col.release(); rk_pos.release(); rk_val.release();
    plan = HaloPlan(); replicated = false;
    if (ph.dev.seq) (void)hipFree(ph.dev.seq);
    // the channel's landing buffers and flags go back to the arena's free
    // list (a gamg hierarchy pushed again every Picard step would otherwise
    // fill the arena and fall back to the bootstrap path without a word)
    peer_give_back(ph);
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

struct FeState;                        // device operator producer (pcd_producer_abi.hpp)

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

void fe_release(pcd_engine_s* h);

inline const int kSlotMat[PCD_KSP_COUNT] = {PCD_MAT_AP, PCD_MAT_MP, PCD_MAT_RP,
                                            PCD_MAT_A00};

inline int grid1d(int64_t n, int per_thread = 1, int cap = 8192) {
  int64_t g = (n + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}
inline int grid_rows(int64_t nrows, int lpr, int cap = 16384) {
  int rpb = kBlock / lpr;
  int64_t g = (nrows + rpb - 1) / rpb;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

inline int choose_lpr(const DCsr& A) {
  double avg = A.nrows ? (double)A.nnz / (double)A.nrows : 1.0;
  int l = 4;
  while (l < 32 && l < avg) l *= 2;
  return l;
}

// every row holds exactly columns 0, 1, ..., ncols-1 in this order
inline bool full_sorted_rows(int64_t nrows, int64_t ncols, const int32_t* rowptr,
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
inline int g_max_rb = 256;              // PCD_MAX_RB: A/B switch
inline int g_min_wgs = 512;             // PCD_MIN_WGS: A/B switch (see rb_for)
inline int g_max_chunks = 2;            // PCD_MAX_CHUNKS: LDS-tile passes per row block
// Pre-composed operators (fused multigrid levels, explicit factors) have long
// and very uneven rows (a stacked restriction row holds hundreds of entries):
// while they are handed over, small row blocks may take this many passes
// through the tile instead of falling back to the CSR-vector kernels.
inline thread_local int g_chunks_override = 0;
inline thread_local bool g_want_wave = false;   // the operator being handed over is a composed one
// rows per workgroup: the largest RB whose every row block fits the LDS tile
// and that still yields `g_min_wgs` workgroups (small operators then take
// smaller row blocks: more, shorter workgroups).  Blocks of 64 rows and
// fewer may take two passes through the tile - the way out of 32-row blocks
// for operators with long rows: on the 3-D velocity block (28 entries per
// row) 64 rows in two passes beat 32 in one by 16 %; larger blocks in two
// passes lose ~1 % to smaller ones in one pass on the 2-D operators.
inline int rb_for(int64_t nrows, const int32_t* rowptr, int tile) {
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
inline int choose_rb(int64_t nrows, const int32_t* rowptr) {
  return rb_for(nrows, rowptr, kTile);
}
// workgroups for a stream kernel: one per row block (capped), multiple of 8
inline int grid_stream(int64_t nrows, int rb, int cap = 1 << 20) {
  if (rb <= 0) rb = 32;                  // (callers check; never divide by zero)
  int64_t nrb = (nrows + rb - 1) / rb;
  int64_t g = std::min<int64_t>(std::max<int64_t>(nrb, 1), cap);
  return (int)((g + 7) / 8 * 8);
}
inline bool g_force_vector = false;   // PCD_FORCE_CSR_VECTOR=1: A/B switch
inline bool g_no_small_tile = false;  // PCD_NO_SMALL_TILE=1: A/B switch
// operators whose launches move more than this stream their matrix arrays with
// non-temporal loads (PCD_NT_BYTES; -1: never): beyond the 256 MiB Infinity Cache
inline long long g_nt_bytes = 256ll << 20;
// the tile kernels of operators streamed from HBM (nt2) take the LANE-MAJOR
// form (k_*_lm: entries straight to registers with coalesced non-temporal
// loads); PCD_VT_NT=0 keeps the direct, default-policy form at every size
inline int g_vt_nt = 1;
inline int g_num_cus = 256;
inline int ensure_pinned(Engine* h, size_t n) {
  if (n <= h->pinned_n) return 0;
  if (h->pinned) (void)hipHostFree(h->pinned);
  h->pinned = nullptr; h->pinned_n = 0;
  HIPCHK(hipHostMalloc((void**)&h->pinned, n * sizeof(double)));
  h->pinned_n = n;
  return 0;
}


// ---- shared small structs ---------------------------------------------------
#include <initializer_list>
// the blocks of a tile-kernel launch in the current phase
struct VtBlocks { int n; const int* list; };

// the halos of several operators in ONE grouped exchange (one latency instead
// of one per operator); each operator keeps its own ghost buffer
struct HaloItem { const DCsr* A; const double* x; };

// rank-local partials -> (pointer, count) the consumer kernels reduce; with
// several ranks the partials are summed into one slot and all-reduced first
struct PartsRef { const double* p; int n; };

// ------------------------------------------------------------ host <-> dev
struct IoMap {
  Engine* h;
  const double* dx = nullptr;
  double* dy = nullptr;
  double* hy = nullptr;
  size_t ny = 0;
  int mem;
};

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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the two-component kernels move 16-byte pairs; three components need no
// more than the 8-byte alignment every double* has
inline bool kron_ok(const DCsr& A, const void* a, const void* b = nullptr,
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

inline bool g_no_kron = false;         // PCD_NO_KRON2=1: A/B switch
// ---- functions the translation units call across each other -----------------
// pcd_apply.hip (launches, inner solvers, multigrid cycle, apply bodies),
// pcd_setup.hip (hand-over, structure detection, hierarchies), pcd_krylov.hip
// (outer GMRES), pcd_abi.hip (lifetime, vectors across the ABI, info,
// communicators), pcd_producer.hip (device operator producer)
int apply_configure_constants();
void halo_collect(Engine* h, const DCsr& A, const double* x,
                         std::vector<Msg>& sends, std::vector<Msg>& recvs);
int halo_exchange(Engine* h, const DCsr& A, const double* x);
bool overlap_ok(Engine* h, const DCsr& A);
int halo_send(Engine* h, const DCsr& A, const double* x);
int halo_wait(Engine* h, const DCsr& A);
int halo_exchange_group(Engine* h, std::initializer_list<HaloItem> items);
int reduce_global(Engine* h, double* parts, int nparts, double* slot,
                         PartsRef* out);
int spmv(Engine* h, const DCsr& A, const double* x, double* y,
                int mode = 0, const double* add = nullptr,
                const double* x2 = nullptr, int64_t n1 = 0,
                bool halo_done = false);
int spmv_other_values(Engine* h, DCsr& A, double*& other, const double* x, double* y);
int refresh_dinv(Engine* h, DCsr& A);
int build_cheb_patch(Engine* h, DCsr& A, int m);
int inner_prepare(Engine* h, int slot);
int launch_cheb_step(Engine* h, const DCsr& A, const double* dinv,
                            const double* b, const double* pm, const double* pk,
                            double* pn, double c0, double c1, double c2);
bool can_fuse_first(const Engine* h, const DCsr& A, const double* dinv);
int launch_cheb_first(Engine* h, const DCsr& A, const double* dinv,
                             const double* b, double* p0, double* pn, double s,
                             double c1, double c2);
int solve_cg_stream(Engine* h, const DCsr& A, Inner& s, const double* b,
                           double* x);
int solve_cg(Engine* h, const DCsr& A, Inner& s, const double* b,
                    double* x);
int solve_cg_sr(Engine* h, const DCsr& A, Inner& s, const double* b,
                       double* x);
int solve_cheb(Engine* h, const DCsr& A, Inner& s, const double* b,
                      double* x, double out_scale = 1.0);
int solve_rich(Engine* h, const DCsr& A, Inner& s, const double* b,
                      double* x);
int mg_smooth(Engine* h, const DCsr& A, double emin, double emax, int nu,
                     const double* b, double* bufs[3], bool zero_guess,
                     double** result);
int mg_vcycle(Engine* h, const DCsr& Afine, Inner& s, int l,
                     const double* b, double** out, double* target = nullptr);
int solve_mg(Engine* h, const DCsr& A, Inner& s, const double* b,
                    double* x);
int inner_solve(Engine* h, int slot, const double* b, double* x,
                       double out_scale = 1.0, bool* scaled = nullptr);
int apply_bc_dev(Engine* h, double* x);
int pcd_apply_dev(Engine* h, const double* x, double* y);
int fs_apply_eager(Engine* h, const double* x, double* y);
bool graph_capturable(const Engine* h);
int fs_apply_split(Engine* h, const double* x, double* y);
int io_begin(Engine* h, IoMap& io, const double* x, size_t nx, double* y,
                    size_t ny, int mem, bool y_in = false);
int io_end(IoMap& io);
int refresh_kron(Engine* h, DCsr& A);
bool kron_pattern(int nc, int64_t nrows, int64_t ncols, const int32_t* rowptr,
                         const int32_t* col);
int build_vec_tile(Engine* h, DCsr& A, int nc, int64_t nn, int64_t nloc,
                          const std::vector<int32_t>& rpc, const std::vector<int32_t>& cc);
int detect_kron(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                       const int32_t* rowptr, const int32_t* col, bool have_vals);
int detect_rowkron(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                          const int32_t* rowptr, const int32_t* col, bool have_vals);
int upload_csr(Engine* h, DCsr& A, int64_t nrows, int64_t ncols,
                      const int32_t* rowptr, const int32_t* col,
                      const double* val, const int64_t* src);
int ensure_space(Engine* h, Space& sp, int64_t n, bool velocity, const char* what);
int upload_global(Engine* h, DCsr& A, const Space* rs, const Space* cs,
                         int64_t nrows, int64_t ncols, const int32_t* rowptr,
                         const int32_t* col, const double* val,
                         const int64_t* src);
int upload_global_r(Engine* h, DCsr& A, const Space* rs, const Space* cs,
                           int64_t nrows, int64_t ncols, const int32_t* rowptr,
                           const int32_t* col, const double* val, const int64_t* src,
                           const Reorder* rr, const Reorder* rc);
int upload_perm(Engine* h, Reorder& r);
int refresh_values(Engine* h, DCsr& A, const double* vals, int mem);
int slice_in(Engine* h, const Space& sp, const double* full, double* loc);
int slice_out(Engine* h, const Space& sp, const double* loc, double* full);
void values_changed(Engine* h, int which);
int peer_check(Engine* h);
void extract_block(int64_t nr, const int32_t* rows, const int32_t* rowptr,
                          const int32_t* col, const std::vector<int32_t>& colmap,
                          std::vector<int32_t>& orp, std::vector<int32_t>& oc,
                          std::vector<int64_t>& osrc);
int gather_block_values(Engine* h, DCsr& A, const double* dvals);
int decide_reordering(Engine* h, int64_t n, const int32_t* rowptr, const int32_t* colidx,
                             int64_t n_u, const int32_t* is_u, int64_t n_p, const int32_t* is_p,
                             std::vector<int32_t>& isu_r, std::vector<int32_t>& isp_r);
int fio_begin(FieldIo& f, Engine* h, const Space* spx, int64_t nx_glob, int64_t nx_loc,
                     const Space* spy, int64_t ny_glob, int64_t ny_loc,
                     const double* x, double* y, int mem, bool y_in = false,
                     const Reorder* rx = nullptr, const Reorder* ry = nullptr);
int fio_end(FieldIo& f);
int dev_norm(Engine* h, int64_t n, const double* v, double* out);
int apply_system(Engine* h, const double* z, double* w);
void mat_spaces(Engine* h, int which, const Space** rs, const Space** cs);
int comm_attach(Engine* h, CommBackend* c, int rank, int nranks);
CommBackend* wrap_peer(Engine* h, CommBackend* boot, int rank, int nranks, ThreadGroup* tg,
                              bool default_on);
