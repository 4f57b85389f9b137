// pcd_kernels.hpp - gfx950 (CDNA4, wave64) device kernels of the PCD engine.
//
// Everything on this path is fp64 CSR SpMV + BLAS-1: HBM-bound (about 0.17
// flop/B), so no MFMA.  What matters (guides: coalescing, wave64 shuffles,
// launch count): every kernel streams its arrays once with unit-stride lanes,
// the vector updates of the Krylov recurrences are fused into the SpMV
// epilogue or into one element-wise kernel, and reductions never leave the
// device: a producer writes one partial per workgroup, every workgroup of the
// consumer re-reduces that short array in a FIXED order (bitwise reproducible,
// no fp64 atomics, no host sync, no extra launch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcd {

constexpr int kBlock = 256;       // 4 waves of 64
constexpr int kMaxParts = 1024;   // upper bound of per-workgroup partials

// ---- deterministic workgroup reductions ---------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// sum over the 256 threads; result valid in EVERY thread (fixed order)
__device__ __forceinline__ double block_sum(double v, double* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__device__ __forceinline__ double reduce_parts(const double* part, int n,
                                               double* sm) {
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += kBlock) v += part[i];
  return block_sum(v, sm);
}

// ---- CSR row sum: LPR lanes cooperate on one row -------------------------
// gathered vector: owned entries in x[0, nloc), ghost entries (filled by the
// halo exchange) in ghost[0, nghost); single GPU: nloc = ncols, no ghosts
struct XVec {
  const double* x; const double* ghost; int nloc;
  __device__ __forceinline__ double operator()(int c) const {
    return c < nloc ? x[c] : ghost[c - nloc];
  }
};

template <int LPR>
__device__ __forceinline__ double row_dot(const int* __restrict__ rowptr,
                                          const int* __restrict__ col,
                                          const double* __restrict__ val,
                                          const XVec x, int row, int nrows,
                                          int lane) {
  double s = 0.0;
  if (row < nrows) {
    const int b = rowptr[row], e = rowptr[row + 1];
    for (int k = b + lane; k < e; k += LPR) s += val[k] * x(col[k]);
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LPR);
  return s;
}

// MODE 0: y = A x   MODE 1: y = add + A x   MODE 2: y = add - A x   MODE 3: y = -A x
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void k_spmv(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XVec x, const double* add,
    double* y) {
  constexpr int RPB = kBlock / LPR;
  const int lane = threadIdx.x % LPR;
  const int nloop = (nrows + RPB - 1) / RPB * RPB;
  for (int row = blockIdx.x * RPB + threadIdx.x / LPR; row < nloop;
       row += gridDim.x * RPB) {
    double s = row_dot<LPR>(rowptr, col, val, x, row, nrows, lane);
    if (lane == 0 && row < nrows) {
      if (MODE == 0) y[row] = s;
      if (MODE == 1) y[row] = add[row] + s;
      if (MODE == 2) y[row] = add[row] - s;
      if (MODE == 3) y[row] = -s;
    }
  }
}

// One Chebyshev / Richardson step, fused (SpMV + residual + Jacobi + 3-term
// update):  pn = c0*pm + c1*pk + c2 * dinv .* (b - A pk)
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_cheb_step(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* __restrict__ dinv,
    const double* b, const double* pm, const double* pk, double* pn,
    double c0, double c1, double c2, const double* ghost, int nloc) {
  constexpr int RPB = kBlock / LPR;
  const int lane = threadIdx.x % LPR;
  const int nloop = (nrows + RPB - 1) / RPB * RPB;
  const XVec xv{pk, ghost, nloc};
  for (int row = blockIdx.x * RPB + threadIdx.x / LPR; row < nloop;
       row += gridDim.x * RPB) {
    double s = row_dot<LPR>(rowptr, col, val, xv, row, nrows, lane);
    if (lane == 0 && row < nrows) {
      double z = b[row] - s;
      if (dinv) z *= dinv[row];
      double out = c1 * pk[row] + c2 * z;
      if (c0 != 0.0) out += c0 * pm[row];
      pn[row] = out;
    }
  }
}

// x = s * dinv .* b   (Chebyshev / Richardson / preonly start, zero guess)
static __global__ __launch_bounds__(kBlock) void k_scale_dinv(
    int n, const double* __restrict__ dinv, const double* b, double s,
    double* x) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    x[i] = s * (dinv ? dinv[i] : 1.0) * b[i];
}

// ---- Jacobi-PCG (natural norm), state kept on the device ------------------
struct CgState {
  double rz0;
  int its;
  int done;
  double betaold, dpiold;   // single-reduction variant only
};

// x = 0, r = b, z = dinv r, p = z; parts_rz[blk] = sum r.z
static __global__ __launch_bounds__(kBlock) void k_cg_init(
    int n, const double* __restrict__ dinv, const double* b, double* x,
    double* r, double* z, double* p, double* parts_rz, CgState* st) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock) {
    const double ri = b[i];
    const double zi = dinv ? dinv[i] * ri : ri;
    x[i] = 0.0; r[i] = ri; z[i] = zi; p[i] = zi;
    acc += ri * zi;
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) {
    parts_rz[blockIdx.x] = acc;
    if (blockIdx.x == 0) { st->its = 0; st->done = 0; st->rz0 = 0.0; }
  }
}

// p = z + (rz_new / rz_old) p; detects convergence (every workgroup takes the
// same decision from the same partials; workgroup 0 publishes it)
static __global__ __launch_bounds__(kBlock) void k_cg_pupdate(
    int n, const double* z, double* p, const double* parts_new,
    const double* parts_old, int nparts, double rtol, CgState* st) {
  __shared__ double sm[4];
  if (st->done) return;
  const double rz_new = reduce_parts(parts_new, nparts, sm);
  const double rz_old = reduce_parts(parts_old, nparts, sm);
  const double rz0 = st->rz0;
  const bool conv = (rz_new == 0.0) ||
      (rtol > 0.0 && sqrt(fabs(rz_new)) <= rtol * sqrt(fabs(rz0)));
  if (conv) {
    // other workgroups may still be reading st->done == 0: that is fine, they
    // reach the same verdict on their own; later kernels see the store
    if (blockIdx.x == 0 && threadIdx.x == 0) st->done = 1;
    return;
  }
  const double beta = rz_new / rz_old;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    p[i] = z[i] + beta * p[i];
}

// q = A p; parts_pq[blk] = sum p.q
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_cg_spmv_dot(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* p, double* q,
    double* parts_pq, const CgState* st, const double* ghost, int nloc) {
  __shared__ double sm[4];
  if (st->done) return;
  constexpr int RPB = kBlock / LPR;
  const int lane = threadIdx.x % LPR;
  const int nloop = (nrows + RPB - 1) / RPB * RPB;
  const XVec xv{p, ghost, nloc};
  double acc = 0.0;
  for (int row = blockIdx.x * RPB + threadIdx.x / LPR; row < nloop;
       row += gridDim.x * RPB) {
    double s = row_dot<LPR>(rowptr, col, val, xv, row, nrows, lane);
    if (lane == 0 && row < nrows) { q[row] = s; acc += p[row] * s; }
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) parts_pq[blockIdx.x] = acc;
}

// alpha = rz / pq; x += alpha p; r -= alpha q; z = dinv r; parts_out = r.z
static __global__ __launch_bounds__(kBlock) void k_cg_update(
    int n, const double* __restrict__ dinv, const double* p, const double* q,
    double* x, double* r, double* z, const double* parts_rz, int nparts_rz,
    const double* parts_pq, int nparts_pq, double* parts_out, int it,
    CgState* st) {
  __shared__ double sm[4];
  if (st->done) return;
  const double rz = reduce_parts(parts_rz, nparts_rz, sm);
  const double pq = reduce_parts(parts_pq, nparts_pq, sm);
  const double alpha = (pq != 0.0) ? rz / pq : 0.0;
  double acc = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    const double zi = dinv ? dinv[i] * ri : ri;
    r[i] = ri; z[i] = zi;
    acc += ri * zi;
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) {
    parts_out[blockIdx.x] = acc;
    if (blockIdx.x == 0) {
      if (it == 0) st->rz0 = rz;
      st->its = it + 1;
    }
  }
}

// ---- CG with a single reduction per iteration ------------------------------
// [ext PETSc] -ksp_cg_single_reduction (Chronopoulos-Gear): beta = (z, r) and
// delta = (z, A z) are formed by ONE kernel, so several ranks need one
// all-reduce of two numbers per iteration instead of two of one; p.Ap follows
// from dpi = delta - beta^2 dpi_old / beta_old^2.  State ping-pongs between
// st_in / st_out so that every workgroup reads the same snapshot.
static __global__ __launch_bounds__(kBlock) void k_cgsr_init(
    int n, const double* __restrict__ dinv, const double* b, double* x,
    double* r, double* z, CgState* st) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double ri = b[i];
    x[i] = 0.0; r[i] = ri; z[i] = dinv ? dinv[i] * ri : ri;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st[0].rz0 = 0.0; st[0].its = 0; st[0].done = 0; st[0].betaold = 0.0; st[0].dpiold = 0.0;
    st[1] = st[0];
  }
}

// s = A z; parts_b[blk] = sum z.r, parts_d[blk] = sum z.s
template <int LPR>
__global__ __launch_bounds__(kBlock) void k_cgsr_spmv_dots(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* z, const double* r, double* sv,
    double* parts_b, double* parts_d, const CgState* st, const double* ghost,
    int nloc) {
  __shared__ double sm[4];
  if (st->done) return;
  constexpr int RPB = kBlock / LPR;
  const int lane = threadIdx.x % LPR;
  const int nloop = (nrows + RPB - 1) / RPB * RPB;
  const XVec xv{z, ghost, nloc};
  double ab = 0.0, ad = 0.0;
  for (int row = blockIdx.x * RPB + threadIdx.x / LPR; row < nloop;
       row += gridDim.x * RPB) {
    const double s = row_dot<LPR>(rowptr, col, val, xv, row, nrows, lane);
    if (lane == 0 && row < nrows) {
      const double zi = z[row];
      sv[row] = s; ab += zi * r[row]; ad += zi * s;
    }
  }
  ab = block_sum(ab, sm);
  ad = block_sum(ad, sm);
  if (threadIdx.x == 0) { parts_b[blockIdx.x] = ab; parts_d[blockIdx.x] = ad; }
}

// the iteration's scalars from the (reduced) dots, then all vector updates:
// p = z + b p, w = s + b w, x += a p, r -= a w, z = dinv r
static __global__ __launch_bounds__(kBlock) void k_cgsr_update(
    int n, const double* __restrict__ dinv, double* z, const double* sv,
    double* p, double* w, double* x, double* r, const double* pb, int nb,
    const double* pd, int nd, double rtol, int it, const CgState* st_in,
    CgState* st_out) {
  __shared__ double sm[4];
  if (st_in->done) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *st_out = *st_in;
    return;
  }
  const double beta = reduce_parts(pb, nb, sm);
  const double delta = reduce_parts(pd, nd, sm);
  const double beta0 = it == 0 ? beta : st_in->rz0;
  const bool conv = (beta == 0.0) ||
      (rtol > 0.0 && sqrt(fabs(beta)) <= rtol * sqrt(fabs(beta0)));
  if (conv) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *st_out = *st_in; st_out->rz0 = beta0; st_out->its = it; st_out->done = 1;
    }
    return;
  }
  double bb = 0.0, dpi = delta;
  if (it > 0) {
    const double bo = st_in->betaold;
    bb = beta / bo;
    dpi = delta - beta * beta * st_in->dpiold / (bo * bo);
  }
  // indefinite / singular operator or cancellation in the recurrence: p.Ap <= 0
  // (PETSc: KSP_DIVERGED_INDEFINITE_MAT).  Stop with the iterate reached so
  // far instead of writing Inf / NaN; its = -(it + 1) tells the host.
  if (!(dpi > 0.0) || !isfinite(dpi)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *st_out = *st_in; st_out->rz0 = beta0; st_out->its = -(it + 1); st_out->done = 1;
    }
    return;
  }
  const double a = beta / dpi;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double pi = it > 0 ? z[i] + bb * p[i] : z[i];
    const double wi = it > 0 ? sv[i] + bb * w[i] : sv[i];
    p[i] = pi; w[i] = wi;
    x[i] += a * pi;
    const double ri = r[i] - a * wi;
    r[i] = ri;
    z[i] = dinv ? dinv[i] * ri : ri;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st_out->rz0 = beta0; st_out->its = it + 1; st_out->done = 0;
    st_out->betaold = beta; st_out->dpiold = dpi;
  }
}

// ==========================================================================
// CSR-stream kernels (the fast path).  One workgroup owns RB consecutive rows.
// Phase 1 streams that row block's val/col arrays with unit-stride lanes
// (fully coalesced, several independent loads in flight per lane), multiplies
// by the gathered vector entry and parks the products in LDS.  Phase 2: the
// 256/RB lanes of a row sum its LDS segment (strided, then a xor-butterfly:
// a fixed order, so results are reproducible; RB = 256 is the scalar CPU
// order).  Phase 3: the fused epilogue runs on the first lane of each row, so
// the vector operands (b, p_k, p_{k-1}, diag) stay coalesced as well.
// Workgroups are mapped to row blocks XCD-aware: workgroups b, b+8, b+16, ...
// share an XCD (round-robin dispatch), so they are given CONTIGUOUS row-block
// ranges - each XCD's private L2 then holds one slab of the gathered vector
// instead of all of it.
// ==========================================================================
constexpr int kTile = 4096;       // LDS doubles per workgroup (32 KiB)

// p = z + beta * p_old formed on the fly (CG direction update fused in SpMV)
struct XCg {
  const double* z; const double* p; double beta; bool first;
  __device__ __forceinline__ double operator()(int c) const {
    return first ? z[c] : z[c] + beta * p[c];
  }
};

// XCD-aware mapping for operators of up to this many (block) rows, i.e. while
// matrix + vectors stay cache-resident between launches; larger operators
// stream from HBM and do better with all XCDs walking one front (DESIGN.md
// section 4 has the A/B).  PCD_NO_XCD_REMAP=1 sets it to 0.
static __constant__ int g_xcd_remap_max_rows = 1 << 20;
// ... and at EVERY size by the two-component kernels that stream the matrix
// non-temporally (bit 0: two-component kernels, bit 1: three-component ones;
// PCD_XCD_REMAP_NT).  Measured: cavity level 7 70.6 -> 68.8 us and the counted
// traffic 1.21 -> 0.99 x the kernel-model bytes; cube N = 64 the traffic falls
// the same way (1.21 -> 1.02 x) but the launch gets 2 % SLOWER (262 -> 267 us:
// the 3-D kernel is bound by its gather instructions, not by bytes), so bit 1
// stays off (profiles/r03_x_xcd_nt_*.txt) ...
static __constant__ int g_xcd_remap_nt = 1;
// ... below this many node rows.  Cube N = 73 (3.18 M node rows; config 5's own
// mesh): counted traffic 1.32 x the kernel-model bytes, and the mapping wins:
// 395.1 -> 376.5 us per launch, 2.913 -> 2.848 ms per PCApply
// (profiles/r03_x_xcd_nt_cube73.txt; PCD_XCD_REMAP_NT3_ROWS).
static __constant__ int g_xcd_remap_nt3_rows = 2600000;

template <int NC, bool NT>
__device__ __forceinline__ bool xcd_remap_always(int nrows) {
  return NT && (((g_xcd_remap_nt >> (NC - 2)) & 1) || (NC == 3 && nrows >= g_xcd_remap_nt3_rows));
}

// contiguous range of row blocks of this workgroup (gridDim.x multiple of 8)
__device__ __forceinline__ void row_block_range(int nrb, int rb_rows, int& begin, int& end,
                                                bool always = false) {
  const int G = gridDim.x;
  // (`always`: kernels that stream the matrix non-temporally keep the mapping at
  // every size - with the matrix out of the caches' way the gathered vector's
  // slab is what the XCD's L2 holds: level 7 70.6 -> 68.8 us,
  // profiles/r03_x_xcd_nt_level7.txt; with default-policy loads it costs 4 %)
  const bool remap = always || (long long)nrb * rb_rows <= g_xcd_remap_max_rows;
  const int slot = remap ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8
                         : (int)blockIdx.x;
  begin = (int)((long long)slot * nrb / G);
  end = (int)((long long)(slot + 1) * nrb / G);
}

// Loads of one pass are all issued before any use: kUnroll independent
// (col, val) pairs per lane, then kUnroll independent gathers - one dependent
// chain rowptr -> col/val -> x per row block instead of one per loop trip.
#ifndef PCD_UNROLL
#define PCD_UNROLL 8
#endif
constexpr int kUnroll = PCD_UNROLL;

// The matrix arrays are read exactly once per launch; the gathered vector is
// what should stay in cache.  Operators whose launch moves more than the
// 256 MiB Infinity Cache holds stream val/col NON-TEMPORALLY (template flag NT
// of the multi-component kernels, chosen per operator by the host): measured
// on the finest A00 of cavity level 7 (439 MB per launch) 75.9 -> 69.5 us; on
// level 6 (91 MB, cache-resident between launches) the same hint costs 12 %
// (profiles/r03_g_pipelined_kernels_negative_result.txt, lib = -DPCD_NT_LOADS).
// products of three-component operators parked in LDS as three planes of
// doubles instead of 24-byte records (A/B switch)
#ifndef PCD_LDS_SOA
#define PCD_LDS_SOA 1
#endif
#define PCD_STREAM_LOAD(p) (*(p))
template <bool NT, class T>
__device__ __forceinline__ T stream_load(const T* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <int RB, class XF, int TILE = kTile>
__device__ __forceinline__ double stream_row_block(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XF& xf, int r0, int nrows,
    double* lds) {
  const int r1 = min(r0 + RB, nrows);
  constexpr int TPR = kBlock / RB;          // lanes that share one row's sum
  const int row = r0 + threadIdx.x / TPR;
  const int sub = threadIdx.x % TPR;
  const bool mine = row < r1;
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  // this lane's own row bounds, fetched up front (used after the barrier)
  const int ra = mine ? rowptr[row] - k0 : 0;
  const int rb = mine ? rowptr[row + 1] - k0 : 0;
  double s = 0.0;
  // the block's entries pass through the LDS tile in chunks (normally one;
  // the host admits a few more for operators with long rows)
  for (int c0 = 0; c0 < k1 - k0; c0 += TILE) {
    const int c1 = min(c0 + TILE, k1 - k0);
    if (c0) __syncthreads();                // readers of the previous chunk
    for (int base = k0 + c0; base < k0 + c1; base += kUnroll * kBlock) {
      int c[kUnroll];
      double v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int k = base + u * kBlock + threadIdx.x;
        const bool in = k < k0 + c1;
        c[u] = in ? PCD_STREAM_LOAD(col + k) : -1;
        v[u] = in ? PCD_STREAM_LOAD(val + k) : 0.0;
      }
      double xv[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) xv[u] = c[u] >= 0 ? xf(c[u]) : 0.0;
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int k = base + u * kBlock + threadIdx.x;
        if (k < k0 + c1) lds[k - k0 - c0] = v[u] * xv[u];
      }
    }
    __syncthreads();
    const int lo = max(ra, c0), hi = min(rb, c1);
    int j = ra + sub;                       // lane `sub` owns j = ra+sub (mod TPR)
    if (j < lo) j += (lo - j + TPR - 1) / TPR * TPR;
    for (; j < hi; j += TPR) s += lds[j - c0];
  }
#pragma unroll
  for (int m = TPR / 2; m > 0; m >>= 1) s += __shfl_xor(s, m);
  return s;                                  // on every lane of the row
}

// TILE: LDS doubles per workgroup.  Operators with short rows (A01: 4.75
// entries per row; 256 rows fill 1 200 of the 4 096 slots) take the half
// tile: 16 KiB leave room for 8 resident workgroups per CU instead of 5.
constexpr int kTileSmall = 2048;
template <int RB, int MODE, int TILE = kTile>
__global__ __launch_bounds__(kBlock) void k_spmv_s(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XVec xf, const double* add,
    double* y) {
  __shared__ double lds[TILE];
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1);
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    double a = 0.0;
    if ((MODE == 1 || MODE == 2) && mine) a = add[row];   // early: hides under phase 1
    const double s = stream_row_block<RB, XVec, TILE>(rowptr, col, val, xf, r0, nrows, lds);
    if (mine) {
      if (MODE == 0) y[row] = s;
      if (MODE == 1) y[row] = a + s;
      if (MODE == 2) y[row] = a - s;
      if (MODE == 3) y[row] = -s;
    }
    __syncthreads();
  }
}

template <int RB>
__global__ __launch_bounds__(kBlock) void k_cheb_step_s(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* __restrict__ dinv,
    const double* b, const double* pm, const double* pk, double* pn,
    double c0, double c1, double c2, const double* ghost, int nloc) {
  __shared__ double lds[kTile];
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1);
  const XVec xf{pk, ghost, nloc};
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    // epilogue operands first: their latency hides under the streaming phase
    double bi = 0.0, d = 1.0, xk = 0.0, xm = 0.0;
    if (mine) {
      bi = b[row]; xk = pk[row];
      if (dinv) d = dinv[row];
      if (c0 != 0.0) xm = pm[row];
    }
    const double s = stream_row_block<RB>(rowptr, col, val, xf, r0, nrows, lds);
    if (mine) pn[row] = c0 * xm + c1 * xk + c2 * (d * (bi - s));
    __syncthreads();
  }
}

// First smoothing step from a ZERO guess, fused with the Jacobi start:
//   p0 = s * dinv .* b  (written once for the owned rows);
//   pn = c1 p0 + c2 dinv .* (b - A p0)
// A p0 = s * (A D^-1) b is taken from the COLUMN-SCALED values `vals`
// (val[k] * dinv[col[k]], refreshed with the diagonal by k_scale_cols), so the
// gather reads b alone - one gathered stream, like every other step (forming
// p0 on the fly for the gathered columns cost two: 27 vs 20 us on the finest
// velocity level).
template <int RB>
__global__ __launch_bounds__(kBlock) void k_cheb_first_s(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ vals, const double* __restrict__ dinv,
    const double* b, double* p0, double* pn, double s, double c1, double c2,
    const double* ghost, int nloc) {
  __shared__ double lds[kTile];
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1);
  // (one GPU: ghost == b, nloc == nrows - the ghost segment is never read;
  // several ranks: the halo of b, exchanged before the launch)
  const XVec xf{b, ghost, nloc};
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    double d = 0.0, bi = 0.0;
    if (mine) { d = dinv[row]; bi = b[row]; }     // early: hides under phase 1
    const double sum = s * stream_row_block<RB>(rowptr, col, vals, xf, r0, nrows, lds);
    if (mine) {
      const double x0 = s * d * bi;
      if (p0) p0[row] = x0;
      pn[row] = c1 * x0 + c2 * d * (bi - sum);
    }
    __syncthreads();
  }
}

// vals[k] = val[k] * dinv[col[k] * stride]   (stride = components per node
// when `val` holds the scalar stencil F of an F (x) I operator)
// (`ghost` / `nloc`: several ranks - the reciprocal diagonal of the ghost
// columns, exchanged like any halo; one GPU: nloc = every column)
static __global__ __launch_bounds__(kBlock) void k_scale_cols(
    int64_t nnz, const int* __restrict__ col, const double* __restrict__ val,
    const double* __restrict__ dinv, int stride, double* vals,
    const double* __restrict__ ghost, int nloc) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock) {
    const int c = col[k];
    vals[k] = val[k] * (c < nloc ? dinv[(int64_t)c * stride]
                                 : ghost[(int64_t)(c - nloc) * stride]);
  }
}

// long rows (dense coarse inverse): one workgroup per row, 4 loads in flight
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_spmv_long(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XVec x, const double* add,
    double* y) {
  __shared__ double sm[4];
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = rowptr[row], e = rowptr[row + 1];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = b + threadIdx.x;
    for (; k + 3 * kBlock < e; k += 4 * kBlock) {
      s0 += val[k] * x(col[k]);
      s1 += val[k + kBlock] * x(col[k + kBlock]);
      s2 += val[k + 2 * kBlock] * x(col[k + 2 * kBlock]);
      s3 += val[k + 3 * kBlock] * x(col[k + 3 * kBlock]);
    }
    for (; k < e; k += kBlock) s0 += val[k] * x(col[k]);
    const double s = block_sum((s0 + s1) + (s2 + s3), sm);
    if (threadIdx.x == 0) {
      if (MODE == 0) y[row] = s;
      if (MODE == 1) y[row] = add[row] + s;
      if (MODE == 2) y[row] = add[row] - s;
      if (MODE == 3) y[row] = -s;
    }
  }
}

// CG, first kernel of an iteration: direction update fused into the SpMV.
//   beta = rz_new / rz_old (both re-reduced from workgroup partials);
//   p_new = z + beta p_old;  q = A p_new;  parts_pq[wg] = sum p_new . q
template <int RB>
__global__ __launch_bounds__(kBlock) void k_cg_spmv_s(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* z, const double* p_old,
    double* p_new, double* q, const double* parts_new, const double* parts_old,
    int nparts, double rtol, int first, double* parts_pq, CgState* st) {
  __shared__ double lds[kTile];
  __shared__ double sm[4];
  if (st->done) return;
  double beta = 0.0;
  if (!first) {
    const double rz_new = reduce_parts(parts_new, nparts, sm);
    const double rz_old = reduce_parts(parts_old, nparts, sm);
    const double rz0 = st->rz0;
    const bool conv = (rz_new == 0.0) ||
        (rtol > 0.0 && sqrt(fabs(rz_new)) <= rtol * sqrt(fabs(rz0)));
    if (conv) {
      if (blockIdx.x == 0 && threadIdx.x == 0) st->done = 1;
      return;
    }
    beta = rz_new / rz_old;
  }
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1);
  const XCg xf{z, p_old, beta, first != 0};
  double acc = 0.0;
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const double s = stream_row_block<RB>(rowptr, col, val, xf, r0, nrows, lds);
    const int row = r0 + threadIdx.x / (kBlock / RB);
    if (threadIdx.x % (kBlock / RB) == 0 && row < nrows) {
      const double pn = xf(row);
      p_new[row] = pn;
      q[row] = s;
      acc += pn * s;
    }
    __syncthreads();
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) parts_pq[blockIdx.x] = acc;
}

// ==========================================================================
// Multi-component (Kronecker) operators: A = F (x) I_NC on interleaved dofs
// (NC*node + component), NC = 2 (plane) or 3 (space).  Picard velocity blocks,
// their Galerkin coarse levels and the velocity prolongations have this
// structure: every component of a node sees the same scalar stencil.  The
// engine then streams F ONCE (1/NC of the matrix bytes) and carries all
// components of a node together: 16/24-byte gathers and coalesced vector
// traffic.  Same three phases as the stream kernels.
// ==========================================================================
#ifndef PCD_TILE3
// 1536 triples = 36 KiB: four workgroups per CU instead of three; a 64-row
// block of a 3-D P2 operator (29 entries per row) then takes two passes.
// Measured on the cube N = 32 (profiles/r02_k_tile_sweep3d.txt): 39.9 us per
// launch at 2048, 35.0 at 1536-1664, 38-44 at 1280-1408, 39.4 at 1792.
#define PCD_TILE3 1536
#endif
#ifndef PCD_TILE2
#define PCD_TILE2 2048
#endif
// LDS nodes per workgroup: PCD_TILE2 pairs x 16 B (2048 = 32 KiB); triples:
// PCD_TILE3 x 24 B (compile-time A/B switches, tools/tile_sweep.sh)
template <int NC> constexpr int tile_c() { return NC == 3 ? PCD_TILE3 : PCD_TILE2; }

template <int NC>
struct alignas(NC == 2 ? 16 : 8) VecC {
  double c[NC];
};
template <int NC>
__device__ __forceinline__ VecC<NC> vzero() {
  VecC<NC> r;
#pragma unroll
  for (int i = 0; i < NC; ++i) r.c[i] = 0.0;
  return r;
}
template <int NC>
__device__ __forceinline__ const VecC<NC>* vc(const double* p) {
  return reinterpret_cast<const VecC<NC>*>(p);
}
template <int NC>
__device__ __forceinline__ VecC<NC>* vc(double* p) {
  return reinterpret_cast<VecC<NC>*>(p);
}

template <int NC>
struct XVecC {
  const VecC<NC>* x; const VecC<NC>* ghost; int nloc;      // in node units
  // (24-byte triples as one 16-byte + one 8-byte load instead of three 8-byte
  // ones: no difference, 35.0 vs 35.1 us on the cube N = 32 -
  // profiles/r03_t_triple_gather_ab_negative_result.txt)
  __device__ __forceinline__ VecC<NC> operator()(int c) const {
    return c < nloc ? x[c] : ghost[c - nloc];
  }
};
template <int NC>
struct XScaledC {
  const VecC<NC>* b; const VecC<NC>* dinv; double s;
  __device__ __forceinline__ VecC<NC> operator()(int c) const {
    const VecC<NC> d = dinv[c], v = b[c];
    VecC<NC> r;
#pragma unroll
    for (int i = 0; i < NC; ++i) r.c[i] = s * d.c[i] * v.c[i];
    return r;
  }
};

template <int RB, int NC, bool NT, class XF>
__device__ __forceinline__ VecC<NC> stream_row_block_c(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XF& xf, int r0, int nrows,
    VecC<NC>* lds) {
  const int r1 = min(r0 + RB, nrows);
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  constexpr int TPR = kBlock / RB;          // lanes that share one row's sum
  const int row = r0 + threadIdx.x / TPR;
  const int sub = threadIdx.x % TPR;
  const bool mine = row < r1;
  const int ra = mine ? rowptr[row] - k0 : 0;
  const int rb = mine ? rowptr[row + 1] - k0 : 0;
  VecC<NC> s = vzero<NC>();
  constexpr int kTileC = tile_c<NC>();
  double* planes = reinterpret_cast<double*>(lds);   // NC planes of kTileC doubles
  for (int c0 = 0; c0 < k1 - k0; c0 += kTileC) {   // chunks of the LDS tile
    const int c1 = min(c0 + kTileC, k1 - k0);
    if (c0) __syncthreads();
    for (int base = k0 + c0; base < k0 + c1; base += kUnroll * kBlock) {
      int c[kUnroll];
      double v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int k = base + u * kBlock + threadIdx.x;
        const bool in = k < k0 + c1;
        c[u] = in ? stream_load<NT>(col + k) : -1;
        v[u] = in ? stream_load<NT>(val + k) : 0.0;
      }
      VecC<NC> xv[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u)
        xv[u] = c[u] >= 0 ? xf(c[u]) : vzero<NC>();
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int k = base + u * kBlock + threadIdx.x;
        if (k < k0 + c1) {
          if (PCD_LDS_SOA && NC == 3) {
            // 24-byte records put consecutive lanes 6 banks apart (2-way
            // conflicts on every access); three planes of doubles do not
#pragma unroll
            for (int i = 0; i < NC; ++i)
              planes[i * kTileC + (k - k0 - c0)] = v[u] * xv[u].c[i];
          } else {
            VecC<NC> t;
#pragma unroll
            for (int i = 0; i < NC; ++i) t.c[i] = v[u] * xv[u].c[i];
            lds[k - k0 - c0] = t;
          }
        }
      }
    }
    __syncthreads();
    const int lo = max(ra, c0), hi = min(rb, c1);
    int j = ra + sub;
    if (j < lo) j += (lo - j + TPR - 1) / TPR * TPR;
    for (; j < hi; j += TPR) {
      if (PCD_LDS_SOA && NC == 3) {
#pragma unroll
        for (int i = 0; i < NC; ++i) s.c[i] += planes[i * kTileC + (j - c0)];
      } else {
        const VecC<NC> t = lds[j - c0];
#pragma unroll
        for (int i = 0; i < NC; ++i) s.c[i] += t.c[i];
      }
    }
  }
#pragma unroll
  for (int m = TPR / 2; m > 0; m >>= 1) {
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] += __shfl_xor(s.c[i], m);
  }
  return s;                                  // on every lane of the row
}

// vectors arrive as plain double* (node-interleaved) and are viewed as VecC
template <int RB, int MODE, int NC, bool NT = false>
__global__ __launch_bounds__(kBlock) void k_spmv_sc(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x, const double* ghost,
    int nloc, const double* add_, double* y_) {
  __shared__ VecC<NC> lds[tile_c<NC>()];
  const XVecC<NC> xf{vc<NC>(x), vc<NC>(ghost), nloc};
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1, xcd_remap_always<NC, NT>(nrows));
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    VecC<NC> a = vzero<NC>();
    if ((MODE == 1 || MODE == 2) && mine) a = add[row];   // early: hides under phase 1
    const VecC<NC> s = stream_row_block_c<RB, NC, NT>(rowptr, col, val, xf, r0, nrows, lds);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
    __syncthreads();
  }
}

// ---- ROW-blocked operators: the NC rows of a node share one column pattern
// (the discrete gradient A01 = B^T: rows = velocity components of a node,
// columns = pressure dofs; [ext PETSc] MatMult(A01) of the fieldsplit apply,
// field_split.py:54-57).  A scalar CSR kernel reads the column index once per
// component - 36 B per node-entry in space, 24 in the plane; here the set-up
// keeps the pattern once per NODE and the values node-entry-major
// (val[k][NC]): 4 + 8 NC B per node-entry (28 / 20), one scalar gather of x
// per node-entry instead of NC.  Same three phases as the stream kernels.
template <int RB, int NC, bool NT>
__device__ __forceinline__ VecC<NC> stream_row_block_rk(
    const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XVec& xf, int r0, int nrows,
    VecC<NC>* lds) {
  const int r1 = min(r0 + RB, nrows);
  const int k0 = rowptr[r0], k1 = rowptr[r1];
  constexpr int TPR = kBlock / RB;
  const int row = r0 + threadIdx.x / TPR;
  const int sub = threadIdx.x % TPR;
  const bool mine = row < r1;
  const int ra = mine ? rowptr[row] - k0 : 0;
  const int rb = mine ? rowptr[row + 1] - k0 : 0;
  VecC<NC> s = vzero<NC>();
  constexpr int kTileC = tile_c<NC>();
  constexpr int U = kUnroll / 2;            // (NC values per entry in flight)
  double* planes = reinterpret_cast<double*>(lds);
  for (int c0 = 0; c0 < k1 - k0; c0 += kTileC) {
    const int c1 = min(c0 + kTileC, k1 - k0);
    if (c0) __syncthreads();
    for (int base = k0 + c0; base < k0 + c1; base += U * kBlock) {
      int c[U];
      double v[U][NC];
      double xv[U];
      if constexpr (NT) {
        // The NC values of an entry sit side by side (val[k][NC]): a lane that
        // loads "its" entry's values reads NC doubles at a stride of 8 NC bytes,
        // and every one of the NC load instructions of a wave touches all the
        // cache lines of the wave's 64 entries.  Here the wave loads its 64 NC
        // doubles CONTIGUOUSLY - double l + 64 j of the wave's stretch on lane
        // l - and fetches the gathered x of the entry that double belongs to,
        // (l + 64 j) / NC, from the lane that gathered it (a wave shuffle).  Same
        // products, same places in LDS.  Operators streamed from HBM (NT) only:
        // same-box A/B on the discrete gradient, cube N = 73: 230.7 -> 198.6 us;
        // cube N = 48 (resident in the Infinity Cache, NT off) 59.8 -> 63.5 us
        // (profiles/r05_o_*).
        const int lane = threadIdx.x & 63, wv0 = threadIdx.x & ~63;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = base + u * kBlock + threadIdx.x;
          c[u] = k < k0 + c1 ? stream_load<NT>(col + k) : -1;
          const size_t f0 = (size_t)NC * (base + u * kBlock + wv0) + lane;
#pragma unroll
          for (int j = 0; j < NC; ++j)
            v[u][j] = f0 + 64 * j < (size_t)NC * (k0 + c1) ? stream_load<NT>(val + f0 + 64 * j) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = c[u] >= 0 ? xf(c[u]) : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const int fl = lane + 64 * j, e = fl / NC, i = fl - e * NC;
            const double xe = __shfl(xv[u], e);
            const int k = base + u * kBlock + wv0 + e;
            if (k < k0 + c1) planes[i * kTileC + (k - k0 - c0)] = v[u][j] * xe;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = base + u * kBlock + threadIdx.x;
          const bool in = k < k0 + c1;
          c[u] = in ? stream_load<NT>(col + k) : -1;
#pragma unroll
          for (int i = 0; i < NC; ++i)
            v[u][i] = in ? stream_load<NT>(val + (size_t)NC * k + i) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = c[u] >= 0 ? xf(c[u]) : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = base + u * kBlock + threadIdx.x;
          if (k < k0 + c1) {
#pragma unroll
            for (int i = 0; i < NC; ++i)
              planes[i * kTileC + (k - k0 - c0)] = v[u][i] * xv[u];
          }
        }
      }
    }
    __syncthreads();
    const int lo = max(ra, c0), hi = min(rb, c1);
    int j = ra + sub;
    if (j < lo) j += (lo - j + TPR - 1) / TPR * TPR;
    for (; j < hi; j += TPR) {
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += planes[i * kTileC + (j - c0)];
    }
  }
#pragma unroll
  for (int m = TPR / 2; m > 0; m >>= 1) {
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] += __shfl_xor(s.c[i], m);
  }
  return s;
}

template <int RB, int MODE, int NC, bool NT = false>
__global__ __launch_bounds__(kBlock) void k_spmv_rk(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, XVec xf, const double* add_, double* y_) {
  __shared__ VecC<NC> lds[tile_c<NC>()];
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1, xcd_remap_always<NC, NT>(nrows));
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    VecC<NC> a = vzero<NC>();
    if ((MODE == 1 || MODE == 2) && mine) a = add[row];
    const VecC<NC> s = stream_row_block_rk<RB, NC, NT>(rowptr, col, val, xf, r0, nrows, lds);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
    __syncthreads();
  }
}

template <int RB, int NC, bool NT = false>
__global__ __launch_bounds__(kBlock) void k_cheb_step_sc(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* __restrict__ dinv_,
    const double* b_, const double* pm_, const double* pk_, double* pn_,
    double c0, double c1, double c2, const double* ghost, int nloc) {
  __shared__ VecC<NC> lds[tile_c<NC>()];
  const VecC<NC>*dinv = vc<NC>(dinv_), *b = vc<NC>(b_), *pm = vc<NC>(pm_),
               *pk = vc<NC>(pk_);
  VecC<NC>* pn = vc<NC>(pn_);
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1, xcd_remap_always<NC, NT>(nrows));
  const XVecC<NC> xf{pk, vc<NC>(ghost), nloc};
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    // the epilogue operands do not depend on the row sums: issue their loads
    // first so that their latency hides under the streaming phase
    VecC<NC> bi = vzero<NC>(), d = bi, xk = bi, xm = bi;
    if (mine) {
      bi = b[row]; d = dinv[row]; xk = pk[row];
      if (c0 != 0.0) xm = pm[row];
    }
    const VecC<NC> s = stream_row_block_c<RB, NC, NT>(rowptr, col, val, xf, r0, nrows, lds);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = c0 * xm.c[i] + c1 * xk.c[i] + c2 * d.c[i] * (bi.c[i] - s.c[i]);
      pn[row] = o;
    }
    __syncthreads();
  }
}

template <int RB, int NC, bool NT = false>
__global__ __launch_bounds__(kBlock) void k_cheb_first_sc(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ vals, const double* __restrict__ dinv_,
    const double* b_, double* p0_, double* pn_, double s, double c1, double c2,
    const double* ghost, int nloc) {
  __shared__ VecC<NC> lds[tile_c<NC>()];
  const VecC<NC>*dinv = vc<NC>(dinv_), *b = vc<NC>(b_);
  VecC<NC>*p0 = vc<NC>(p0_), *pn = vc<NC>(pn_);
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1, xcd_remap_always<NC, NT>(nrows));
  const XVecC<NC> xf{b, vc<NC>(ghost), nloc};    // vals carry D^-1 (see k_cheb_first_s)
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int row = r0 + threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && row < nrows;
    VecC<NC> d = vzero<NC>(), bi = d;
    if (mine) { d = dinv[row]; bi = b[row]; }
    const VecC<NC> sum = stream_row_block_c<RB, NC, NT>(rowptr, col, vals, xf, r0, nrows, lds);
    if (mine) {
      VecC<NC> x0, o;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        x0.c[i] = s * d.c[i] * bi.c[i];
        o.c[i] = c1 * x0.c[i] + c2 * d.c[i] * (bi.c[i] - s * sum.c[i]);
      }
      if (p0) p0[row] = x0;
      pn[row] = o;
    }
    __syncthreads();
  }
}

// ==========================================================================
// LDS-staged VECTOR TILES for the multi-component operators (north-star:
// "CSR SpMV ... with coalesced HBM row-pointer/col-index reads and LDS-staged
// vector tiles").  The kernels above stage PRODUCTS in LDS and gather the
// vector per entry from L1/L2: on a 3-D P2 stencil (29 entries per row, the
// gathered nodes of a row in ~19 short runs) every wave-wide gather touches
// ~30 cache lines for 1.5 KB of payload, and the texture-addresser, not HBM,
// bounds the kernel (profiles/r02_s_*: TA busy 61 %).  Here the set-up gives
// every row block the list of CONTIGUOUS column segments it touches
// (pcd_setup.hip build_vec_tile): the workgroup loads those segments into
// LDS with coalesced loads - every line once per block - and the row sums
// gather from the tile with 16-bit offsets instead of from L1 / L2
// LDS; (value, offset) pairs come straight from the matrix stream.  No product
// array, 10 B instead of 12 B of matrix stream per entry.
//   blocks:  <= ROWS consecutive rows, chosen greedily so that the tile
//            holds <= kVtNodes nodes
//   tsrc:    vector node of every tile slot (ascending: runs of consecutive
//            nodes); nodes >= nloc live in `ghost`
//   rowoff:  entry offsets of the block's rows relative to its first entry
//   loc:     tile offset of every entry's column (uint16)
// ==========================================================================
#ifndef PCD_VT_NODES
#define PCD_VT_NODES 768
#endif
#ifndef PCD_VT_U
#define PCD_VT_U 8
#endif
#ifndef PCD_VT_U2
#define PCD_VT_U2 (PCD_VT_U / 2)
#endif
// rows per block = template parameter ROWS of the kernels (256 / ROWS lanes
// share a row): 64, chosen on the host per operator
constexpr int kVtNodes = PCD_VT_NODES;      // tile nodes (x NC doubles)
constexpr int vt_rowoff(int rows) { return rows + 2; }   // row offsets per block (rows + 1, padded)
static_assert(kVtNodes % kBlock == 0, "one lane per tile node, whole passes");
// per block: x = first row, y = first entry, z = first slot in `tsrc`,
// w = rows | tile nodes << 8.  tsrc[z + t] = vector node of tile slot t (the
// block's distinct columns, ascending: runs of consecutive nodes, so a wave's
// loads of the vector are coalesced).  A block costs two dependent memory
// round trips after its descriptor: (row offsets, tile sources, epilogue
// operands), then (vector tile, matrix entries).  Earlier forms: a loop over
// segment descriptors serialised one load latency per segment (70 us per
// launch at cube N = 32 against 35 for the gather kernel); a binary search of
// the segments in LDS per tile slot (35 us: the LDS waits took the place of
// the texture-addresser stalls, profiles/r04_f_*).
// This DIRECT form serves operators that stay in the caches between launches;
// operators streamed from HBM take the lane-major form below (k_*_lm).  (A
// third form that staged the block's entries in LDS with non-temporal loads
// was the HBM form of round 4: 85.2 against 76.7 us at cube N = 48, 341.8
// against 285.6 at N = 73 - profiles/r05_d_*; removed.)
template <int NC, int ROWS>
__device__ __forceinline__ VecC<NC> tile_row_block(
    const int4 d, const unsigned short* __restrict__ rowoff, int blk,
    const int* __restrict__ tsrc, const double* __restrict__ val,
    const unsigned short* __restrict__ loc, const double* x,
    const double* ghost, int nloc, double* tile,
    const double* xscale = nullptr, const double* gscale = nullptr) {
  // (xscale / gscale: the tile holds x .* xscale - one factor per NODE, its
  // first component's, as the diagonal of F (x) I repeats: the zero-guess
  // first step gathers D^-1 b with the unscaled operator instead of b with a
  // column-scaled COPY of it, which would be a second 8 B / entry array
  // streaming through the Infinity Cache once per cycle)
  constexpr int TPR = kBlock / ROWS;
  constexpr int kVtRowOff = vt_rowoff(ROWS);
  const int lr = threadIdx.x / TPR, sub = threadIdx.x % TPR;
  const int k0 = d.y, nr = d.w & 0xff, tn = d.w >> 8;
  const bool mine = lr < nr;
  const int ra = mine ? rowoff[blk * kVtRowOff + lr] : 0;
  const int rb = mine ? rowoff[blk * kVtRowOff + lr + 1] : 0;
  int node[kVtNodes / kBlock];
#pragma unroll
  for (int u = 0; u < kVtNodes / kBlock; ++u) {
    const int t = threadIdx.x + u * kBlock;
    node[u] = t < tn ? tsrc[d.z + t] : -1;
  }
  const double* src[kVtNodes / kBlock];
#pragma unroll
  for (int u = 0; u < kVtNodes / kBlock; ++u)
    src[u] = node[u] < 0 ? nullptr
             : (node[u] < nloc ? x + (size_t)NC * node[u] : ghost + (size_t)NC * (node[u] - nloc));
  // (one node = one VecC: a 16-byte load for two components, 16 + 8 for three)
  double tv[kVtNodes / kBlock][NC];
#pragma unroll
  for (int u = 0; u < kVtNodes / kBlock; ++u) {
    VecC<NC> t = vzero<NC>();
    if (src[u]) {
      t = *vc<NC>(src[u]);
      if (xscale) {
        const double dd = node[u] < nloc ? xscale[(size_t)NC * node[u]]
                                         : gscale[(size_t)NC * (node[u] - nloc)];
#pragma unroll
        for (int i = 0; i < NC; ++i) t.c[i] *= dd;
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) tv[u][i] = t.c[i];
  }
  VecC<NC> s = vzero<NC>();
  // the first entries of my row straight from the stream (the lanes of a row
  // read contiguous bytes per step)
  // (entries in flight per lane; two components: 4 - with 4 lanes per row
  // that covers the 2-D P2 rows, and 58 instead of 70 VGPRs are an eighth
  // wave per SIMD: cavity level 6 15.54 -> 15.20 us; three components:
  // 8 - N = 32 29.6, 30.9 with 4)
  constexpr int U = NC == 2 ? PCD_VT_U2 : PCD_VT_U;
  double v[U];
  int o[U];
  // a lane takes PAIRS of consecutive entries: one 16-byte load of the
  // values and one 4-byte load of the offsets per pair (8- and 2-byte
  // aligned: the hardware's unaligned mode), half the load instructions
  typedef double dv2u __attribute__((ext_vector_type(2), aligned(8)));
  typedef unsigned short us2u __attribute__((ext_vector_type(2), aligned(2)));
#pragma unroll
  for (int u = 0; u < U / 2; ++u) {
    const int jj = ra + 2 * sub + u * 2 * TPR;
    const bool in = jj < rb, in2 = jj + 1 < rb;
    dv2u vv = dv2u(0.0);
    us2u ll = us2u((unsigned short)0);
    if (in) {
      vv = *reinterpret_cast<const dv2u*>(val + k0 + jj);
      ll = *reinterpret_cast<const us2u*>(loc + k0 + jj);
    }
    v[2 * u] = vv.x; o[2 * u] = ll.x;
    v[2 * u + 1] = in2 ? vv.y : 0.0; o[2 * u + 1] = in2 ? (int)ll.y : 0;
  }
#pragma unroll
  for (int u = 0; u < kVtNodes / kBlock; ++u)
    if (src[u]) {
#pragma unroll
      for (int i = 0; i < NC; ++i) tile[i * kVtNodes + threadIdx.x + u * kBlock] = tv[u][i];
    }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] += v[u] * tile[i * kVtNodes + o[u]];
  }
  for (int jj = ra + sub + U * TPR; jj < rb; jj += TPR) {     // long rows
    const double vv = val[k0 + jj];
    const int oo = loc[k0 + jj];
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] += vv * tile[i * kVtNodes + oo];
  }
#pragma unroll
  for (int m = TPR / 2; m > 0; m >>= 1) {
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] += __shfl_xor(s.c[i], m);
  }
  return s;
}

#define PCD_VT_SHARED(NC) __shared__ double tile[NC * kVtNodes]

// (`blist`: the blocks to run, or null = all of them - several ranks with
// PCD_OVERLAP=1 run the blocks that read no ghost column while the halo is
// on its way and the others after it has landed)
#define PCD_VT_ARGS                                                                   \
  int nblocks, const int* __restrict__ blist, const int4* __restrict__ desc,          \
  const unsigned short* __restrict__ rowoff,                                          \
  const int* __restrict__ tsrc, const double* __restrict__ val,                        \
  const unsigned short* __restrict__ loc

template <int MODE, int NC, int ROWS>
__global__ __launch_bounds__(kBlock) void k_spmv_tc(
    PCD_VT_ARGS, const double* x, const double* ghost, int nloc, const double* add_,
    double* y_) {
  PCD_VT_SHARED(NC);
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  int b0, b1;
  row_block_range(nblocks, ROWS, b0, b1, true);
  for (int bq = b0; bq < b1; ++bq) {
    const int blk = blist ? blist[bq] : bq;
    const int4 d = desc[blk];
    const int lr = threadIdx.x / (kBlock / ROWS);
    const int row = d.x + lr;
    const bool mine = threadIdx.x % (kBlock / ROWS) == 0 && lr < (d.w & 0xff);
    VecC<NC> a = vzero<NC>();
    if ((MODE == 1 || MODE == 2) && mine) a = add[row];
    const VecC<NC> s = tile_row_block<NC, ROWS>(d, rowoff, blk, tsrc, val, loc, x, ghost, nloc, tile);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
    __syncthreads();
  }
}

template <int NC, int ROWS>
__global__ __launch_bounds__(kBlock) void k_cheb_step_tc(
    PCD_VT_ARGS, const double* __restrict__ dinv_, const double* b_, const double* pm_,
    const double* pk_, double* pn_, double c0, double c1, double c2,
    const double* ghost, int nloc) {
  PCD_VT_SHARED(NC);
  const VecC<NC>*dinv = vc<NC>(dinv_), *b = vc<NC>(b_), *pm = vc<NC>(pm_),
               *pk = vc<NC>(pk_);
  VecC<NC>* pn = vc<NC>(pn_);
  int b0, b1;
  row_block_range(nblocks, ROWS, b0, b1, true);
  for (int bq = b0; bq < b1; ++bq) {
    const int blk = blist ? blist[bq] : bq;
    const int4 d4 = desc[blk];
    const int lr = threadIdx.x / (kBlock / ROWS);
    const int row = d4.x + lr;
    const bool mine = threadIdx.x % (kBlock / ROWS) == 0 && lr < (d4.w & 0xff);
    VecC<NC> bi = vzero<NC>(), d = bi, xk = bi, xm = bi;
    if (mine) {
      bi = b[row]; d = dinv[row]; xk = pk[row];
      if (c0 != 0.0) xm = pm[row];
    }
    const VecC<NC> s = tile_row_block<NC, ROWS>(d4, rowoff, blk, tsrc, val, loc, pk_, ghost, nloc, tile);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = c0 * xm.c[i] + c1 * xk.c[i] + c2 * d.c[i] * (bi.c[i] - s.c[i]);
      pn[row] = o;
    }
    __syncthreads();
  }
}

template <int NC, int ROWS>
__global__ __launch_bounds__(kBlock) void k_cheb_first_tc(
    PCD_VT_ARGS, const double* __restrict__ dinv_, const double* b_, double* p0_,
    double* pn_, double s, double c1, double c2, const double* ghost, int nloc,
    const double* __restrict__ dghost) {
  PCD_VT_SHARED(NC);
  const VecC<NC>*dinv = vc<NC>(dinv_), *b = vc<NC>(b_);
  VecC<NC>*p0 = vc<NC>(p0_), *pn = vc<NC>(pn_);
  int b0, b1;
  row_block_range(nblocks, ROWS, b0, b1, true);
  for (int bq = b0; bq < b1; ++bq) {
    const int blk = blist ? blist[bq] : bq;
    const int4 d4 = desc[blk];
    const int lr = threadIdx.x / (kBlock / ROWS);
    const int row = d4.x + lr;
    const bool mine = threadIdx.x % (kBlock / ROWS) == 0 && lr < (d4.w & 0xff);
    VecC<NC> d = vzero<NC>(), bi = d;
    if (mine) { d = dinv[row]; bi = b[row]; }
    // (the tile holds D^-1 b - b with its halo when there are several ranks,
    // the reciprocal diagonal of the ghost columns kept from its own exchange)
    const VecC<NC> sum = tile_row_block<NC, ROWS>(d4, rowoff, blk, tsrc, val, loc, b_, ghost, nloc, tile,
                                                  dinv_, dghost);
    if (mine) {
      VecC<NC> x0, o;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        x0.c[i] = s * d.c[i] * bi.c[i];
        o.c[i] = c1 * x0.c[i] + c2 * d.c[i] * (bi.c[i] - s * sum.c[i]);
      }
      if (p0) p0[row] = x0;
      pn[row] = o;
    }
    __syncthreads();
  }
}

// ==========================================================================
// LANE-MAJOR form of the vector-tile kernels (operators streamed from HBM).
// Round 4's form for such operators passed every (value, offset) pair through
// LDS: 10 B written + 10 B + 8 NC B read per entry, and the entry buffers were
// half of the workgroup's LDS, i.e. they decided how many workgroups a CU
// holds - which is what its launch time followed (profiles/r04_j_*: T = a +
// b / W).  Here the
// set-up stores each block's entries LANE-MAJOR: lane t owns the kLmE = 8
// CONSECUTIVE entries 8 t .. 8 t + 7 of the block's row-major stream (blocks
// are padded to whole lanes with zero entries: < 0.4 % of the stream); the
// values of the pair (2 k, 2 k + 1) of all L active lanes sit side by side
// (16-byte non-temporal loads, fully coalesced), a lane's eight 16-bit offsets
// are one 16-byte load.  Entries go straight to REGISTERS; LDS holds the
// vector tile, one partial sum per lane and one sum per row:
//   loc:   bits 0-10 tile slot of the entry's column, bit 15 = last entry of
//          its row; bits 11-14 of a lane's entries 0 / 1 = low / high nibble
//          of the block row its first entry belongs to
//   lane:  s += v * tile[slot]; at a row end R[row] = s, next row, s = 0;
//          after its entries T[lane] = s (the part of a row that goes on in
//          the next lane)
//   row r (one lane per row, epilogue operands coalesced): entries [ra, rb)
//          live in lanes la = ra / 8 .. lb = (rb - 1) / 8:
//          sum = T[la] + ... + T[lb - 1] + R[r]   - a fixed order: reproducible
// Two barriers per block, LDS per entry 8 NC B read.
// ==========================================================================
#ifndef PCD_LM_NODES3
#define PCD_LM_NODES3 512
#endif
#ifndef PCD_LM_NODES2
#define PCD_LM_NODES2 768
#endif
#ifndef PCD_LM_ROWS3
#define PCD_LM_ROWS3 64
#endif
#ifndef PCD_LM_ROWS2
#define PCD_LM_ROWS2 160
#endif
constexpr int kLmE = 8;                              // entries per lane
constexpr int kLmEntries = kLmE * kBlock;            // entries per block
constexpr int lm_nodes(int nc) { return nc == 3 ? PCD_LM_NODES3 : PCD_LM_NODES2; }
constexpr int lm_rows(int nc) { return nc == 3 ? PCD_LM_ROWS3 : PCD_LM_ROWS2; }
static_assert(lm_nodes(2) % 64 == 0 && lm_nodes(3) % 64 == 0, "tile nodes: whole waves");
static_assert(lm_nodes(2) <= 2048 && lm_nodes(3) <= 2048, "11-bit tile slots");
static_assert(lm_rows(2) <= 256 && lm_rows(3) <= 256, "one lane per row, 8-bit first rows");
// descriptor: x = first row, y = lanes of all blocks before this one (its
// values start at 8 y doubles, its offsets at 8 y shorts), z = first slot in
// `tsrc` (= block x lm_nodes: a fixed stride, padded with -1), w = rows |
// tile nodes << 9 | active lanes << 20

#define PCD_LM_SHARED(NC)                                   \
  __shared__ double tile[NC * lm_nodes(NC)];                \
  __shared__ double lmR[NC * lm_rows(NC)];                  \
  __shared__ double lmT[NC * kBlock]

// Measured and not kept (profiles/r05_b_* ... r05_i_*; the switches are gone):
// non-temporal accesses to the epilogue's vectors (same-box A/B: cube N = 73
// 310.9 / 310.6 us, N = 48 75.3 / 75.2, cavity level 7 52.7 -> 55.6), non-
// temporal index streams (+ 6 %), the tile sources after the entries (cube
// N = 48: 75.0 against 69.7 us) or at their compact positions (69.7 against
// 68.5), the epilogue's operands with the block's first loads (N = 73: 312 ->
// 309, level 7 55.5 -> 53.7 with them late), a register allocation forced to
// eight waves per SIMD (spills: 81.3 against 74.0 at N = 48), the tile of a
// three-component operator gathered per DOUBLE instead of per node (lane t the
// doubles t + 256 j of the tile's 3 TN, so that a wave's load covers 512
// contiguous bytes where the nodes are consecutive: N = 48 69.2 -> 103.5 us,
// N = 73 301 -> 412, profiles/r05_p_*).
template <int NC>
struct LmRegs {
  typedef double dv2 __attribute__((ext_vector_type(2)));
  typedef unsigned uv4 __attribute__((ext_vector_type(4)));
  dv2 ve[kLmE / 2];
  uv4 le;
  int ra, rb;
  bool act;
};

// entries -> registers, vector tile -> LDS, barrier
template <int NC>
__device__ __forceinline__ void lm_stage(
    LmRegs<NC>& g, const int4 d, const unsigned short* __restrict__ rowoff, int blk,
    const int* __restrict__ tsrc, const double* __restrict__ val,
    const unsigned short* __restrict__ loc, const double* x,
    const double* ghost, int nloc, double* tile,
    const double* xscale = nullptr, const double* gscale = nullptr) {
  constexpr int TN = lm_nodes(NC), RO = vt_rowoff(lm_rows(NC));
  typedef typename LmRegs<NC>::dv2 dv2;
  typedef typename LmRegs<NC>::uv4 uv4;
  const int t = threadIdx.x;
  const int nr = d.w & 0x1ff, L = (d.w >> 20) & 0x1ff;
  constexpr int NP = (TN + kBlock - 1) / kBlock;      // passes over the tile
  int node[NP];
  // Loads return in the order they were issued: the tile sources head the
  // block's only chain of two dependent round trips (sources -> vector
  // nodes), so they go first - from an address that depends on the block
  // index alone (stride TN, padded with -1), i.e. together with the descriptor
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    const int q = t + u * kBlock;
    node[u] = q < TN ? tsrc[(size_t)blk * TN + q] : -1;
  }
  // the matrix stream: addresses known with the descriptor
  g.act = t < L;
  const dv2* vb = reinterpret_cast<const dv2*>(val) + (size_t)d.y * (kLmE / 2);
#pragma unroll
  for (int k = 0; k < kLmE / 2; ++k)
    g.ve[k] = g.act ? __builtin_nontemporal_load(vb + k * L + t) : dv2(0.0);
  g.le = uv4(0u);
  if (g.act) g.le = __builtin_nontemporal_load(reinterpret_cast<const uv4*>(loc) + d.y + t);
  // entries [ra, rb) of the row whose sum this lane completes
  g.ra = t < nr ? rowoff[blk * RO + t] : 0;
  g.rb = t < nr ? rowoff[blk * RO + t + 1] : 0;
#pragma unroll
  for (int u = 0; u < NP; ++u) {
    if (node[u] < 0) continue;
    const bool own = node[u] < nloc;
    const double* src = own ? x + (size_t)NC * node[u] : ghost + (size_t)NC * (node[u] - nloc);
    VecC<NC> v = *vc<NC>(src);
    if (xscale) {
      const double dd = own ? xscale[(size_t)NC * node[u]] : gscale[(size_t)NC * (node[u] - nloc)];
#pragma unroll
      for (int i = 0; i < NC; ++i) v.c[i] *= dd;
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) tile[i * TN + t + u * kBlock] = v.c[i];
  }
  __syncthreads();
}

// lane sums, barrier, row sums (valid on the lanes t < rows of the block)
template <int NC>
__device__ __forceinline__ VecC<NC> lm_sum(const LmRegs<NC>& g, const double* tile,
                                           double* lmR, double* lmT) {
  constexpr int TN = lm_nodes(NC), RM = lm_rows(NC);
  const int t = threadIdx.x;
  if (g.act) {
    int cur = ((g.le.x >> 11) & 0xf) | (((g.le.x >> 27) & 0xf) << 4);
    VecC<NC> s = vzero<NC>();
#pragma unroll
    for (int u = 0; u < kLmE; ++u) {
      const unsigned w = u & 1 ? g.le[u >> 1] >> 16 : g.le[u >> 1] & 0xffffu;
      const double vv = u & 1 ? g.ve[u >> 1].y : g.ve[u >> 1].x;
      const int oo = w & 0x7ff;
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += vv * tile[i * TN + oo];
      if (w & 0x8000u) {
#pragma unroll
        for (int i = 0; i < NC; ++i) { lmR[i * RM + cur] = s.c[i]; s.c[i] = 0.0; }
        ++cur;
      }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) lmT[i * kBlock + t] = s.c[i];
  }
  __syncthreads();
  VecC<NC> sum = vzero<NC>();
  if (g.rb > g.ra) {
    const int la = g.ra / kLmE, lb = (g.rb - 1) / kLmE;
    for (int q = la; q < lb; ++q) {
#pragma unroll
      for (int i = 0; i < NC; ++i) sum.c[i] += lmT[i * kBlock + q];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) sum.c[i] += lmR[i * RM + t];
  }
  return sum;
}

// (the epilogue's operands are requested after the block's first barrier,
// when the tile's registers are free again)
template <int MODE, int NC>
__global__ __launch_bounds__(kBlock) void k_spmv_lm(
    PCD_VT_ARGS, const double* x, const double* ghost, int nloc, const double* add_,
    double* y_) {
  PCD_LM_SHARED(NC);
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  int b0, b1;
  row_block_range(nblocks, lm_rows(NC), b0, b1, true);
  for (int bq = b0; bq < b1; ++bq) {
    const int blk = blist ? blist[bq] : bq;
    const int4 d = desc[blk];
    const int row = d.x + threadIdx.x;
    const bool mine = (int)threadIdx.x < (d.w & 0x1ff);
    VecC<NC> a = vzero<NC>();
    LmRegs<NC> g;
    lm_stage<NC>(g, d, rowoff, blk, tsrc, val, loc, x, ghost, nloc, tile);
    if ((MODE == 1 || MODE == 2) && mine) a = add[row];
    const VecC<NC> s = lm_sum<NC>(g, tile, lmR, lmT);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
  }
}

template <int NC>
__global__ __launch_bounds__(kBlock) void k_cheb_step_lm(
    PCD_VT_ARGS, const double* __restrict__ dinv_, const double* b_, const double* pm_,
    const double* pk_, double* pn_, double c0, double c1, double c2,
    const double* ghost, int nloc) {
  PCD_LM_SHARED(NC);
  const VecC<NC>*dinv = vc<NC>(dinv_), *b = vc<NC>(b_), *pm = vc<NC>(pm_),
               *pk = vc<NC>(pk_);
  VecC<NC>* pn = vc<NC>(pn_);
  int b0, b1;
  row_block_range(nblocks, lm_rows(NC), b0, b1, true);
  for (int bq = b0; bq < b1; ++bq) {
    const int blk = blist ? blist[bq] : bq;
    const int4 d4 = desc[blk];
    const int row = d4.x + threadIdx.x;
    const bool mine = (int)threadIdx.x < (d4.w & 0x1ff);
    VecC<NC> bi = vzero<NC>(), d = bi, xk = bi, xm = bi;
    LmRegs<NC> g;
    lm_stage<NC>(g, d4, rowoff, blk, tsrc, val, loc, pk_, ghost, nloc, tile);
    if (mine) {
      bi = b[row]; d = dinv[row]; xk = pk[row];
      if (c0 != 0.0) xm = pm[row];
    }
    const VecC<NC> s = lm_sum<NC>(g, tile, lmR, lmT);
    if (mine) {
      VecC<NC> o;
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = c0 * xm.c[i] + c1 * xk.c[i] + c2 * d.c[i] * (bi.c[i] - s.c[i]);
      pn[row] = o;
    }
  }
}

template <int NC>
__global__ __launch_bounds__(kBlock) void k_cheb_first_lm(
    PCD_VT_ARGS, const double* __restrict__ dinv_, const double* b_, double* p0_,
    double* pn_, double s, double c1, double c2, const double* ghost, int nloc,
    const double* __restrict__ dghost) {
  PCD_LM_SHARED(NC);
  const VecC<NC>*dinv = vc<NC>(dinv_), *b = vc<NC>(b_);
  VecC<NC>*p0 = vc<NC>(p0_), *pn = vc<NC>(pn_);
  int b0, b1;
  row_block_range(nblocks, lm_rows(NC), b0, b1, true);
  for (int bq = b0; bq < b1; ++bq) {
    const int blk = blist ? blist[bq] : bq;
    const int4 d4 = desc[blk];
    const int row = d4.x + threadIdx.x;
    const bool mine = (int)threadIdx.x < (d4.w & 0x1ff);
    VecC<NC> d = vzero<NC>(), bi = d;
    LmRegs<NC> g;
    // (the tile holds D^-1 b - b with its halo when there are several ranks,
    // the reciprocal diagonal of the ghost columns kept from its own exchange)
    lm_stage<NC>(g, d4, rowoff, blk, tsrc, val, loc, b_, ghost, nloc, tile, dinv_, dghost);
    if (mine) { d = dinv[row]; bi = b[row]; }
    const VecC<NC> sum = lm_sum<NC>(g, tile, lmR, lmT);
    if (mine) {
      VecC<NC> x0, o;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        x0.c[i] = s * d.c[i] * bi.c[i];
        o.c[i] = c1 * x0.c[i] + c2 * d.c[i] * (bi.c[i] - s * sum.c[i]);
      }
      if (p0) p0[row] = x0;
      pn[row] = o;
    }
  }
}

// ==========================================================================
// m Chebyshev-Jacobi steps in ONE launch (ChebPatch, pcd_internal.hpp): a
// workgroup holds the patch of its cluster in LDS - b, D^-1 and a ring of three
// iterates - and runs step k on the nodes within m - k edges of the cluster
// (they are the first cnt[m - k] nodes of the patch: the list is ordered by
// distance); the matrix rows come as ELL slices (entry e of all rows side by
// side: coalesced) with patch-local 16-bit columns (WMAX: the widest row the
// instantiation holds).  Zero initial guess:
// p0 = scale D^-1 b, then p_{k+1} = c0 p_{k-1} + c1 p_k + c2 D^-1 (b - A p_k)
// with the host's coefficients (solve_cheb).  What a cluster computes for the
// nodes around it other clusters compute as well: redundant arithmetic instead
// of m - 1 dependent launches.
// ==========================================================================
struct ChebPatchCoef {
  double c0[8], c1[8], c2[8];
  double scale;
  int m;
};
constexpr int kPatchThreads = 512;
constexpr int kPatchRows = 3;                 // matrix rows per thread: 3 x 512 = the 1536 patch nodes
// The matrix rows a thread works on are the same in every step: they are read
// ONCE, into registers (kPatchRows x WMAX values and patch-local columns per
// thread, every loop unrolled) - the steps themselves touch LDS only.
template <int WMAX>
static __global__ __launch_bounds__(kPatchThreads) void k_cheb_patch(
    const int4* __restrict__ desc, const int* __restrict__ cnt_, const int* __restrict__ node,
    const unsigned short* __restrict__ col, const double* __restrict__ val,
    const double* __restrict__ dinv, const double* b, double* x, ChebPatchCoef cf) {
  constexpr int PM = 1536, NT = kPatchThreads, RT = kPatchRows;
  static_assert(NT * RT >= PM, "every patch row has a thread");
  __shared__ double lds[5 * PM];
  double *bl = lds, *dl = lds + PM, *p0 = lds + 2 * PM, *p1 = lds + 3 * PM, *p2 = lds + 4 * PM;
  const int4 d = desc[blockIdx.x];
  const int* cnt = cnt_ + (size_t)blockIdx.x * 9;
  const int P = d.z, Rpad = d.w & 0xffff, W = d.w >> 16;
  const int t = threadIdx.x;
  const int R0 = cnt[cf.m - 1];                       // rows that are ever updated
  double av[RT][WMAX];
  unsigned short ac[RT][WMAX];
  const unsigned short* cb = col + d.y;
  const double* vb = val + d.y;
#pragma unroll
  for (int j = 0; j < RT; ++j) {
    const int r = t + j * NT;
#pragma unroll
    for (int e = 0; e < WMAX; ++e) {
      const bool in = r < R0 && e < W;
      av[j][e] = in ? vb[(size_t)e * Rpad + r] : 0.0;
      ac[j][e] = in ? cb[(size_t)e * Rpad + r] : (unsigned short)0;
    }
  }
  for (int q = t; q < P; q += NT) {
    const int nd = node[d.x + q];
    const double bq = b[nd], dq = dinv[nd];
    bl[q] = bq; dl[q] = dq; p0[q] = cf.scale * dq * bq;
  }
  __syncthreads();
  double *pm = p2, *pk = p0, *pn = p1;
  for (int it = 0; it < cf.m; ++it) {
    const int Rk = cnt[cf.m - 1 - it];
    const double c0 = cf.c0[it], c1 = cf.c1[it], c2 = cf.c2[it];
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int r = t + j * NT;
      if (r < Rk) {
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < WMAX; ++e) s += av[j][e] * pk[ac[j][e]];
        const double keep = c0 != 0.0 ? c0 * pm[r] : 0.0;
        pn[r] = keep + c1 * pk[r] + c2 * dl[r] * (bl[r] - s);
      }
    }
    __syncthreads();
    double* o = pm; pm = pk; pk = pn; pn = o;
  }
  const int own = cnt[0];
  for (int r = t; r < own; r += NT) x[node[d.x + r]] = pk[r];
}

// lane-major values from F's row-major ones: out[s] = pos[s] < 0 ? 0 : valc[pos[s]]
static __global__ __launch_bounds__(kBlock) void k_lm_values(
    int64_t nslots, const int* __restrict__ pos, const double* valc, double* out) {
  for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < nslots;
       s += (int64_t)gridDim.x * kBlock) {
    const int p = pos[s];
    out[s] = p < 0 ? 0.0 : valc[p];
  }
}

// valc[k] = val[pos[k]]; *mismatch |= (val[pos[c*nnzc + k]] differs, c >= 1):
// pos holds, component-major, where the entry k of F sits in each component's
// rows of the full matrix
static __global__ __launch_bounds__(kBlock) void k_kron_gather(
    int nnzc, int nc, const int* __restrict__ pos, const double* val,
    double* valc, int* mismatch) {
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < nnzc;
       k += gridDim.x * kBlock) {
    const double a = val[pos[k]];
    valc[k] = a;
    for (int c = 1; c < nc; ++c)
      if (val[pos[(int64_t)c * nnzc + k]] != a) *mismatch = 1;
  }
}

// ---- BLAS-1 glue of the apply bodies --------------------------------------
static __global__ __launch_bounds__(kBlock) void k_copy(int n, const double* x,
                                                  double* y) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    y[i] = x[i];
}

// y = a*x + b*y  (b == 0 never reads y)
static __global__ __launch_bounds__(kBlock) void k_axpby(int n, double a,
                                                   const double* x, double b,
                                                   double* y) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

// halo pack: out[i] = x[idx[i]]
static __global__ __launch_bounds__(kBlock) void k_pack(
    int n, const int* __restrict__ idx, const double* x, double* out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    out[i] = x[idx[i]];
}

// slot[j] = sum parts[j*stride .. j*stride + n)   (one workgroup per j): the
// rank-local value that then goes through ncclAllReduce
static __global__ __launch_bounds__(kBlock) void k_sum_parts(const double* parts,
                                                       int n, int stride,
                                                       double* slot) {
  __shared__ double sm[4];
  const double s = reduce_parts(parts + (int64_t)blockIdx.x * stride, n, sm);
  if (threadIdx.x == 0) slot[blockIdx.x] = s;
}

// in-process test backend (ranks = threads on one GPU): out = sum over ranks
// of their operands, added in rank order on every rank (identical results)
struct RankBufs { const double* p[16]; int n; };
static __global__ __launch_bounds__(kBlock) void k_sum_ranks(RankBufs bufs, int64_t count,
                                                       double* out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * kBlock) {
    double s = 0.0;
    for (int r = 0; r < bufs.n; ++r) s += bufs.p[r][i];
    out[i] = s;
  }
}

// z = x with the subfield BC values inserted (copy + VecSetValues fused):
// slot[i] = position in val[] of row i's BC value, or -1
static __global__ __launch_bounds__(kBlock) void k_copy_bc(
    int n, const double* x, const int* __restrict__ slot,
    const double* __restrict__ val, double* z) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock) {
    const int m = slot[i];
    z[i] = m >= 0 ? val[m] : x[i];
  }
}

// SubfieldBC::apply: x[idx[i]] = val[i]  (VecSetValues INSERT)
static __global__ __launch_bounds__(kBlock) void k_bc_set(
    int n_bc, const int* __restrict__ idx, const double* __restrict__ val,
    double* x) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n_bc) x[idx[i]] = val[i];
}

// fieldsplit scatter: out[i] = in[perm[i]]  /  out[perm[i]] = in[i]
// (four independent index -> value chains per thread in flight)
static __global__ __launch_bounds__(kBlock) void k_gather(
    int n, const int* __restrict__ perm, const double* in, double* out) {
  const int stride = gridDim.x * kBlock;
  int i = blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const int p0 = perm[i], p1 = perm[i + stride], p2 = perm[i + 2 * stride],
              p3 = perm[i + 3 * stride];
    const double v0 = in[p0], v1 = in[p1], v2 = in[p2], v3 = in[p3];
    out[i] = v0; out[i + stride] = v1; out[i + 2 * stride] = v2; out[i + 3 * stride] = v3;
  }
  for (; i < n; i += stride) out[i] = in[perm[i]];
}
static __global__ __launch_bounds__(kBlock) void k_scatter(
    int n, const int* __restrict__ perm, const double* in, double* out) {
  const int stride = gridDim.x * kBlock;
  int i = blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const int p0 = perm[i], p1 = perm[i + stride], p2 = perm[i + 2 * stride],
              p3 = perm[i + 3 * stride];
    const double v0 = in[i], v1 = in[i + stride], v2 = in[i + 2 * stride],
                 v3 = in[i + 3 * stride];
    out[p0] = v0; out[p1] = v1; out[p2] = v2; out[p3] = v3;
  }
  for (; i < n; i += stride) out[perm[i]] = in[i];
}

// block values from the caller's monolithic value array
static __global__ __launch_bounds__(kBlock) void k_gather_vals(
    int64_t nnz, const int64_t* __restrict__ src, const double* vals,
    double* out) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock)
    out[k] = vals[src[k]];
}

// reciprocal diagonal (1 where the diagonal is zero or absent)
static __global__ __launch_bounds__(kBlock) void k_dinv(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, double* dinv) {
  const int row = blockIdx.x * kBlock + threadIdx.x;
  if (row >= nrows) return;
  double d = 0.0;
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k)
    if (col[k] == row) d += val[k];
  dinv[row] = (d != 0.0) ? 1.0 / d : 1.0;
}

// ---- GMRES: classical Gram-Schmidt as batched dots -------------------------
// parts[(j0+jj)*G + blk] = sum_i V[(j0+jj)*ld + i] * w[i]
constexpr int kDotTile = 8;
static __global__ __launch_bounds__(kBlock) void k_mdot(
    int64_t n, const double* V, int64_t ld, int nvec, const double* w,
    double* parts, int G) {
  __shared__ double sm[4];
  const int j0 = blockIdx.y * kDotTile;
  double acc[kDotTile];
#pragma unroll
  for (int jj = 0; jj < kDotTile; ++jj) acc[jj] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)G * kBlock) {
    const double wi = w[i];
#pragma unroll
    for (int jj = 0; jj < kDotTile; ++jj)
      if (j0 + jj < nvec) acc[jj] += V[(int64_t)(j0 + jj) * ld + i] * wi;
  }
#pragma unroll
  for (int jj = 0; jj < kDotTile; ++jj) {
    const double s = block_sum(acc[jj], sm);
    if (threadIdx.x == 0 && j0 + jj < nvec)
      parts[(int64_t)(j0 + jj) * G + blockIdx.x] = s;
  }
}

// h[j] = sum_blk parts[j*G + blk]   (one workgroup per j)
static __global__ __launch_bounds__(kBlock) void k_mdot_reduce(const double* parts,
                                                         int G, double* h) {
  __shared__ double sm[4];
  const double s = reduce_parts(parts + (int64_t)blockIdx.x * G, G, sm);
  if (threadIdx.x == 0) h[blockIdx.x] = s;
}

// w -= sum_j h[j] V_j ; parts[blk] = sum w^2.  The basis vectors are read
// eight at a time into independent registers (the coefficients sit in LDS):
// a dependent chain of nvec loads per element left most of the bandwidth idle.
static __global__ __launch_bounds__(kBlock) void k_maxpy_norm(
    int64_t n, const double* __restrict__ V, int64_t ld, int nvec,
    const double* __restrict__ h, double* __restrict__ w, double sign,
    double* __restrict__ parts) {
  __shared__ double sm[4];
  __shared__ double hs[256];                 // restart <= 255 (checked by the host)
  for (int j = threadIdx.x; j < nvec; j += kBlock) hs[j] = sign * h[j];
  __syncthreads();
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    double wi = w[i];
    int j = 0;
    for (; j + 8 <= nvec; j += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = V[(int64_t)(j + u) * ld + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) wi += hs[j + u] * v[u];
    }
    for (; j < nvec; ++j) wi += hs[j] * V[(int64_t)j * ld + i];
    w[i] = wi;
    acc += wi * wi;
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) parts[blockIdx.x] = acc;
}

// nrm = sqrt(sum parts); w /= nrm; *out_nrm = nrm
static __global__ __launch_bounds__(kBlock) void k_normalize(
    int64_t n, double* w, const double* parts, int nparts, double* out_nrm) {
  __shared__ double sm[4];
  const double nrm = sqrt(reduce_parts(parts, nparts, sm));
  const double inv = (nrm != 0.0) ? 1.0 / nrm : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock)
    w[i] *= inv;
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_nrm = nrm;
}

// ==========================================================================
// Wave-per-row kernels for operators with LONG rows (pre-composed multigrid
// operators: 30 ... 300 entries per row).  One wave64 owns one row: lanes
// stride over its entries (coalesced 12-byte records), gather, and a shuffle
// tree sums - no LDS, no workgroup barrier, 8 waves per SIMD in flight.  The
// CSR-stream kernels above lose here: a row block of 32 such rows overflows
// the LDS tile several times over and serialises its passes.
// ==========================================================================
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_spmv_w(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const XVec x, const double* add,
    double* y) {
  const int lane = threadIdx.x & 63;
  const int wpb = kBlock / 64;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < nrows;
       row += gridDim.x * wpb) {
    const int b = rowptr[row], e = rowptr[row + 1];
    double s0 = 0.0, s1 = 0.0;
    int k = b + lane;
    for (; k + 64 < e; k += 128) {           // two independent records in flight
      const int c0 = col[k], c1 = col[k + 64];
      const double v0 = val[k], v1 = val[k + 64];
      s0 += v0 * x(c0); s1 += v1 * x(c1);
    }
    if (k < e) s0 += val[k] * x(col[k]);
    const double s = wave_sum(s0 + s1);
    if (lane == 0) {
      if (MODE == 0) y[row] = s;
      if (MODE == 1) y[row] = add[row] + s;
      if (MODE == 2) y[row] = add[row] - s;
      if (MODE == 3) y[row] = -s;
    }
  }
}

template <int MODE, int NC>
__global__ __launch_bounds__(kBlock) void k_spmv_wc(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x_, const double* ghost_,
    int nloc, const double* add_, double* y_) {
  const XVecC<NC> x{vc<NC>(x_), vc<NC>(ghost_), nloc};
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  const int lane = threadIdx.x & 63;
  const int wpb = kBlock / 64;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < nrows;
       row += gridDim.x * wpb) {
    const int b = rowptr[row], e = rowptr[row + 1];
    VecC<NC> s = vzero<NC>();
    int k = b + lane;
    for (; k + 64 < e; k += 128) {
      const int c0 = col[k], c1 = col[k + 64];
      const double v0 = val[k], v1 = val[k + 64];
      const VecC<NC> x0 = x(c0), x1 = x(c1);
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += v0 * x0.c[i] + v1 * x1.c[i];
    }
    if (k < e) {
      const double v0 = val[k];
      const VecC<NC> x0 = x(col[k]);
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += v0 * x0.c[i];
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] = wave_sum(s.c[i]);
    if (lane == 0) {
      VecC<NC> o;
      VecC<NC> a = vzero<NC>();
      if (MODE == 1 || MODE == 2) a = add[row];
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
  }
}

// Very long rows (>= 1024 entries: the residual-restriction product of a 3-D
// two-grid cycle has thousands per coarse row) of a multi-component operator:
// one workgroup per node row, as k_spmv_long does for scalar operators.
template <int MODE, int NC>
__global__ __launch_bounds__(kBlock) void k_spmv_longc(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x_, const double* ghost_,
    int nloc, const double* add_, double* y_) {
  __shared__ double sm[4];
  const XVecC<NC> x{vc<NC>(x_), vc<NC>(ghost_), nloc};
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int b = rowptr[row], e = rowptr[row + 1];
    VecC<NC> s0 = vzero<NC>(), s1 = vzero<NC>();
    int k = b + threadIdx.x;
    for (; k + kBlock < e; k += 2 * kBlock) {
      const int c0 = col[k], c1 = col[k + kBlock];
      const double v0 = val[k], v1 = val[k + kBlock];
      const VecC<NC> x0 = x(c0), x1 = x(c1);
#pragma unroll
      for (int i = 0; i < NC; ++i) { s0.c[i] += v0 * x0.c[i]; s1.c[i] += v1 * x1.c[i]; }
    }
    if (k < e) {
      const double v0 = val[k];
      const VecC<NC> x0 = x(col[k]);
#pragma unroll
      for (int i = 0; i < NC; ++i) s0.c[i] += v0 * x0.c[i];
    }
    VecC<NC> s;
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] = block_sum(s0.c[i] + s1.c[i], sm);
    if (threadIdx.x == 0) {
      VecC<NC> o, a = vzero<NC>();
      if (MODE == 1 || MODE == 2) a = add[row];
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
  }
}

// Dense operator (explicit coarse inverse), row-major, no column indices: one
// workgroup per (node) row; all NC components of a node share the row of the
// scalar inverse (inv(F (x) I) = inv(F) (x) I).  8 B per entry instead of the
// 12 B of a CSR record, and no dependent index load.
template <int MODE, int NC>
__global__ __launch_bounds__(kBlock) void k_dense_c(
    int nrows, int ncols, const double* __restrict__ M, const double* x_,
    const double* add_, double* y_) {
  __shared__ double sm[4];
  const VecC<NC>* x = vc<NC>(x_);
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const double* r = M + (size_t)row * ncols;
    VecC<NC> s0 = vzero<NC>(), s1 = vzero<NC>();
    int k = threadIdx.x;
    for (; k + kBlock < ncols; k += 2 * kBlock) {
      const double a0 = r[k], a1 = r[k + kBlock];
      const VecC<NC> x0 = x[k], x1 = x[k + kBlock];
#pragma unroll
      for (int i = 0; i < NC; ++i) { s0.c[i] += a0 * x0.c[i]; s1.c[i] += a1 * x1.c[i]; }
    }
    if (k < ncols) {
      const double a0 = r[k];
      const VecC<NC> x0 = x[k];
#pragma unroll
      for (int i = 0; i < NC; ++i) s0.c[i] += a0 * x0.c[i];
    }
    VecC<NC> s;
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] = block_sum(s0.c[i] + s1.c[i], sm);
    if (threadIdx.x == 0) {
      VecC<NC> o, a = vzero<NC>();
      if (MODE == 1 || MODE == 2) a = add[row];
#pragma unroll
      for (int i = 0; i < NC; ++i)
        o.c[i] = MODE == 0 ? s.c[i] : (MODE == 1 ? a.c[i] + s.c[i]
                                        : (MODE == 2 ? a.c[i] - s.c[i] : -s.c[i]));
      y[row] = o;
    }
  }
}

// ---- bandwidth probe: the practical roof next to which the path is priced ---
// 16 bytes per lane, unit stride, grid-stride loop: what a streaming kernel of
// this engine can reach on this box ("confirm with a device-to-
// device copy/triad microbench on the box and report THAT as the practical
// roof").  kind 0: a = b (16 B/entry-pair moved: 1 read + 1 write);
// kind 1: a = b + s c (triad: 2 reads + 1 write); kind 2: read-only sweep;
// kind 3: read-mostly (6 % writes, the mix of the dominant kernel); kind 4:
// read-only with non-temporal loads.
// read sweeps: one contiguous chunk per workgroup, eight independent 16-byte
// loads per lane in flight - the shape that reads fastest on this GPU with FEW
// workgroups per CU (profiles/r03_f_read_bandwidth_sweep.txt: 6.1-6.5 TB/s at
// 2-4 per CU against 5.3 at 32; non-temporal 7.1).  KIND 2: read-only; 3:
// read-mostly (one 16-byte store per sixteen loads of a lane: the 6 % writes
// of the fused Chebyshev step); 4: read-only, non-temporal.
template <int KIND>
__global__ __launch_bounds__(kBlock) void k_bw_read(int64_t n2, const double2* __restrict__ b,
                                                     double2* __restrict__ a) {
  typedef double dv2 __attribute__((ext_vector_type(2)));
  const dv2* bb = reinterpret_cast<const dv2*>(b);
  dv2* aa = reinterpret_cast<dv2*>(a);
  const int64_t per = (n2 + gridDim.x - 1) / gridDim.x;
  int64_t i = (int64_t)blockIdx.x * per + threadIdx.x;
  const int64_t end = min(n2, (int64_t)(blockIdx.x + 1) * per);
  dv2 acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = dv2(0.0);
  int trip = 0;
  for (; i + 7 * kBlock < end; i += 8 * kBlock, ++trip) {
    dv2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = KIND == 4 ? __builtin_nontemporal_load(bb + i + u * kBlock) : bb[i + u * kBlock];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += v[u];
    if (KIND == 3 && (trip & 1) == 0) aa[i >> 4] = acc[0];   // 1 store per 16 loads
  }
  for (; i < end; i += kBlock) acc[0] += bb[i];
#pragma unroll
  for (int u = 1; u < 8; ++u) acc[0] += acc[u];
  aa[n2 / 16 + (int64_t)blockIdx.x * kBlock + threadIdx.x] = acc[0];
}

// copy (kind 0) / triad (kind 1), grid-stride
static __global__ __launch_bounds__(kBlock) void k_bw_probe(
    int kind, int64_t n2, const double2* __restrict__ b,
    const double2* __restrict__ c, double s, double2* __restrict__ a) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2;
       i += (int64_t)gridDim.x * kBlock) {
    double2 v = b[i];
    if (kind == 1) { const double2 w = c[i]; v.x += s * w.x; v.y += s * w.y; }
    a[i] = v;
  }
}

// ---- GMRES bookkeeping on the device ----------------------------------------
// The Hessenberg column of iteration k arrives in hcol[0..k+1] (k+1 dots and
// the norm of the orthogonalised vector).  One wave applies the stored Givens
// rotations, forms the new one, updates the rotated right-hand side g and the
// residual estimate |g[k+1]| and decides on convergence - what the host did
// after a D2H copy + stream synchronise per iteration.  The host now only
// reads the 32-byte status, one iteration LATE (so that the next iteration is
// already queued while it waits): after `done` the state is frozen, later
// calls (the one over-run iteration) change nothing.
struct GmresStatus {
  double res;        // residual estimate after the last counted iteration
  int done;          // 1: converged or broke down at iteration kconv
  int kconv;         // columns of this cycle that belong to the solution
  int code;          // 0 ok, 1 non-finite entry, 2 singular column, 3 lucky breakdown
  int pad;
};

static __global__ void k_gmres_reset(double beta, int m, double* g, GmresStatus* st) {
  for (int i = threadIdx.x; i <= m; i += blockDim.x) g[i] = (i == 0) ? beta : 0.0;
  if (threadIdx.x == 0) { st->res = beta; st->done = 0; st->kconv = 0; st->code = 0; }
}

// H: column-major, leading dimension m + 1
static __global__ void k_gmres_column(int k, int m, const double* hcol, double* H,
                               double* cs, double* sn, double* g, double tol,
                               GmresStatus* st) {
  if (threadIdx.x != 0 || st->done) return;
  double* hc = H + (size_t)k * (m + 1);
  for (int j = 0; j <= k + 1; ++j) hc[j] = hcol[j];
  const double hn = hc[k + 1];
  if (!isfinite(hn)) { st->done = 1; st->code = 1; st->kconv = k; return; }
  for (int j = 0; j < k; ++j) {
    const double t = cs[j] * hc[j] + sn[j] * hc[j + 1];
    hc[j + 1] = -sn[j] * hc[j] + cs[j] * hc[j + 1];
    hc[j] = t;
  }
  const double d = hypot(hc[k], hc[k + 1]);
  if (!(d > 0.0) || !isfinite(d)) { st->done = 1; st->code = 2; st->kconv = k; return; }
  cs[k] = hc[k] / d; sn[k] = hc[k + 1] / d;
  hc[k] = d; hc[k + 1] = 0.0;
  g[k + 1] = -sn[k] * g[k]; g[k] = cs[k] * g[k];
  const double res = fabs(g[k + 1]);
  st->res = res; st->kconv = k + 1;
  if (res <= tol) st->done = 1;
  else if (hn == 0.0) { st->done = 1; st->code = 3; }
}

// y = H(0:k,0:k)^-1 g(0:k) by back substitution (one wave; k <= 255)
static __global__ void k_gmres_ysolve(int k, int m, const double* H, const double* g,
                               double* y) {
  __shared__ double ys[256];
  const int lane = threadIdx.x;
  for (int i = k - 1; i >= 0; --i) {
    double s = 0.0;
    for (int j = i + 1 + lane; j < k; j += 64) s += H[(size_t)j * (m + 1) + i] * ys[j];
    s = wave_sum(s);
    if (lane == 0) ys[i] = (g[i] - s) / H[(size_t)i * (m + 1) + i];
    __syncthreads();
  }
  for (int i = lane; i < k; i += 64) y[i] = ys[i];
}

// out = sum_j y[j] V_j
static __global__ __launch_bounds__(kBlock) void k_combine(
    int64_t n, const double* V, int64_t ld, int nvec, const double* y,
    double* out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    double s = 0.0;
    for (int j = 0; j < nvec; ++j) s += y[j] * V[(int64_t)j * ld + i];
    out[i] = s;
  }
}

}  // namespace pcd
