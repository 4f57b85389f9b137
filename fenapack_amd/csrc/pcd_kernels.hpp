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
template <int LPR>
__device__ __forceinline__ double row_dot(const int* __restrict__ rowptr,
                                          const int* __restrict__ col,
                                          const double* __restrict__ val,
                                          const double* x, int row, int nrows,
                                          int lane) {
  double s = 0.0;
  if (row < nrows) {
    const int b = rowptr[row], e = rowptr[row + 1];
    for (int k = b + lane; k < e; k += LPR) s += val[k] * x[col[k]];
  }
#pragma unroll
  for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_down(s, off, LPR);
  return s;
}

// MODE 0: y = A x      MODE 1: y = add + A x      MODE 2: y = add - A x
template <int LPR, int MODE>
__global__ __launch_bounds__(kBlock) void k_spmv(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x, const double* add,
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
    double c0, double c1, double c2) {
  constexpr int RPB = kBlock / LPR;
  const int lane = threadIdx.x % LPR;
  const int nloop = (nrows + RPB - 1) / RPB * RPB;
  for (int row = blockIdx.x * RPB + threadIdx.x / LPR; row < nloop;
       row += gridDim.x * RPB) {
    double s = row_dot<LPR>(rowptr, col, val, pk, row, nrows, lane);
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
__global__ __launch_bounds__(kBlock) void k_scale_dinv(
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
};

// x = 0, r = b, z = dinv r, p = z; parts_rz[blk] = sum r.z
__global__ __launch_bounds__(kBlock) void k_cg_init(
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
__global__ __launch_bounds__(kBlock) void k_cg_pupdate(
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
    double* parts_pq, const CgState* st) {
  __shared__ double sm[4];
  if (st->done) return;
  constexpr int RPB = kBlock / LPR;
  const int lane = threadIdx.x % LPR;
  const int nloop = (nrows + RPB - 1) / RPB * RPB;
  double acc = 0.0;
  for (int row = blockIdx.x * RPB + threadIdx.x / LPR; row < nloop;
       row += gridDim.x * RPB) {
    double s = row_dot<LPR>(rowptr, col, val, p, row, nrows, lane);
    if (lane == 0 && row < nrows) { q[row] = s; acc += p[row] * s; }
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) parts_pq[blockIdx.x] = acc;
}

// alpha = rz / pq; x += alpha p; r -= alpha q; z = dinv r; parts_out = r.z
__global__ __launch_bounds__(kBlock) void k_cg_update(
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

// ---- BLAS-1 glue of the apply bodies --------------------------------------
__global__ __launch_bounds__(kBlock) void k_copy(int n, const double* x,
                                                  double* y) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    y[i] = x[i];
}

// y = a*x + b*y  (b == 0 never reads y)
__global__ __launch_bounds__(kBlock) void k_axpby(int n, double a,
                                                   const double* x, double b,
                                                   double* y) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

// SubfieldBC::apply: x[idx[i]] = val[i]  (VecSetValues INSERT)
__global__ __launch_bounds__(kBlock) void k_bc_set(
    int n_bc, const int* __restrict__ idx, const double* __restrict__ val,
    double* x) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n_bc) x[idx[i]] = val[i];
}

// fieldsplit scatter: out[i] = in[perm[i]]  /  out[perm[i]] = in[i]
__global__ __launch_bounds__(kBlock) void k_gather(
    int n, const int* __restrict__ perm, const double* in, double* out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    out[i] = in[perm[i]];
}
__global__ __launch_bounds__(kBlock) void k_scatter(
    int n, const int* __restrict__ perm, const double* in, double* out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n;
       i += gridDim.x * kBlock)
    out[perm[i]] = in[i];
}

// block values from the caller's monolithic value array
__global__ __launch_bounds__(kBlock) void k_gather_vals(
    int64_t nnz, const int64_t* __restrict__ src, const double* vals,
    double* out) {
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz;
       k += (int64_t)gridDim.x * kBlock)
    out[k] = vals[src[k]];
}

// reciprocal diagonal (1 where the diagonal is zero or absent)
__global__ __launch_bounds__(kBlock) void k_dinv(
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
__global__ __launch_bounds__(kBlock) void k_mdot(
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
__global__ __launch_bounds__(kBlock) void k_mdot_reduce(const double* parts,
                                                         int G, double* h) {
  __shared__ double sm[4];
  const double s = reduce_parts(parts + (int64_t)blockIdx.x * G, G, sm);
  if (threadIdx.x == 0) h[blockIdx.x] = s;
}

// w -= sum_j h[j] V_j ; parts[blk] = sum w^2
__global__ __launch_bounds__(kBlock) void k_maxpy_norm(
    int64_t n, const double* V, int64_t ld, int nvec, const double* h,
    double* w, double sign, double* parts) {
  __shared__ double sm[4];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    double wi = w[i];
    for (int j = 0; j < nvec; ++j) wi += sign * h[j] * V[(int64_t)j * ld + i];
    w[i] = wi;
    acc += wi * wi;
  }
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) parts[blockIdx.x] = acc;
}

// nrm = sqrt(sum parts); w /= nrm; *out_nrm = nrm
__global__ __launch_bounds__(kBlock) void k_normalize(
    int64_t n, double* w, const double* parts, int nparts, double* out_nrm) {
  __shared__ double sm[4];
  const double nrm = sqrt(reduce_parts(parts, nparts, sm));
  const double inv = (nrm != 0.0) ? 1.0 / nrm : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock)
    w[i] *= inv;
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_nrm = nrm;
}

// out = sum_j y[j] V_j
__global__ __launch_bounds__(kBlock) void k_combine(
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
