// short_row_prolongation.hip - is the CSR-stream kernel the right form for the
// prolongation-add of a smoothed-aggregation level (x_fine += P e_coarse, three
// components per node, ~4.6 entries per row)?
//
// In one PCApply of config 5's own mesh (cube N = 73, -pc_type gamg) that launch
// is pcd::k_spmv_sc<128, 1, 3, true>: 371 MB in 116-122 us = 0.40 of 8 TB/s
// (profiles/r06_zzz_n73_pmc_roofline.json), the lowest fraction among the
// launches above 100 us.  The stream kernel stages PRODUCTS in LDS and pays two
// workgroup barriers per 128 rows; with rows this short a block moves ~13 KB
// between barriers.  The probe builds a synthetic P of the same shape (147^3
// fine nodes, 3x3x3 aggregates, tensor-product smoothing: (5/3)^3 = 4.63 entries
// per row) and times, on the same arrays:
//   stream : the product's kernel, as launched by pcd_apply.hip
//   tpr    : one lane per row, entries read straight from global memory
//   tpr8   : the same with the first 8 entries' loads issued before any use
//   tprnt  : tpr8 with non-temporal loads of col / val
//   ..., coalesced epilogue: `add` read and `y` written as the block's
//            consecutive doubles through LDS instead of strided triples
// "hot": launches back to back; "cold": 1 GB written between launches (the
// state the cycle leaves: the finest level's passes have flushed the caches).
// build: hipcc -O3 --offload-arch=gfx950 -std=c++17 short_row_prolongation.hip
#include "../../fenapack_amd/csrc/pcd_kernels.hpp"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using namespace pcd;

template <int NC, int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_prolong_tpr(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x_, const double* add_, double* y_) {
  const VecC<NC>* x = vc<NC>(x_);
  const VecC<NC>* add = vc<NC>(add_);
  VecC<NC>* y = vc<NC>(y_);
  // contiguous stretches of rows per XCD (grid is a multiple of 8)
  const int G = gridDim.x;
  const int slot = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
  const int row = slot * kBlock + threadIdx.x;
  if (row >= nrows) return;
  const int b = rowptr[row], e = rowptr[row + 1];
  VecC<NC> s = add[row];
  int k = b;
  if constexpr (U > 1) {
    int c[U]; double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = k + u < e;
      c[u] = in ? stream_load<NT>(col + k + u) : -1;
      v[u] = in ? stream_load<NT>(val + k + u) : 0.0;
    }
    VecC<NC> xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : vzero<NC>();
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += v[u] * xv[u].c[i];
    k += U;
  }
  for (; k < e; ++k) {
    const int c = col[k];
    const double v = val[k];
    const VecC<NC> xv = x[c];
#pragma unroll
    for (int i = 0; i < NC; ++i) s.c[i] += v * xv.c[i];
  }
  y[row] = s;
}

// the product's stream kernel with a COALESCED epilogue: `add` read and `y`
// written as the block's RB * NC consecutive doubles (the row sums pass through
// LDS) instead of NC strided 8-byte accesses on every TPR-th lane
template <int RB, int NC, bool NT>
__global__ __launch_bounds__(kBlock) void k_spmv_sc_co(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x, const double* ghost,
    int nloc, const double* add_, double* y_) {
  __shared__ VecC<NC> lds[tile_c<NC>()];
  __shared__ double outb[RB * NC];
  const XVecC<NC> xf{vc<NC>(x), vc<NC>(ghost), nloc};
  const int nrb = (nrows + RB - 1) / RB;
  int rb0, rb1;
  row_block_range(nrb, RB, rb0, rb1, xcd_remap_always<NC, NT>(nrows));
  constexpr int P = (RB * NC + kBlock - 1) / kBlock;
  for (int rb = rb0; rb < rb1; ++rb) {
    const int r0 = rb * RB;
    const int nb = min(RB, nrows - r0) * NC;
    const int lr = threadIdx.x / (kBlock / RB);
    const bool mine = threadIdx.x % (kBlock / RB) == 0 && r0 + lr < nrows;
    double a[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int t = threadIdx.x + q * kBlock;
      a[q] = t < nb ? add_[(size_t)r0 * NC + t] : 0.0;
    }
    const VecC<NC> s = stream_row_block_c<RB, NC, NT>(rowptr, col, val, xf, r0, nrows, lds);
    if (mine) {
#pragma unroll
      for (int i = 0; i < NC; ++i) outb[lr * NC + i] = s.c[i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const int t = threadIdx.x + q * kBlock;
      if (t < nb) y_[(size_t)r0 * NC + t] = a[q] + outb[t];
    }
    __syncthreads();
  }
}

// one lane per row with the same coalesced epilogue
template <int NC, int U>
__global__ __launch_bounds__(kBlock) void k_prolong_tpr_co(
    int nrows, const int* __restrict__ rowptr, const int* __restrict__ col,
    const double* __restrict__ val, const double* x_, const double* add_, double* y_) {
  __shared__ double outb[kBlock * NC];
  const VecC<NC>* x = vc<NC>(x_);
  const int G = gridDim.x;
  const int slot = (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8;
  const int r0 = slot * kBlock;
  if (r0 >= nrows) return;
  const int nb = min(kBlock, nrows - r0) * NC;
  double a[NC];
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const int t = threadIdx.x + q * kBlock;
    a[q] = t < nb ? add_[(size_t)r0 * NC + t] : 0.0;
  }
  const int row = r0 + threadIdx.x;
  VecC<NC> s = vzero<NC>();
  if (row < nrows) {
    const int b = rowptr[row], e = rowptr[row + 1];
    int k = b;
    int c[U]; double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = k + u < e;
      c[u] = in ? col[k + u] : -1;
      v[u] = in ? val[k + u] : 0.0;
    }
    VecC<NC> xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : vzero<NC>();
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += v[u] * xv[u].c[i];
    for (k += U; k < e; ++k) {
      const int cc = col[k];
      const double vv = val[k];
      const VecC<NC> xx = x[cc];
#pragma unroll
      for (int i = 0; i < NC; ++i) s.c[i] += vv * xx.c[i];
    }
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) outb[threadIdx.x * NC + i] = s.c[i];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NC; ++q) {
    const int t = threadIdx.x + q * kBlock;
    if (t < nb) y_[(size_t)r0 * NC + t] = a[q] + outb[t];
  }
}

static __global__ void k_fill(size_t n, double* p, double v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main(int argc, char** argv) {
  const int NF = argc > 1 ? atoi(argv[1]) : 147, AG = 3;
  const int NCo = (NF + AG - 1) / AG;
  const int64_t nf = (int64_t)NF * NF * NF, nc = (int64_t)NCo * NCo * NCo;
  std::vector<int> rp(nf + 1, 0), col;
  std::vector<double> val;
  col.reserve(nf * 5); val.reserve(nf * 5);
  srand(7);
  auto axis = [&](int i, int* out) {
    const int a = i / AG, m = i % AG;
    int n = 0;
    if (m == 0 && a > 0) out[n++] = a - 1;
    out[n++] = a;
    if (m == AG - 1 && a + 1 < NCo) out[n++] = a + 1;
    return n;
  };
  for (int k = 0; k < NF; ++k) for (int j = 0; j < NF; ++j) for (int i = 0; i < NF; ++i) {
    int ai[2], aj[2], ak[2];
    const int ni = axis(i, ai), nj = axis(j, aj), nk = axis(k, ak);
    for (int c = 0; c < nk; ++c) for (int b = 0; b < nj; ++b) for (int a = 0; a < ni; ++a) {
      col.push_back((ak[c] * NCo + aj[b]) * NCo + ai[a]);
      val.push_back(0.1 + (rand() % 1000) * 1e-3);
    }
    const int64_t r = ((int64_t)k * NF + j) * NF + i;
    rp[r + 1] = (int)col.size();
  }
  const int64_t nnz = col.size();
  printf("fine nodes %lld  coarse nodes %lld  entries %lld (%.2f per row)\n", (long long)nf, (long long)nc,
         (long long)nnz, (double)nnz / nf);
  const double bytes = 12.0 * nnz + 4.0 * (nf + 1) + 2 * 24.0 * nf + 24.0 * nc;
  printf("bytes by construction per launch: %.1f MB\n", bytes / 1e6);
  std::vector<double> x(3 * nc), add(3 * nf);
  for (auto& v : x) v = (rand() % 2000 - 1000) * 1e-3;
  for (auto& v : add) v = (rand() % 2000 - 1000) * 1e-3;
  int *d_rp, *d_col; double *d_val, *d_x, *d_add, *d_y, *d_y2, *d_flush;
  CK(hipMalloc(&d_rp, (nf + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * 8));
  CK(hipMalloc(&d_x, 3 * nc * 8)); CK(hipMalloc(&d_add, 3 * nf * 8));
  CK(hipMalloc(&d_y, 3 * nf * 8)); CK(hipMalloc(&d_y2, 3 * nf * 8));
  const size_t nflush = (size_t)1 << 27;                    // 1 GB of doubles
  CK(hipMalloc(&d_flush, nflush * 8));
  CK(hipMemcpy(d_rp, rp.data(), (nf + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_x, x.data(), 3 * nc * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_add, add.data(), 3 * nf * 8, hipMemcpyHostToDevice));
  constexpr int RB = 128;
  const int g_stream = (int)((((nf + RB - 1) / RB) + 7) / 8 * 8);
  const int g_tpr = (int)((((nf + kBlock - 1) / kBlock) + 7) / 8 * 8);
  auto run = [&](int which, double* y) {
    switch (which) {
      case 0: hipLaunchKernelGGL((k_spmv_sc<RB, 1, 3, true>), dim3(g_stream), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col,
                                 d_val, d_x, (const double*)nullptr, (int)nc, d_add, y); break;
      case 1: hipLaunchKernelGGL((k_prolong_tpr<3, 1, false>), dim3(g_tpr), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col, d_val, d_x, d_add, y); break;
      case 2: hipLaunchKernelGGL((k_prolong_tpr<3, 8, false>), dim3(g_tpr), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col, d_val, d_x, d_add, y); break;
      case 3: hipLaunchKernelGGL((k_prolong_tpr<3, 8, true>), dim3(g_tpr), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col, d_val, d_x, d_add, y); break;
      case 4: hipLaunchKernelGGL((k_prolong_tpr<3, 4, true>), dim3(g_tpr), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col, d_val, d_x, d_add, y); break;
      case 5: hipLaunchKernelGGL((k_spmv_sc_co<RB, 3, true>), dim3(g_stream), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col,
                                 d_val, d_x, (const double*)nullptr, (int)nc, d_add, y); break;
      case 6: hipLaunchKernelGGL((k_prolong_tpr_co<3, 8>), dim3(g_tpr), dim3(kBlock), 0, 0, (int)nf, d_rp, d_col, d_val, d_x, d_add, y); break;
    }
  };
  const char* names[] = {"stream k_spmv_sc<128,1,3,NT>", "tpr", "tpr8", "tpr8 nt", "tpr4 nt", "stream, coalesced epilogue", "tpr8, coalesced epilogue"};
  const int NV = 7;
  // results agree
  run(0, d_y); CK(hipDeviceSynchronize());
  std::vector<double> y0(3 * nf), y1(3 * nf);
  CK(hipMemcpy(y0.data(), d_y, 3 * nf * 8, hipMemcpyDeviceToHost));
  for (int w = 1; w < NV; ++w) {
    run(w, d_y2); CK(hipDeviceSynchronize());
    CK(hipMemcpy(y1.data(), d_y2, 3 * nf * 8, hipMemcpyDeviceToHost));
    double m = 0, s = 0;
    for (size_t i = 0; i < y0.size(); ++i) { m = std::max(m, std::fabs(y0[i] - y1[i])); s = std::max(s, std::fabs(y0[i])); }
    printf("%-30s max |diff| / max |y| against the stream kernel: %.2e\n", names[w], m / s);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int cold = 0; cold < 2; ++cold) {
    for (int w = 0; w < NV; ++w) {
      const int reps = 20;
      std::vector<float> t(reps);
      for (int r = 0; r < reps + 3; ++r) {
        if (cold) hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, nflush, d_flush, (double)r);
        CK(hipEventRecord(e0, 0));
        run(w, d_y2);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 3) t[r - 3] = ms;
      }
      std::sort(t.begin(), t.end());
      const double us = t[reps / 2] * 1e3;
      printf("%-5s %-30s median %7.1f us  (min %7.1f)  %.2f TB/s by construction = %.2f of 8 TB/s\n", cold ? "cold" : "hot",
             names[w], us, t[0] * 1e3, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0);
    }
  }
  return 0;
}
