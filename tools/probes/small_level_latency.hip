// small_level_latency.hip - what does one dependent memory round trip cost a
// SMALL CSR-stream kernel on gfx950?  A chain of dependent Jacobi-like steps
// x <- D^-1 (b - A x) on a banded matrix of `n` rows runs as kernel nodes of a
// hipGraph, with the row-block kernel in two variants:
//   csr : block start / end read from rowptr (rowptr -> col/val -> x: three
//         dependent round trips, as pcd_kernels.hpp: stream_row_block)
//   pad : every row block owns a fixed-capacity slice (entries padded with
//         col = -1): col/val addresses follow from blockIdx alone (two trips)
// "warm": every step reads the same arrays (they stay in the L2s); "cold": step
// s reads its own copy of the matrix / diagonal / right-hand side, 40 copies
// in all - more than the L2s hold, like the operators of one PCApply, each of
// which is read once per apply.
// build: hipcc -O3 --offload-arch=gfx950 small_level_latency.hip -o small_level_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <int RB, bool PAD>
__global__ __launch_bounds__(kBlock) void k_step(int nrows, const int* __restrict__ rowptr,
                                                 const int* __restrict__ col, const double* __restrict__ val,
                                                 int cap, const double* __restrict__ dinv,
                                                 const double* __restrict__ b, const double* x, double* y) {
  __shared__ double lds[4096];
  const int r0 = blockIdx.x * RB, r1 = min(r0 + RB, nrows);
  constexpr int TPR = kBlock / RB;
  const int row = r0 + threadIdx.x / TPR, sub = threadIdx.x % TPR;
  const bool mine = row < r1;
  // PAD: the slice of this block starts at blockIdx * cap, whatever rowptr says
  const int base0 = rowptr[r0];
  const int k0 = PAD ? blockIdx.x * cap : base0;
  const int k1 = PAD ? k0 + cap : rowptr[r1];
  const int ra = mine ? rowptr[row] - base0 : 0, rb = mine ? rowptr[row + 1] - base0 : 0;
  double bi = 0.0, d = 0.0;
  if (mine && sub == 0) { bi = b[row]; d = dinv[row]; }
  for (int base = k0; base < k1; base += kUnroll * kBlock) {
    int c[kUnroll]; double v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int k = base + u * kBlock + threadIdx.x;
      const bool in = k < k1;
      c[u] = in ? col[k] : -1;
      v[u] = in ? val[k] : 0.0;
    }
    double xv[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) xv[u] = c[u] >= 0 ? x[c[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int k = base + u * kBlock + threadIdx.x;
      if (k < k1) lds[k - k0] = v[u] * xv[u];
    }
  }
  __syncthreads();
  double s = 0.0;
  for (int j = ra + sub; j < rb; j += TPR) s += lds[j];
#pragma unroll
  for (int m = TPR / 2; m > 0; m >>= 1) s += __shfl_xor(s, m);
  if (mine && sub == 0) y[row] = d * (bi - s) + x[row] * 0.0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int steps = 40, reps = 50;
  constexpr int RB = 32;
  printf("%-4s %-6s %-8s %-6s %10s %10s\n", "l2", "fmt", "rows", "nnz/r", "us/step", "max|diff|");
  for (int n : {512, 1792, 6656, 26624}) {
    for (int w : {12, 40}) {
      // banded matrix, w entries per row (fewer at the ends)
      std::vector<int> rp(n + 1, 0), ci; std::vector<double> va, dinv(n), b(n), x0(n);
      for (int i = 0; i < n; ++i) {
        for (int o = -w / 2; o < w - w / 2; ++o) {
          const int j = i + o * 3;
          if (j < 0 || j >= n) continue;
          ci.push_back(j); va.push_back(o == 0 ? 4.0 + w : -1.0 / (1 + std::abs(o)));
        }
        rp[i + 1] = (int)ci.size();
        dinv[i] = 1.0 / (4.0 + w); b[i] = std::sin(0.01 * i); x0[i] = 0.0;
      }
      const int nb = (n + RB - 1) / RB;
      int cap = 0;
      for (int bb = 0; bb < nb; ++bb) cap = std::max(cap, rp[std::min(n, (bb + 1) * RB)] - rp[bb * RB]);
      cap = (cap + 255) / 256 * 256;
      std::vector<int> pc((size_t)nb * cap, -1); std::vector<double> pv((size_t)nb * cap, 0.0);
      for (int bb = 0; bb < nb; ++bb) {
        const int a = rp[bb * RB], e = rp[std::min(n, (bb + 1) * RB)];
        for (int k = a; k < e; ++k) { pc[(size_t)bb * cap + k - a] = ci[k]; pv[(size_t)bb * cap + k - a] = va[k]; }
      }
      int *drp, *dci, *dpc; double *dva, *dpv, *dd, *db, *dx, *dy;
      const size_t K = steps;          // copies
      const size_t nz = ci.size(), pz = pc.size();
      CK(hipMalloc(&drp, K * (n + 1) * 4)); CK(hipMalloc(&dci, K * nz * 4)); CK(hipMalloc(&dva, K * nz * 8));
      CK(hipMalloc(&dpc, K * pz * 4)); CK(hipMalloc(&dpv, K * pz * 8));
      CK(hipMalloc(&dd, K * n * 8)); CK(hipMalloc(&db, K * n * 8)); CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&dy, n * 8));
      for (size_t k = 0; k < K; ++k) {
        CK(hipMemcpy(drp + k * (n + 1), rp.data(), (n + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dci + k * nz, ci.data(), nz * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dva + k * nz, va.data(), nz * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dpc + k * pz, pc.data(), pz * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dpv + k * pz, pv.data(), pz * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(dd + k * n, dinv.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(db + k * n, b.data(), n * 8, hipMemcpyHostToDevice));
      }
      std::vector<double> res[2];
      for (int cold = 0; cold < 2; ++cold)
      for (int pad = 0; pad < 2; ++pad) {
        CK(hipMemcpy(dx, x0.data(), n * 8, hipMemcpyHostToDevice));
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        double *xa = dx, *xb = dy;
        for (int s = 0; s < steps; ++s) {
          const size_t k = cold ? s : 0;
          if (pad) k_step<RB, true><<<nb, kBlock, 0, st>>>(n, drp + k * (n + 1), dpc + k * pz, dpv + k * pz, cap, dd + k * n, db + k * n, xa, xb);
          else k_step<RB, false><<<nb, kBlock, 0, st>>>(n, drp + k * (n + 1), dci + k * nz, dva + k * nz, cap, dd + k * n, db + k * n, xa, xb);
          std::swap(xa, xb);
        }
        CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
        res[pad].resize(n);
        CK(hipMemcpy(res[pad].data(), (steps % 2) ? dy : dx, n * 8, hipMemcpyDeviceToHost));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double diff = 0;
        if (pad) for (int i = 0; i < n; ++i) diff = std::fmax(diff, std::fabs(res[0][i] - res[1][i]));
        printf("%-4s %-6s %-8d %-6d %10.3f %10.2e\n", cold ? "cold" : "warm", pad ? "pad" : "csr", n, w, ms * 1e3 / reps / steps, diff);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
      }
      hipFree(drp); hipFree(dci); hipFree(dva); hipFree(dpc); hipFree(dpv); hipFree(dd); hipFree(db); hipFree(dx); hipFree(dy);
    }
  }
  return 0;
}
