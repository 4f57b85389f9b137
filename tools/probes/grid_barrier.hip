// grid_barrier.hip - what does one dependent step cost on gfx950:
//   (a) a kernel node in a hipGraph chain, (b) a grid-wide barrier inside one
//   resident kernel?  Each step is the same small dependent update
//   y[i] = 0.5 * (x[(i + shift) % n] + x[i]) + 1, ping-ponged, so every step
//   reads what OTHER workgroups (other XCDs) wrote in the step before; the
//   result is checked against the host.
// build: hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void step(const double* x, double* y, int n, int shift,
                                     int gtid, int gsize) {
  for (int i = gtid; i < n; i += gsize) {
    int j = i + shift; if (j >= n) j -= n;
    y[i] = 0.5 * (x[j] + x[i]) + 1.0;
  }
}

__global__ void __launch_bounds__(256) k_step(const double* x, double* y, int n, int shift) {
  step(x, y, n, shift, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// monotone counter barrier; `bar` starts at 0, `gen` counts barriers passed
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE);   // agent scope: writes back this XCD's L2
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  // the acquire of thread 0 invalidated L2 lines of this XCD; L1/K$ of the CU too
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ void __launch_bounds__(256) k_resident(double* a, double* b, int n, int shift, int steps, unsigned* bar) {
  const int gtid = blockIdx.x * 256 + threadIdx.x, gsize = gridDim.x * 256;
  double* x = a; double* y = b;
  for (int s = 0; s < steps; ++s) {
    step(x, y, n, shift, gtid, gsize);
    grid_barrier(bar, (unsigned)(s + 1) * gridDim.x);
    double* t = x; x = y; y = t;
  }
}

// barrier only (no data), to separate the cost of the cache maintenance
__global__ void __launch_bounds__(256) k_barrier_only(int steps, unsigned* bar) {
  for (int s = 0; s < steps; ++s) grid_barrier(bar, (unsigned)(s + 1) * gridDim.x);
}

static void host_ref(std::vector<double>& x, int n, int shift, int steps) {
  std::vector<double> y(n);
  for (int s = 0; s < steps; ++s) {
    for (int i = 0; i < n; ++i) y[i] = 0.5 * (x[(i + shift) % n] + x[i]) + 1.0;
    x.swap(y);
  }
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned* bar; CK(hipMalloc(&bar, 4));
  const int steps = 40, reps = 50;
  const int sizes[] = {4096, 32768, 131072, 524288};
  const int grids[] = {32, 64, 128, 256, 512};
  printf("%-10s %-8s %-6s %10s %10s\n", "variant", "n", "wgs", "us/step", "check");
  for (int n : sizes) {
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) h[i] = std::sin(0.001 * i);
    std::vector<double> ref = h; host_ref(ref, n, 1237 % n, steps);
    double *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    for (int g : grids) {
      if ((long)g * 256 > 4L * n && g != 32) continue;
      // (a) graph of `steps` kernel nodes
      {
        CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        double* x = a; double* y = b;
        for (int s = 0; s < steps; ++s) { k_step<<<g, 256, 0, st>>>(x, y, n, 1237 % n); std::swap(x, y); }
        CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
        std::vector<double> out(n); CK(hipMemcpy(out.data(), (steps % 2) ? b : a, n * 8, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < n; ++i) err = std::fmax(err, std::fabs(out[i] - ref[i]));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %-8d %-6d %10.3f %10.2e\n", "graph", n, g, ms * 1e3 / reps / steps, err);
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(gr));
      }
      // (b) one resident kernel with grid barriers
      {
        CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(bar, 0, 4, st));
        k_resident<<<g, 256, 0, st>>>(a, b, n, 1237 % n, steps, bar);
        CK(hipStreamSynchronize(st));
        std::vector<double> out(n); CK(hipMemcpy(out.data(), (steps % 2) ? b : a, n * 8, hipMemcpyDeviceToHost));
        double err = 0; for (int i = 0; i < n; ++i) err = std::fmax(err, std::fabs(out[i] - ref[i]));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) {
          CK(hipMemsetAsync(bar, 0, 4, st));
          k_resident<<<g, 256, 0, st>>>(a, b, n, 1237 % n, steps, bar);
        }
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-10s %-8d %-6d %10.3f %10.2e\n", "resident", n, g, ms * 1e3 / reps / steps, err);
      }
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  for (int g : grids) {
    CK(hipMemsetAsync(bar, 0, 4, st));
    k_barrier_only<<<g, 256, 0, st>>>(steps, bar); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) { CK(hipMemsetAsync(bar, 0, 4, st)); k_barrier_only<<<g, 256, 0, st>>>(steps, bar); }
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s %-8d %-6d %10.3f %10s\n", "barrier", 0, g, ms * 1e3 / reps / steps, "-");
  }
  return 0;
}
