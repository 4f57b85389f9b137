// bw_sweep.hip - what READ bandwidth can a streaming kernel reach on this
// MI355X, from HBM (1 GiB) and from the Infinity Cache (64 / 96 MiB working
// sets)?  Sweeps access shape (grid-stride vs one contiguous chunk per
// workgroup), bytes per lane-load (8 / 16), independent loads in flight per lane
// and workgroups per CU.  The roof next to which the dominant kernel of the
// PCD engine is priced (bench.py: measured_probes_gbs).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/bw_sweep.hip -o /tmp/bw && /tmp/bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));   // 16 bytes per lane-load
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <class T, int U, bool BLOCKED, bool NT>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ a, long n, T* out) {
  T acc[U];
  for (int u = 0; u < U; ++u) acc[u] = T(0.0);
  long i, end, step;
  if (BLOCKED) {           // one contiguous chunk per workgroup
    const long per = (n + gridDim.x - 1) / gridDim.x;
    i = (long)blockIdx.x * per + threadIdx.x;
    end = min(n, (long)(blockIdx.x + 1) * per);
    step = 256;
  } else {
    i = (long)blockIdx.x * 256 + threadIdx.x;
    end = n;
    step = (long)gridDim.x * 256;
  }
  for (; i + (U - 1) * step < end; i += U * step) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * step) : a[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += v[u];
  }
  for (; i < end; i += step) acc[0] += a[i];
  T s = acc[0];
  for (int u = 1; u < U; ++u) s += acc[u];
  out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

template <class T, int U, bool BLOCKED, bool NT>
double run(const void* a, size_t bytes, void* out, int grid, int reps) {
  const long n = bytes / sizeof(T);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < reps + 2; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_read<T, U, BLOCKED, NT>), dim3(grid), dim3(256), 0, 0, (const T*)a, n, (T*)out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return bytes / (best * 1e-3) / 1e12;
}

int main() {
  const size_t big = 1ull << 30;
  void *a, *out;
  CK(hipMalloc(&a, big)); CK(hipMalloc(&out, 64 << 20));
  CK(hipMemset(a, 0, big));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("# %s, %d CUs; TB/s read, best of 10\n", p.name, cus);
  printf("# bytes  shape  B/lane  inflight  nt  wg/cu  TB/s\n");
  for (size_t bytes : {(size_t)64 << 20, (size_t)96 << 20, (size_t)1 << 30}) {
    for (int wg : {1, 2, 4, 8, 16, 32}) {
      const int grid = cus * wg;
#define ROW(T, U, B, NT) printf("%5zu MiB  %s  %2zu  %2d  %d  %2d  %.2f\n", bytes >> 20, B ? "blocked" : "strided", sizeof(T), U, (int)NT, wg, run<T, U, B, NT>(a, bytes, out, grid, 10));
      ROW(d2, 4, false, false) ROW(d2, 8, false, false) ROW(d2, 16, false, false)
      ROW(d2, 4, true, false) ROW(d2, 8, true, false) ROW(d2, 16, true, false)
      ROW(double, 8, false, false) ROW(double, 8, true, false)
      ROW(d2, 8, false, true) ROW(d2, 8, true, true)
    }
  }
  return 0;
}
