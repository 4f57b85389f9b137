// pcd_launch.hpp - every kernel launch of the engine is counted (per host
// thread: one rank = one thread).  PCD_INFO_LAUNCHES reads the counter: the
// launches of one PCApply on R ranks against one GPU is how the cost of the
// partitioned path is stated (DESIGN.md "Several ranks").
#pragma once
#include <hip/hip_runtime.h>

namespace pcd {
inline long long& launch_count() {
  static thread_local long long n = 0;
  return n;
}
}  // namespace pcd

#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...) \
  do {                                                                                   \
    ++pcd::launch_count();                                                               \
    kernelName<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);   \
  } while (0)
