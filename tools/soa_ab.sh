#!/bin/bash
# A/B: LDS layout of the three-component products (planes vs 24-byte records)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
SRC=fenapack_amd/csrc/pcd_engine.hip
mkdir -p /tmp/pcdlibs
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPCD_LDS_SOA=0 -o /tmp/pcdlibs/aos.so $SRC
for i in 1 2; do
  python3 tools/time_a00_kernel.py 3 cube
  FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/aos.so python3 tools/time_a00_kernel.py 3 cube
done
