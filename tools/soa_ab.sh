#!/bin/bash
# A/B: LDS layout of the three-component products (planes vs 24-byte records)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/pcdlibs
tools/build_hip.sh /tmp/pcdlibs/aos.so -DPCD_LDS_SOA=0
for i in 1 2; do
  python3 tools/time_a00_kernel.py 3 cube
  FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/aos.so python3 tools/time_a00_kernel.py 3 cube
done
