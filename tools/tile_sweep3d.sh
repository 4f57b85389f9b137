#!/bin/bash
# finer sweep of the triple tile (3-D multi-component kernels), cube N = 32
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/pcdlibs
python3 tools/time_a00_kernel.py 3 cube
for T3 in 1280 1408 1536 1664 1792 2048; do
  tools/build_hip.sh /tmp/pcdlibs/t3_$T3.so -DPCD_TILE3=$T3
  for CH in 2 3; do
    FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/t3_$T3.so PCD_MAX_RB=64 PCD_MAX_CHUNKS=$CH python3 tools/time_a00_kernel.py 3 cube
    FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/t3_$T3.so PCD_MAX_RB=64 PCD_MAX_CHUNKS=$CH python3 tools/time_a00_kernel.py 3 cube
  done
done
