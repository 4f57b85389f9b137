#!/bin/bash
# A/B of the LDS tile of the multi-component stream kernels (compile-time) x
# rows per workgroup (run time) on the finest A00: 2-D cavity level 6 and the
# 3-D cube N = 32.  Builds the variants on the box (hipcc, ~15 s each).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/pcdlibs
build() { tools/build_hip.sh /tmp/pcdlibs/$2.so $1; }
python3 tools/time_a00_kernel.py 6
python3 tools/time_a00_kernel.py 3 cube
for T2 in 1024 1536 3072; do
  build "-DPCD_TILE2=$T2" t2_$T2
  for RB in 256 64; do
    FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/t2_$T2.so PCD_MAX_RB=$RB python3 tools/time_a00_kernel.py 6
  done
done
for T3 in 1024 1344 1536; do
  build "-DPCD_TILE3=$T3" t3_$T3
  for RB in 64 32; do
    for CH in 1 2; do
      FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/t3_$T3.so PCD_MAX_RB=$RB PCD_MAX_CHUNKS=$CH python3 tools/time_a00_kernel.py 3 cube
    done
  done
done
