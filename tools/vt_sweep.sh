#!/bin/bash
# Compile-time sweep of the vector-tile kernels (rows per block, entries in
# flight per lane) against the gather kernels, finest A00 of the unit cube.
#   tools/vt_sweep.sh "<level> cube <n0>" ...   e.g.  tools/vt_sweep.sh "3 cube 4" "3 cube 6"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
SRC=fenapack_amd/csrc/pcd_engine.hip
mkdir -p /tmp/pcdlibs
for WL in "$@"; do
  PCD_VEC_TILE=0 python3 tools/time_a00_kernel.py $WL
  for CFG in "64 8" "32 4" "32 8" "64 4"; do
    set -- $CFG
    LIB=/tmp/pcdlibs/vt_$1_$2.so
    [ -f $LIB ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPCD_VT_ROWS=$1 -DPCD_VT_U=$2 -o $LIB $SRC
    for NT in 0 1; do
      echo -n "rows $1 U $2: "
      FENAPACK_AMD_HIP_LIB=$LIB PCD_VEC_TILE=2 PCD_VT_NT=$NT python3 tools/time_a00_kernel.py $WL
    done
  done
done
