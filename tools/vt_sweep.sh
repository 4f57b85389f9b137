#!/bin/bash
# The vector-tile kernels against the gather kernels on the finest A00 of a
# workload: rows per block of the two-component kernels (PCD_VT_ROWS2), direct
# vs staged non-temporal form (PCD_VT_NT), entries in flight (-DPCD_VT_U).
#   tools/vt_sweep.sh "<level> [cube <n0>]" ...   e.g.  tools/vt_sweep.sh "6" "7" "3 cube 4" "3 cube 6"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for WL in "$@"; do
  PCD_VEC_TILE=0 python3 tools/time_a00_kernel.py $WL
  for R2 in 64 128; do
    for NT in 1 0; do
      echo -n "rows2 $R2: "
      PCD_VT_ROWS2=$R2 PCD_VEC_TILE=2 PCD_VT_NT=$NT python3 tools/time_a00_kernel.py $WL
    done
  done
done
