#!/bin/bash
# compile-time variants of the lane-major tile kernels, us per launch of the
# Chebyshev step on the finest A00 (first line per workload: the tree's default build).
#   tools/lm_variants.sh "<flags>;<flags>;..." "<workload>" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/pcdlibs
IFS=';' read -ra VARS <<< "$1"
shift
i=0
for V in "${VARS[@]}"; do
  tools/build_hip.sh /tmp/pcdlibs/v$i.so $V &
  i=$((i+1))
done
wait
for WL in "$@"; do
  python3 tools/time_a00_kernel.py $WL
  i=0
  for V in "${VARS[@]}"; do
    echo -n "[$V] "
    FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/v$i.so python3 tools/time_a00_kernel.py $WL
    echo -n "[$V] "
    FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/v$i.so python3 tools/time_a00_kernel.py $WL
    i=$((i+1))
  done
done
