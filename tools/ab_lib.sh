#!/bin/bash
# Same-box A/B of two builds of libpcd_hip.so on the fused Chebyshev step of
# the finest A00 (tools/time_a00_kernel.py), alternating A B A B per workload:
#   tools/ab_lib.sh <base.so> <new.so> "<workload>" ...
# (a baseline build of HEAD's sources: git archive HEAD fenapack_amd/csrc
# include tools/build_hip.sh into a scratch tree, then build_hip.sh there)
A=$1; B=$2; shift 2
for WL in "$@"; do
  for rep in 1 2; do
    echo -n "[base] "; FENAPACK_AMD_HIP_LIB=$A python3 tools/time_a00_kernel.py $WL
    echo -n "[new ] "; FENAPACK_AMD_HIP_LIB=$B python3 tools/time_a00_kernel.py $WL
  done
done
