#!/bin/bash
# Lane-major tile kernels (k_*_lm) against the form that stages the entries in
# LDS (PCD_VT_LM=0) on the finest A00 of a workload, us per launch of the
# Chebyshev step.   tools/lm_ab.sh "7" "3 cube 6" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for WL in "$@"; do
  for LM in 0 1 0 1; do
    PCD_VT_LM=$LM python3 tools/time_a00_kernel.py $WL
  done
done
