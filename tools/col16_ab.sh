#!/bin/bash
# A/B: 16-bit column offsets (PCD_NO_COL16=1 switches them off)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for W in "6" "7" "3 cube"; do
  for i in 1 2; do
    python3 tools/time_a00_kernel.py $W
    PCD_NO_COL16=1 python3 tools/time_a00_kernel.py $W
  done
done
