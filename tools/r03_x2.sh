#!/bin/bash
# cube N = 64 (non-temporal three-component kernel): XCD-aware mapping on / off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
export FENAPACK_AMD_MAX_CELLS=4000000
out=gpurun_out/r03_x_xcd_nt_cube64.txt; : > $out
for v in 3 1 3 1; do PCD_XCD_REMAP_NT=$v python tools/time_a00_kernel.py 4 cube >> $out 2>&1; done
grep "us per launch" $out
