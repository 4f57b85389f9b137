#!/bin/bash
# level 7 (HBM-streaming): XCD-aware row-block mapping on/off x non-temporal on/off
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
out=gpurun_out/r03_x_xcd_nt_level7.txt; : > $out
python tools/time_a00_kernel.py 7 >> $out 2>&1
PCD_XCD_REMAP_MAX_ROWS=100000000 python tools/time_a00_kernel.py 7 >> $out 2>&1
PCD_NT_BYTES=-1 python tools/time_a00_kernel.py 7 >> $out 2>&1
PCD_NT_BYTES=-1 PCD_XCD_REMAP_MAX_ROWS=100000000 python tools/time_a00_kernel.py 7 >> $out 2>&1
grep "us per launch" $out
