#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
FENAPACK_AMD_MAX_CELLS=2000000 OPENBLAS_NUM_THREADS=8 timeout 600 python tools/setup_breakdown.py --geometry cube --level 4 --n0 4 --steps 2 --profile > gpurun_out/r03_setup_n64_profile.txt 2>&1
grep -E "^[a-z].* s$|set-up total|^  [a-z_]+ +[0-9]+ +[0-9.]+$" gpurun_out/r03_setup_n64_profile.txt
