#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
FENAPACK_AMD_MAX_CELLS=3000000 OPENBLAS_NUM_THREADS=8 timeout 900 python tools/setup_breakdown.py --geometry cube --level 0 --n0 ${1:-73} --algebraic --steps 2 --profile > gpurun_out/r03_setup_n${1:-73}_gamg_profile.txt 2>&1
grep -E "^[a-z].* s$|set-up total" gpurun_out/r03_setup_n${1:-73}_gamg_profile.txt
