#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OPENBLAS_NUM_THREADS=8 FENAPACK_AMD_MAX_CELLS=3000000 timeout 1500 python tools/parity_large.py --geometry cube --level 0 --n0 ${1:-73} --algebraic > gpurun_out/r03_parity_cube${1:-73}_gamg.json 2> gpurun_out/r03_parity_cube${1:-73}_gamg.err || tail -5 gpurun_out/r03_parity_cube${1:-73}_gamg.err
cat gpurun_out/r03_parity_cube${1:-73}_gamg.json
