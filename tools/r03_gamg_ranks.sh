#!/bin/bash
# the algebraic hierarchy of a 3-D problem without nested meshes on 8 thread ranks, rank-local hand-over
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OPENBLAS_NUM_THREADS=8 FENAPACK_AMD_LOCAL_HANDOVER=1 PCD_REPLICATE_BELOW=${2:-20000} FENAPACK_AMD_MAX_CELLS=3000000 timeout 1700 python tools/steady_thread_ranks.py --host --algebraic --n0=${1:-18} cube 0 ${3:-1 2 8} > gpurun_out/r03_gamg_ranks_n${1:-18}.jsonl 2> gpurun_out/r03_gamg_ranks_n${1:-18}.err
cat gpurun_out/r03_gamg_ranks_n${1:-18}.jsonl | cut -c1-600; tail -3 gpurun_out/r03_gamg_ranks_n${1:-18}.err
