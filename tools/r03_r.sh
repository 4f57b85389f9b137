#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time timeout 800 python -m pytest tests/test_reorder_gpu.py tests/test_configs_thread_ranks_gpu.py -x -q -m gpu -s --durations=6 ) > gpurun_out/r03_r.txt 2>&1
grep -E "level [0-9]:|GMRES:|passed|failed|s call|real" gpurun_out/r03_r.txt | cut -c1-170
