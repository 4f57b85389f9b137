#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_multi_gpu_threads.py tests/test_configs_thread_ranks_gpu.py -m gpu -x -q -k "rank_local or host_driven" 2>&1 | tail -30
