#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 800 python -m pytest tests/test_two_gpus.py tests/test_multi_gpu_threads.py tests/test_configs_thread_ranks_gpu.py tests/test_reorder_gpu.py -m gpu -x -q 2>&1 | tail -8
