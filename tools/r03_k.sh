#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time python -m pytest tests/test_full_size_gpu.py tests/test_configs_thread_ranks_gpu.py -x -q -m gpu --durations=14 ) > gpurun_out/r03_k_big.txt 2>&1
tail -24 gpurun_out/r03_k_big.txt | cut -c1-160
