#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time timeout 400 python -m pytest tests/test_reorder_gpu.py -x -q -m gpu -s ) > gpurun_out/r03_p_reorder.txt 2>&1
tail -40 gpurun_out/r03_p_reorder.txt | cut -c1-200
