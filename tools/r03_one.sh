#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "anchor" 2>&1 | tail -12
