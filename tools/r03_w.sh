#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
timeout 600 python -m pytest tests/test_two_gpus.py -x -q -m gpu 2>&1 | tail -40 | cut -c1-250
