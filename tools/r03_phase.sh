#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
PCD_SETUP_TIMING=1 OPENBLAS_NUM_THREADS=8 python tools/setup_breakdown.py 2>&1 | grep -E "pcd set-up|set-up total|problem|nonlinear" | head -40
