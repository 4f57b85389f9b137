#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python tools/host_micro.py > gpurun_out/r03_host_micro.txt 2>&1
cat gpurun_out/r03_host_micro.txt
