#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tag=${1:-s}
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
nproc > gpurun_out/r03_${tag}_setup.txt; lscpu | grep -E "Model name|^CPU\(s\)|Socket|NUMA node\(s\)" >> gpurun_out/r03_${tag}_setup.txt
python tools/setup_breakdown.py >> gpurun_out/r03_${tag}_setup.txt 2>&1
python tools/setup_breakdown.py --bind >> gpurun_out/r03_${tag}_setup.txt 2>&1
python tools/setup_breakdown.py --profile > gpurun_out/r03_${tag}_setup_profile.txt 2>&1
grep -E "^[a-z].* s$|set-up total|^  [a-z_]+ +[0-9]+ +[0-9.]+$|Model|CPU|NUMA|Socket" gpurun_out/r03_${tag}_setup.txt | head -90
