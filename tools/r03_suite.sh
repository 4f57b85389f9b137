#!/bin/bash
# GPU suite with per-test durations (tag = $1), then a bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag=${1:-x}
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r03_${tag}_gpu_suite.txt 2>&1
tail -32 gpurun_out/r03_${tag}_gpu_suite.txt | cut -c1-150
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r03_${tag}_bench_l6.json 2> gpurun_out/r03_${tag}_bench_l6.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_${tag}_bench_l6.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']})
"
