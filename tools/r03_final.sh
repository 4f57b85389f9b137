#!/bin/bash
# round-3 evidence set on the final sources: GPU suite with durations, PMC
# traffic (level 6 / level 7 / cube N = 64), kernel stats + timeline, bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-r03_z}
python -c "import __graft_entry__ as g; g.build()"
( time python -m pytest tests -m gpu -x -q --durations=30 ) > gpurun_out/${T}_gpu_suite.txt 2>&1
tail -40 gpurun_out/${T}_gpu_suite.txt | cut -c1-150
bash tools/gpu_pmc.sh ${T}_l6
bash tools/gpu_kernel_stats.sh ${T}_l6 --steps 200 --warmup 20
bash tools/gpu_timeline.sh ${T}_l6
python bench.py > gpurun_out/${T}_bench_level6.json 2> gpurun_out/${T}_bench_level6.err
python -c "
import json
d=json.loads(open('gpurun_out/${T}_bench_level6.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}); print(d['roofline']); print(d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
"
