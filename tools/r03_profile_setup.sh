#!/bin/bash
# cProfile of the bench set-up at level 6 (tag = $1)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
tag=${1:-x}
python -c "import __graft_entry__ as g; g.build()"
python -m cProfile -o gpurun_out/r03_${tag}_bench_l6.prof bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_${tag}_bench_l6.json 2> gpurun_out/r03_${tag}_bench_l6.err
python - <<PY > gpurun_out/r03_${tag}_bench_l6_profile.txt
import pstats
p = pstats.Stats('gpurun_out/r03_${tag}_bench_l6.prof'); p.sort_stats('cumulative').print_stats(140)
p.sort_stats('tottime').print_stats(40)
PY
tail -c 600 gpurun_out/r03_${tag}_bench_l6.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_${tag}_bench_l6.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']})
"
