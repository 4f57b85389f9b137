#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for lv in 6 7; do
for opt in "" "--rediscretise-u"; do
  python bench.py --level $lv --steps 200 --warmup 20 --no-cpu-baseline --no-producer $opt > gpurun_out/r03_redisc.json 2> gpurun_out/r03_redisc.err || tail -5 gpurun_out/r03_redisc.err
  python -c "
import json
d=json.loads(open('gpurun_out/r03_redisc.json').read().strip().splitlines()[-1])
print('level $lv [$opt]', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']})
"
done; done
