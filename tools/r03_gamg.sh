#!/bin/bash
# -pc_type gamg on 3-D matrices without a nested hierarchy (tag, n0 ...)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
free -g | head -2
for n0 in "$@"; do
  FENAPACK_AMD_MAX_CELLS=3000000 timeout 1500 python bench.py --geometry cube --level 0 --n0 $n0 --algebraic --steps 20 --warmup 3 --no-cpu-baseline --no-producer > gpurun_out/r03_gamg_cube_n${n0}.json 2> gpurun_out/r03_gamg_cube_n${n0}.err || tail -5 gpurun_out/r03_gamg_cube_n${n0}.err
  python -c "
import json
d=json.loads(open('gpurun_out/r03_gamg_cube_n${n0}.json').read().strip().splitlines()[-1])
print('N=${n0}', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, d['config'].get('workload'))
"
done
