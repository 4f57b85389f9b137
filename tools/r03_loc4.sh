#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
for lh in 0 1; do
PCD_FORCE_COMM=1 FENAPACK_AMD_LOCAL_HANDOVER=$lh python bench.py --level 5 --steps 50 --warmup 5 --no-cpu-baseline --no-producer > gpurun_out/r03_loc4.json 2> gpurun_out/r03_loc4.err || tail -20 gpurun_out/r03_loc4.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_loc4.json').read().strip().splitlines()[-1])
print('local handover $lh', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']})
"
done
timeout 300 python -m pytest tests/test_two_gpus.py -m gpu -x -q 2>&1 | tail -3
