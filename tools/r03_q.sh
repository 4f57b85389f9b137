#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_reorder_gpu.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_q_bench.json 2> gpurun_out/r03_q_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_q_bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']})
"
