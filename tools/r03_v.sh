#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
for v in 1 0; do PCD_NO_SMALL_TILE=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-producer > gpurun_out/r03_v_bench_nosmall$v.json 2> gpurun_out/r03_v_bench_nosmall$v.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_v_bench_nosmall$v.json').read().strip().splitlines()[-1])
print('NO_SMALL_TILE=$v', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, d['roofline']['measured_probes_gbs'])
"; done
( time timeout 600 python -m pytest tests/test_two_gpus.py tests/test_kernels_random_gpu.py tests/test_hip_parity.py -x -q -m gpu ) 2>&1 | tail -6
