#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time python -m pytest tests/test_full_size_gpu.py tests/test_configs_thread_ranks_gpu.py tests/test_multi_gpu_threads.py -x -q -m gpu --durations=12 ) > gpurun_out/r03_o_big.txt 2>&1
tail -22 gpurun_out/r03_o_big.txt | cut -c1-160
for lv in 7; do python bench.py --level $lv --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_o_bench_l$lv.json 2> gpurun_out/r03_o_bench_l$lv.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_o_bench_l$lv.json').read().strip().splitlines()[-1])
print('level $lv', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, d['roofline']['us_per_launch'], d['roofline']['kernel'])
"; done
python bench.py --geometry cube --level 3 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_o_bench_cube32.json 2> gpurun_out/r03_o_bench_cube32.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_o_bench_cube32.json').read().strip().splitlines()[-1])
print('cube32', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, d['roofline']['us_per_launch'], d['roofline']['kernel'])
"
