#!/bin/bash
# NOTE: the PCD_PIPE kernels this script A/B-ed were in the tree at commit 82fcaca only (negative result: profiles/r03_g_pipelined_kernels_negative_result.txt)
# quick GPU check: a slice of the suite with timings + pipelined-kernel A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time python -m pytest tests/test_hip_parity.py tests/test_kernels_random_gpu.py tests/test_precomposed_gpu.py tests/test_multi_gpu_threads.py -x -q -m gpu --durations=8 ) > gpurun_out/r03_e_slice.txt 2>&1
tail -16 gpurun_out/r03_e_slice.txt | cut -c1-160
out=gpurun_out/r03_d_pipe_ab.txt; : > $out
for lv in 6 7; do
  PCD_PIPE=0 python tools/time_a00_kernel.py $lv >> $out 2>&1
  for w in 0 4 6 8; do PCD_PIPE=1 PCD_PIPE_WGS=$w python tools/time_a00_kernel.py $lv >> $out 2>&1; done
done
PCD_PIPE=0 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
PCD_PIPE=1 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
grep "us per launch" $out
for p in 0 1; do PCD_PIPE=$p python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-producer > gpurun_out/r03_e_bench_pipe$p.json 2> gpurun_out/r03_e_bench_pipe$p.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_e_bench_pipe$p.json').read().strip().splitlines()[-1])
print('PIPE=$p', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, d['roofline']['us_per_launch'], d['roofline']['measured_probes_gbs'])
"; done
PCD_PIPE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-producer --coarse-u 4000 > gpurun_out/r03_e_bench_coarse4000.json 2> gpurun_out/r03_e_bench_coarse4000.err; python -c "
import json
d=json.loads(open('gpurun_out/r03_e_bench_coarse4000.json').read().strip().splitlines()[-1])
print('coarse 4000', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']})
"
