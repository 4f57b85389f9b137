#!/bin/bash
# PMC traffic + bench line at config-5 size (cube N = 64), level 7; gamg on N = 73 / 36
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
export FENAPACK_AMD_MAX_CELLS=4000000
bash tools/gpu_pmc.sh r03_u_cube64 --geometry cube --level 4 --n0 4
cp gpurun_out/r03_u_cube64_pmc_roofline.json profiles/r03_u_pmc_roofline_cube_n64.json
bash tools/gpu_pmc.sh r03_u_l7 --level 7
cp gpurun_out/r03_u_l7_pmc_roofline.json profiles/r03_u_pmc_roofline_level7.json
python bench.py --geometry cube --level 4 --n0 4 --steps 50 --warmup 10 --no-cpu-baseline --no-producer > gpurun_out/r03_u_bench_cube64.json 2> gpurun_out/r03_u_bench_cube64.err
python bench.py --level 7 --steps 50 --warmup 10 --no-cpu-baseline --no-producer > gpurun_out/r03_u_bench_level7.json 2> gpurun_out/r03_u_bench_level7.err
for f in cube64 level7; do python -c "
import json
d=json.loads(open('gpurun_out/r03_u_bench_$f.json').read().strip().splitlines()[-1])
r=d['roofline']; q=d['pcapply_roofline']
print('$f', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, {k:r.get(k) for k in ['us_per_launch','frac_traffic','traffic_stale','frac_vs_measured_roof','measured_roof_gbs','frac_kernel_model','frac']}, {k:q.get(k) for k in ['frac_traffic','frac']})
"; done
( time timeout 900 python bench.py --geometry cube --level 0 --n0 36 --algebraic --steps 20 --warmup 5 --no-cpu-baseline --no-producer > gpurun_out/r03_u_bench_cube36_gamg.json 2> gpurun_out/r03_u_bench_cube36_gamg.err ) 2>&1 | tail -3
tail -2 gpurun_out/r03_u_bench_cube36_gamg.err; cut -c1-400 gpurun_out/r03_u_bench_cube36_gamg.json
