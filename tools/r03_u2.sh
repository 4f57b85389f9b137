#!/bin/bash
# PMC traffic + bench lines at level 7 and cube N = 64 on the final kernels
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
export FENAPACK_AMD_MAX_CELLS=4000000
bash tools/gpu_pmc.sh r03_z_l7 --level 7
cp gpurun_out/r03_z_l7_pmc_roofline.json profiles/r03_z_pmc_roofline_level7.json
bash tools/gpu_pmc.sh r03_z_cube64 --geometry cube --level 4 --n0 4
cp gpurun_out/r03_z_cube64_pmc_roofline.json profiles/r03_z_pmc_roofline_cube_n64.json
python bench.py --level 7 --steps 50 --warmup 10 --no-cpu-baseline --no-producer > gpurun_out/r03_z_bench_level7.json 2> gpurun_out/r03_z_bench_level7.err
python bench.py --geometry cube --level 4 --n0 4 --steps 50 --warmup 10 --no-cpu-baseline --no-producer > gpurun_out/r03_z_bench_cube64.json 2> gpurun_out/r03_z_bench_cube64.err
python bench.py --geometry cube --level 3 --n0 4 --steps 100 --warmup 10 --no-cpu-baseline --no-producer > gpurun_out/r03_z_bench_cube32.json 2> gpurun_out/r03_z_bench_cube32.err
for f in level7 cube64 cube32; do python -c "
import json
d=json.loads(open('gpurun_out/r03_z_bench_$f.json').read().strip().splitlines()[-1])
r=d['roofline']; q=d['pcapply_roofline']
print('$f', {k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}, {k:r.get(k) for k in ['us_per_launch','traffic','kernel_model_bytes_per_launch','frac_traffic','traffic_stale','frac_vs_measured_roof','measured_roof_gbs','measured_roof_probe']}, {k:q.get(k) for k in ['frac_traffic','frac']})
"; done
