#!/bin/bash
# PMC traffic + bench line of config 5's own mesh (cube N = 73, gamg)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export FENAPACK_AMD_MAX_CELLS=3000000 OPENBLAS_NUM_THREADS=8
timeout 1500 bash tools/gpu_pmc.sh r03_n73 --geometry cube --level 0 --n0 73 --algebraic
cat gpurun_out/r03_n73_pmc_roofline.json | head -c 600; echo
cp gpurun_out/r03_n73_pmc_roofline.json profiles/r03_zzz_pmc_roofline_cube_n73.json
timeout 900 python bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 3 --no-cpu-baseline --no-producer > gpurun_out/r03_gamg_cube_n73.json 2> gpurun_out/r03_gamg_cube_n73.err || tail -5 gpurun_out/r03_gamg_cube_n73.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_gamg_cube_n73.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}); r=d['roofline']; print({k:r.get(k) for k in ['kernel','us_per_launch','traffic','frac_traffic','traffic_stale','frac_vs_measured_roof','frac_kernel_model']}); q=d['pcapply_roofline']; print({k:q.get(k) for k in ['frac_traffic','traffic','frac']})
"
