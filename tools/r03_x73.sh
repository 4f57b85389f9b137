#!/bin/bash
# XCD-aware row-block mapping of the three-component non-temporal kernels at config 5's own size
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export FENAPACK_AMD_MAX_CELLS=3000000 OPENBLAS_NUM_THREADS=8
: > gpurun_out/r03_x_xcd_nt_cube73.txt
for m in 1 3; do
PCD_XCD_REMAP_NT=$m timeout 900 python bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 3 --no-cpu-baseline --no-producer > gpurun_out/r03_x73.json 2> gpurun_out/r03_x73.err || tail -5 gpurun_out/r03_x73.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_x73.json').read().strip().splitlines()[-1])
r=d['roofline']
print('PCD_XCD_REMAP_NT=$m', {k:d.get(k) for k in ['value','ms_per_step','gmres_its_per_newton_step']}, 'dominant kernel us', round(r['us_per_launch'],1))
" | tee -a gpurun_out/r03_x_xcd_nt_cube73.txt
done
