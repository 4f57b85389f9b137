#!/bin/bash
# set-up time of the level-6 bench workload: BLAS cap by threadpoolctl (package import) vs by environment
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_setup_env2.txt; : > $out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run() { echo "== $*" >> $out; env "$@" timeout 100 python tools/setup_breakdown.py 2>&1 | grep -E "problem |nonlinear|set-up total" >> $out; }
run A=cap_by_threadpoolctl
run A=cap_by_threadpoolctl
run A=cap_by_threadpoolctl
run OPENBLAS_NUM_THREADS=8
run OPENBLAS_NUM_THREADS=8
run OPENBLAS_NUM_THREADS=8
run OPENBLAS_NUM_THREADS=8 OMP_PROC_BIND=spread OMP_PLACES=cores
run OPENBLAS_NUM_THREADS=8 OMP_PROC_BIND=spread OMP_PLACES=cores
run FENAPACK_AMD_BLAS_THREADS=0
cat $out
