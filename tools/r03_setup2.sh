#!/bin/bash
# set-up time of the level-6 bench workload under thread-runtime settings
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_setup_env.txt; : > $out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
run() { echo "== $*" >> $out; env "$@" timeout 120 python tools/setup_breakdown.py 2>&1 | grep -E "problem |nonlinear|set-up total" >> $out; }
run A=default
run A=default
run OPENBLAS_NUM_THREADS=8
run OPENBLAS_NUM_THREADS=1
run OPENBLAS_NUM_THREADS=8 OMP_NUM_THREADS=16
run OPENBLAS_NUM_THREADS=8 OMP_NUM_THREADS=8
run OPENBLAS_NUM_THREADS=8 OMP_WAIT_POLICY=passive
run OPENBLAS_NUM_THREADS=8 OMP_NUM_THREADS=32 OMP_PROC_BIND=close OMP_PLACES=cores
run OPENBLAS_NUM_THREADS=8 OMP_NUM_THREADS=1
cat $out
