#!/bin/bash
# Round 6, call S: rows of the wave-per-row kernel (k_spmv_wc: the restriction of
# the first algebraic level) given to the XCDs in contiguous stretches
# (PCD_XCD_WAVE=1, the default after this call if it wins) against the plain
# order (PCD_XCD_WAVE=0), config 5's size; parity of the kernels first.
N73="--geometry cube --level 0 --n0 73 --algebraic"
t0=$(date +%s); lap() { echo "[lap] $1 rc=$2 t=$(( $(date +%s) - t0 ))s"; }
timeout 900 python -m pytest tests/test_kernels_random_gpu.py tests/test_hip_parity.py tests/test_api_gpu.py -m gpu -x -q 2>&1 | tail -3; lap parity $?
for v in 0 1 0 1; do
  PCD_XCD_WAVE=$v timeout 500 bash tools/gpu_timeline.sh r06_s_n73_xcd_wave${v} $N73; lap wave$v $?
  echo "== PCD_XCD_WAVE=$v"; tail -1 gpurun_out/r06_s_n73_xcd_wave${v}_timeline.txt; grep -E "k_spmv_wc<|k_cheb_first_tc" gpurun_out/r06_s_n73_xcd_wave${v}_timeline.txt
done
