#!/bin/bash
# Round 6, call R: LDS tile of the 3-component stream kernels (prolongation-add
# k_spmv_sc<.,1,3>, discrete gradient k_spmv_rk<.,2,3>) at config 5's size -
# compile-time variants built in the container (tools/build_hip.sh
# -DPCD_TILE3=768/1024/2048, -DPCD_UNROLL=4), one eager-PCApply timeline each.
N73="--geometry cube --level 0 --n0 73 --algebraic"
t0=$(date +%s); lap() { echo "[lap] $1 rc=$2 t=$(( $(date +%s) - t0 ))s"; }
timeout 500 bash tools/gpu_timeline.sh r06_r_n73_base $N73; lap base $?
for v in t3_1024 t3_768 t3_2048 u4; do
  FENAPACK_AMD_HIP_LIB=$(pwd)/fenapack_amd/lib/ab/$v.so timeout 500 bash tools/gpu_timeline.sh r06_r_n73_$v $N73; lap $v $?
done
PCD_MAX_RB=64 timeout 500 bash tools/gpu_timeline.sh r06_r_n73_rb64 $N73; lap rb64 $?
timeout 500 bash tools/gpu_timeline.sh r06_r_n73_base2 $N73; lap base2 $?
for f in base t3_1024 t3_768 t3_2048 u4 rb64 base2; do echo "== $f"; tail -1 gpurun_out/r06_r_n73_${f}_timeline.txt; grep -E "k_spmv_sc<|k_spmv_rk<" gpurun_out/r06_r_n73_${f}_timeline.txt; done
