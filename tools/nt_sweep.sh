#!/bin/bash
# A/B: non-temporal loads of the matrix stream, unroll depth (compile-time)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/pcdlibs
python3 tools/time_a00_kernel.py 6; python3 tools/time_a00_kernel.py 6
python3 tools/time_a00_kernel.py 3 cube
for V in "-DPCD_NT_LOADS=1:nt" "-DPCD_UNROLL=4:u4" "-DPCD_UNROLL=6:u6" "-DPCD_NT_LOADS=1 -DPCD_UNROLL=6:nt_u6"; do
  F=${V%%:*}; N=${V##*:}
  tools/build_hip.sh /tmp/pcdlibs/$N.so $F
  FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/$N.so python3 tools/time_a00_kernel.py 6
  FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/$N.so python3 tools/time_a00_kernel.py 6
  FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/$N.so python3 tools/time_a00_kernel.py 3 cube
done
