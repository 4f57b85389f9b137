#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p /tmp/pcdlibs
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPCD_FAKE_ROWLEN=6 -o /tmp/pcdlibs/fake.so fenapack_amd/csrc/pcd_engine.hip
for i in 1 2; do
  python3 tools/time_small_solve.py 6
  FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/fake.so python3 tools/time_small_solve.py 6
done
python3 tools/time_small_solve.py 4
FENAPACK_AMD_HIP_LIB=/tmp/pcdlibs/fake.so python3 tools/time_small_solve.py 4
