#!/bin/bash
# three-component gather: one 16-byte + one 8-byte load per node vs three 8-byte loads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPCD_TRIPLE_3X8=1 -o /tmp/libpcd_3x8.so fenapack_amd/csrc/pcd_engine.hip
out=gpurun_out/r03_t_triple_gather_ab.txt; : > $out
for rep in 1 2; do
  python tools/time_a00_kernel.py 3 cube >> $out 2>&1
  FENAPACK_AMD_HIP_LIB=/tmp/libpcd_3x8.so python tools/time_a00_kernel.py 3 cube >> $out 2>&1
done
grep "us per launch" $out
timeout 300 python -m pytest tests/test_hip_parity.py tests/test_kernels_random_gpu.py -x -q -m gpu 2>&1 | tail -2
