#!/bin/bash
# 3-D kernel: smaller LDS tile (more resident workgroups) with 32-row blocks in one pass
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
out=gpurun_out/r03_t_tile3_rb32.txt; : > $out
python tools/time_a00_kernel.py 3 cube >> $out 2>&1
PCD_MAX_RB=32 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
for t in 1024 1280; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPCD_TILE3=$t -o /tmp/libpcd_t$t.so fenapack_amd/csrc/pcd_engine.hip
  FENAPACK_AMD_HIP_LIB=/tmp/libpcd_t$t.so PCD_MAX_RB=32 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
  FENAPACK_AMD_HIP_LIB=/tmp/libpcd_t$t.so PCD_MAX_RB=32 PCD_MAX_CHUNKS=1 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
done
grep "us per launch" $out
