#!/bin/bash
# NOTE: the PCD_PIPE kernels this script A/B-ed were in the tree at commit 82fcaca only (negative result: profiles/r03_g_pipelined_kernels_negative_result.txt)
# pipelined persistent kernel x workgroups per CU x non-temporal matrix stream
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -DPCD_NT_LOADS=1 -o /tmp/libpcd_nt.so fenapack_amd/csrc/pcd_engine.hip
out=gpurun_out/r03_g_pipe_nt.txt; : > $out
for lv in 6 7; do
  for lib in "" /tmp/libpcd_nt.so; do
    FENAPACK_AMD_HIP_LIB=$lib PCD_PIPE=0 python tools/time_a00_kernel.py $lv >> $out 2>&1
    for w in 1 2 3 4 5; do FENAPACK_AMD_HIP_LIB=$lib PCD_PIPE=1 PCD_PIPE_WGS=$w python tools/time_a00_kernel.py $lv >> $out 2>&1; done
  done
done
grep "us per launch" $out
