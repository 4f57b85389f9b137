#!/bin/bash
# A/B of the software-pipelined persistent stream kernels (PCD_PIPE) on the
# finest A00 of cavity level 6 / 7 and cube N = 32
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
out=gpurun_out/r03_d_pipe_ab.txt; : > $out
python -c "import __graft_entry__ as g; g.build()"
for lv in 6 7; do
  PCD_PIPE=0 python tools/time_a00_kernel.py $lv >> $out 2>&1
  for w in 0 4 5 6 8; do PCD_PIPE=1 PCD_PIPE_WGS=$w python tools/time_a00_kernel.py $lv >> $out 2>&1; done
done
PCD_PIPE=0 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
PCD_PIPE=1 python tools/time_a00_kernel.py 3 cube >> $out 2>&1
grep "us per launch" $out
python -m pytest tests/test_hip_parity.py tests/test_kernels_random_gpu.py tests/test_precomposed_gpu.py -x -q -m gpu 2>&1 | tail -5
