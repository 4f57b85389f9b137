#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/bw_sweep.hip -o /tmp/bw_sweep && timeout 300 /tmp/bw_sweep > gpurun_out/r03_f_bw_sweep.txt 2>&1
sort -k7 -n -r gpurun_out/r03_f_bw_sweep.txt | head -40
