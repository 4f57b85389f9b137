#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/small_level_latency.hip -o /tmp/small_level_latency && timeout 120 /tmp/small_level_latency > gpurun_out/r03_y_small_level_latency.txt 2>&1
cat gpurun_out/r03_y_small_level_latency.txt
