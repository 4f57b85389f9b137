#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier.hip -o /tmp/grid_barrier && timeout 200 /tmp/grid_barrier > gpurun_out/r03_y_grid_barrier.txt 2>&1
cat gpurun_out/r03_y_grid_barrier.txt
