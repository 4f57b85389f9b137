#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/short_row_prolongation.hip -o /tmp/short_row_prolongation && timeout 300 /tmp/short_row_prolongation > gpurun_out/r06_t_short_row_prolongation.txt 2>&1
cat gpurun_out/r06_t_short_row_prolongation.txt
