#!/bin/bash
# round-3 profile set of the headline workload: PMC traffic (re-keyed to the
# current kernel sources), rocprofv3 kernel stats of the bench command, timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
bash tools/gpu_pmc.sh r03_s
bash tools/gpu_kernel_stats.sh r03_s --steps 200 --warmup 20
bash tools/gpu_timeline.sh r03_s
ls -la gpurun_out/r03_s_*; cat gpurun_out/r03_s_pmc_roofline.json | head -40; tail -5 gpurun_out/r03_s_pmc_roofline.err
head -12 gpurun_out/r03_s_kernel_stats.csv | cut -c1-200
tail -8 gpurun_out/r03_s_timeline.txt
