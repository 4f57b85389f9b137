#!/bin/bash
# the set-up of `bench.py --gpus 8` at the headline size on 8 THREAD ranks of one GPU
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
OPENBLAS_NUM_THREADS=8 FENAPACK_AMD_LOCAL_HANDOVER=1 timeout 900 python tools/steady_thread_ranks.py --host cavity 6 1 2 8 > gpurun_out/r03_scale_shape_level6.jsonl 2> gpurun_out/r03_scale_shape_level6.err
cat gpurun_out/r03_scale_shape_level6.jsonl | cut -c1-700; tail -3 gpurun_out/r03_scale_shape_level6.err
