#!/bin/bash
# configs 4 and 5 with the round-3 producer / thread backend: timings
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time python tools/unsteady_thread_ranks.py 4 100 1 2 4 ) > gpurun_out/r03_i_unsteady_level4_100steps_thread_ranks.jsonl 2> gpurun_out/r03_i_unsteady.err
cat gpurun_out/r03_i_unsteady_level4_100steps_thread_ranks.jsonl | cut -c1-300; tail -4 gpurun_out/r03_i_unsteady.err
( time python tools/steady_thread_ranks.py cube 3 1 8 ) > gpurun_out/r03_i_cube32_thread_ranks.jsonl 2> gpurun_out/r03_i_cube32.err
cat gpurun_out/r03_i_cube32_thread_ranks.jsonl | cut -c1-300; tail -4 gpurun_out/r03_i_cube32.err
export FENAPACK_AMD_MAX_CELLS=4000000
( time python tools/parity_large.py --geometry cube --level 3 --n0 6 ) > gpurun_out/r03_i_parity_cube48.json 2> gpurun_out/r03_i_parity_cube48.err
cat gpurun_out/r03_i_parity_cube48.json; tail -4 gpurun_out/r03_i_parity_cube48.err
( time python tools/parity_large.py --geometry cube --level 4 --n0 4 ) > gpurun_out/r03_i_parity_cube64.json 2> gpurun_out/r03_i_parity_cube64.err
cat gpurun_out/r03_i_parity_cube64.json; tail -4 gpurun_out/r03_i_parity_cube64.err
