#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
( time python -m pytest tests/test_configs_thread_ranks_gpu.py tests/test_multi_gpu_threads.py -x -q -m gpu --durations=12 ) > gpurun_out/r03_j_threads.txt 2>&1
tail -22 gpurun_out/r03_j_threads.txt | cut -c1-160
python -m cProfile -o gpurun_out/r03_j_l7.prof -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "level7" > gpurun_out/r03_j_l7.txt 2>&1
tail -3 gpurun_out/r03_j_l7.txt
python - <<'PY' > gpurun_out/r03_j_l7_profile.txt
import pstats
p = pstats.Stats('gpurun_out/r03_j_l7.prof'); p.sort_stats('tottime').print_stats(45)
PY
head -75 gpurun_out/r03_j_l7_profile.txt | cut -c1-150
