#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
export FENAPACK_AMD_TEST_WORKERS=0
python -m pytest tests -m gpu -x -q --durations=0 --deselect tests/test_configs_thread_ranks_gpu.py --deselect tests/test_full_size_gpu.py::test_cube_n64_config5_size > gpurun_out/r03_n_gpu_suite_all_durations.txt 2>&1
tail -3 gpurun_out/r03_n_gpu_suite_all_durations.txt
python -m cProfile -o gpurun_out/r03_n_n64.prof -m pytest tests/test_full_size_gpu.py -x -q -m gpu -k "n64" > gpurun_out/r03_n_n64.txt 2>&1
tail -3 gpurun_out/r03_n_n64.txt
python - <<'PY' > gpurun_out/r03_n_n64_profile.txt
import pstats
p = pstats.Stats('gpurun_out/r03_n_n64.prof'); p.sort_stats('tottime').print_stats(40)
PY
head -60 gpurun_out/r03_n_n64_profile.txt | cut -c1-150
