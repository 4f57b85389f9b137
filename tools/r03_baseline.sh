#!/bin/bash
# Round-3 baseline on the GPU box: cProfile of the level-6 bench set-up and the
# GPU suite with per-test durations.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
python -m cProfile -o gpurun_out/r03_a_bench_l6.prof bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03_a_bench_l6.json 2> gpurun_out/r03_a_bench_l6.err
python - <<'PY' > gpurun_out/r03_a_bench_l6_profile.txt
import pstats
p = pstats.Stats('gpurun_out/r03_a_bench_l6.prof'); p.sort_stats('cumulative').print_stats(120)
PY
(nproc; lscpu | head -20) > gpurun_out/r03_a_host.txt
python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r03_a_gpu_suite.txt 2>&1
tail -3 gpurun_out/r03_a_gpu_suite.txt
