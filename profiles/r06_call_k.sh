#!/bin/bash
# round 6, GPU call K: what a cluster numbering of the finest level is worth on
# config 5's own mesh through the algebraic hierarchy (the engine's existing
# renumbering switch; host-driven: the device producer refuses a renumbered engine)
out=gpurun_out
for mode in none cluster; do
  PCD_REORDER=$mode timeout 900 python3 bench.py --geometry cube --level 0 --n0 73 --algebraic --steps 20 --warmup 5 --no-cpu-baseline --no-producer > $out/r06_k_bench_cube_n73_reorder_$mode.json 2> $out/r06_k_bench_cube_n73_reorder_$mode.err
  echo "n73 $mode rc $?"
done
for mode in none cluster; do
  PCD_REORDER=$mode timeout 900 python3 bench.py --geometry cube --level 3 --n0 6 --algebraic --steps 20 --warmup 5 --no-cpu-baseline --no-producer > $out/r06_k_bench_cube_n48_gamg_reorder_$mode.json 2> $out/r06_k_bench_cube_n48_gamg_reorder_$mode.err
  echo "n48 $mode rc $?"
done
