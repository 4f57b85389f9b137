#!/bin/bash
# Round 6, call V: the whole one-GPU PCApply (split gather + apply + scatter) as
# ONE graph launch (PCD_FS_ONE_GRAPH=1) against gather / graph / scatter as three
# submissions (0), alternating on one box: headline, level 7, cube N = 48.
for args in "" "--level 7" "--geometry cube --level 0 --n0 48 --algebraic"; do
  for v in 0 1 0 1; do
    PCD_FS_ONE_GRAPH=$v python3 bench.py --no-cpu-baseline --no-producer --steps 200 --warmup 20 $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one_graph=$v', '$args', round(d['value'],1), 'PCApply/s', round(d['ms_per_step'],4), 'ms')"
  done
done
