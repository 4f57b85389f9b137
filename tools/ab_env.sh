#!/bin/bash
# A/B of an engine environment switch on one bench.py workload:
#   tools/ab_env.sh <tag> <VAR> "<value A> <value B> ..." <bench.py args...>
# -> gpurun_out/<tag>_<VAR>_<value>.json per value and a one-line summary
# (ms per PCApply, us per launch of the dominant A00 kernel).
tag=$1; var=$2; vals=$3; shift 3
mkdir -p gpurun_out
for v in $vals; do
  out=gpurun_out/${tag}_${var}_${v}.json
  env $var=$v python3 bench.py --no-cpu-baseline --no-producer "$@" > $out 2> ${out%.json}.err
  python3 - $out $var $v <<'P'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%s=%s: %.4f ms per PCApply, %.2f us per A00 launch (%s), gmres %s, %s" % (
        sys.argv[2], sys.argv[3], d["ms_per_step"], d["roofline"]["us_per_launch"],
        d["roofline"]["kernel"], d["gmres_its_per_newton_step"], d["config"]["workload"]))
except Exception as ex:
    print("%s=%s: FAILED %s" % (sys.argv[2], sys.argv[3], ex))
P
done
