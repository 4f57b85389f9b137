#!/bin/bash
# Per-launch timeline of one PCApply of the default bench workload (eager
# launches under rocprofv3 --kernel-trace) + kernel stats of the graph run.
# usage (on the GPU box, from the repo root): tools/gpu_timeline.sh TAG [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-producer --no-graph --steps 5 --warmup 2 "$@" > $OUT/${TAG}_trace.log 2>&1
cd $ROOT
python tools/trace_one_apply.py $OUT/${TAG}_trace > $OUT/${TAG}_timeline.txt 2>&1
rm -rf $OUT/${TAG}_trace
