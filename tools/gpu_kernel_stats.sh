#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (GPU box):
#   tools/gpu_kernel_stats.sh TAG [bench args] -> gpurun_out/TAG_kernel_stats.csv + TAG_bench_prof.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/${TAG}_stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $ROOT/bench.py --no-cpu-baseline --no-producer "$@" > $OUT/${TAG}_bench_prof.json 2> $OUT/${TAG}_stats.log
cd $ROOT
f=$(find $OUT/${TAG}_stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_stats
