ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r06_u_trace
rocprofv3 --kernel-trace --output-format csv -d $OUT/r06_u_trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-producer --steps 5 --warmup 2 > $OUT/r06_u_trace.log 2>&1
cd $ROOT
python tools/trace_one_apply.py $OUT/r06_u_trace > $OUT/r06_u_graph_mode_timeline.txt 2>&1
cat $OUT/r06_u_graph_mode_timeline.txt
rm -rf $OUT/r06_u_trace
