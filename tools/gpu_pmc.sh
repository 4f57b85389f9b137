#!/bin/bash
# The two PMC passes behind roofline.traffic (GPU box, repo root):
#   tools/gpu_pmc.sh TAG [pmc_pcapply args]   ->  gpurun_out/TAG_pmc_roofline.json
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -- python3 $ROOT/tools/pmc_pcapply.py "$@" > $OUT/${TAG}_pmc_$C.log 2>&1
done
cd $ROOT
NU=$(grep -o "n_u [0-9]*" $OUT/${TAG}_pmc_FETCH_SIZE.log | tail -1 | cut -d" " -f2)
python3 tools/pmc_roofline.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $NU > $OUT/${TAG}_pmc_roofline.json 2> $OUT/${TAG}_pmc_roofline.err
# (kept when the summary failed: what was collected says why)
if [ -s $OUT/${TAG}_pmc_roofline.json ]; then rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE; else
  for C in FETCH_SIZE WRITE_SIZE; do find $OUT/${TAG}_pmc_$C -name "*kernel_trace.csv" -delete; find $OUT/${TAG}_pmc_$C -name "*counter_collection.csv" -exec sh -c 'cut -d, -f1-12 "$1" | head -400 > "$1.head"; rm "$1"' _ {} \; ; done
fi
