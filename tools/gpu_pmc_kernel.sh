#!/bin/bash
# What bounds the roofline kernel?  SQ / TCC / TCP counter passes over
# tools/time_a00_kernel.py (cavity level 6), one --pmc set per pass.
#   tools/gpu_pmc_kernel.sh TAG [time_a00_kernel args, e.g. "3 cube"]  ->  gpurun_out/TAG_pmc_kernel.txt
TAG=$1; shift
WL=${@:-6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TCC|SQ|TA|TD)_[A-Za-z0-9_]+" | sort -u > $OUT/${TAG}_counters_available.txt
: > $OUT/${TAG}_pmc_kernel.txt
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TD_BUSY_avr" "TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rm -rf $OUT/${TAG}_pk_$i
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $OUT/${TAG}_pk_$i -- python3 $ROOT/tools/time_a00_kernel.py $WL > $OUT/${TAG}_pk_$i.log 2>&1
  python3 - $OUT/${TAG}_pk_$i "$SET" >> $OUT/${TAG}_pmc_kernel.txt <<'PY'
import csv, glob, os, sys
root, sets = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_cheb_step_sc" not in n and "k_cheb_step_tc" not in n and "k_cheb_step_lm" not in n:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if not acc:
    print("set [%s]: no data (counter names not accepted?)" % sets)
for k, v in sorted(acc.items()):
    print("%-40s mean %.6g over %d launches of k_cheb_step_[st]c|lm" % (k, sum(v) / len(v), len(v)))
PY
  rm -rf $OUT/${TAG}_pk_$i
done
