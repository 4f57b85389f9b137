#!/bin/bash
# Same-box A/B of two builds of libpcd_hip.so on the per-launch timeline of one
# eager PCApply (tools/gpu_timeline.sh): prints the launches whose duration
# differs, and the spans.
#   tools/ab_lib_timeline.sh <tag> <base.so> <new.so> [bench args ...]
tag=$1; A=$(readlink -f $2); B=$(readlink -f $3); shift 3
FENAPACK_AMD_HIP_LIB=$A tools/gpu_timeline.sh ${tag}_base "$@"
FENAPACK_AMD_HIP_LIB=$B tools/gpu_timeline.sh ${tag}_new "$@"
FENAPACK_AMD_HIP_LIB=$A tools/gpu_timeline.sh ${tag}_base2 "$@"
FENAPACK_AMD_HIP_LIB=$B tools/gpu_timeline.sh ${tag}_new2 "$@"
paste <(awk '{print $1, $2, $4}' gpurun_out/${tag}_base_timeline.txt) \
      <(awk '{print $4}' gpurun_out/${tag}_new_timeline.txt) \
      <(awk '{print $4}' gpurun_out/${tag}_base2_timeline.txt) \
      <(awk '{print $4}' gpurun_out/${tag}_new2_timeline.txt)
