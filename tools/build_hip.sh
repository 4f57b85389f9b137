#!/bin/bash
# Build libpcd_hip.so from the engine's translation units (the same commands
# as __graft_entry__.build(), for A/B builds with extra compiler flags).
#   tools/build_hip.sh OUT.so [extra hipcc flags ...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
OBJ=$(mktemp -d /tmp/pcdobj.XXXXXX)
pids=()
for u in apply setup krylov abi producer; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden "$@" \
    -c -o $OBJ/pcd_$u.o $ROOT/fenapack_amd/csrc/pcd_$u.hip &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/pcd_*.o || rc=1
rm -rf $OBJ
exit $rc
