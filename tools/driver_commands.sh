#!/bin/bash
# The driver's three round-end commands, verbatim, in its order - plus the
# host's memory facts and the suite's resident-set table (tests/conftest.py).
#   tools/driver_commands.sh [tag]      -> gpurun_out/<tag>_{host,pytest,smoke,bench}.txt
# Every python process below runs under the resident-set watchdog
# (fenapack_amd/_guard.py); nothing here builds more than the suite does.
tag=${1:-driver}
out=gpurun_out
mkdir -p $out
{
  echo "nproc $(nproc)"; grep -E "MemTotal|MemAvailable" /proc/meminfo
  echo "cgroup memory.max $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"
  echo "cgroup memory.current $(cat /sys/fs/cgroup/memory.current 2>/dev/null)"
  lscpu | grep -E "Model name|Socket|Core|Thread|NUMA node\(s\)"
} > $out/${tag}_host.txt 2>&1
export FENAPACK_AMD_RSS_TABLE=$out/${tag}_suite_rss.txt
t0=$(date +%s)
timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=25 > $out/${tag}_pytest.txt 2>&1
echo "pytest rc $? in $(( $(date +%s) - t0 )) s" | tee -a $out/${tag}_pytest.txt
unset FENAPACK_AMD_RSS_TABLE
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $out/${tag}_smoke.txt 2>&1
echo "smoke rc $?" | tee -a $out/${tag}_smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench rc $?" | tee -a $out/${tag}_bench.err
tail -5 $out/${tag}_pytest.txt
tail -c 600 $out/${tag}_bench.json
