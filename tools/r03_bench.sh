#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
python bench.py > gpurun_out/r03_zzz_bench_level6.json 2> gpurun_out/r03_zzz_bench_level6.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_zzz_bench_level6.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','setup_seconds','gmres_its_per_newton_step']}); r=d['roofline']; print({k:r[k] for k in r if k!='measured_probes_gbs'}); print(r['measured_probes_gbs']); q=d['pcapply_roofline']; print({k:q.get(k) for k in ['frac_traffic','traffic','frac','frac_vs_measured_roof']}); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
python -c "import __graft_entry__ as g; g.smoke()"
