#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/kprof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof -- python scratch/run_prep_only.py > gpurun_out/kprof.log 2>&1
f=$(find gpurun_out/kprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    if 'at::' not in r['Name']:
        per = float(r['TotalDurationNs'])/6e3
        tot += per
        print(f"{r['Name'][:64]:64s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.2f} per_prep_us={per:8.1f}")
print("sum per prep us", tot)
PY
