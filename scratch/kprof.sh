#!/bin/bash
# usage: kprof.sh <which: nnconv|gin> ; per-kernel average durations of the single-op runner
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/kprof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof -- python scratch/run_nnconv_only.py $1 > gpurun_out/kprof.log 2>&1
f=$(find gpurun_out/kprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f}")
PY
