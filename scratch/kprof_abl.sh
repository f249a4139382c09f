#!/bin/bash
# usage: kprof_abl.sh TAG...   -- rocprofv3 kernel-trace average of the NNConv column kernel for every ablation library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for t in "$@"; do
  rm -rf /tmp/kp_$t
  TGNN_LIB_PATH=$PWD/scratch/libs/libtgnn_$t.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp_$t -- python scratch/time_nnconv.py > /tmp/kp_$t.log 2>&1
  f=$(find /tmp/kp_$t -name "*kernel_stats.csv" | head -1)
  python - "$f" "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if 'cols_kernel' in r['Name'] or 'weight_image' in r['Name']:
        print(f"{sys.argv[2]:10s} {r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f}")
PY
done
