#!/bin/bash
# scratch/abl_ps.sh <tag> <libtags...>: rocprofv3 average of the ps kernel (20 launches at 100k) for the default library and ablation builds
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/$tag
for lt in default "$@"; do
  if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
  rm -rf /tmp/abl_$lt
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$lt -o t -- python scratch/run_ps_only.py ps 20 > /tmp/abl_$lt.log 2>&1
  f=$(find /tmp/abl_$lt -name "*kernel_stats.csv" | head -1)
  python - <<PY | tee -a gpurun_out/$tag/abl.txt
import csv
rows=[r for r in csv.DictReader(open("$f")) if "nnconv32_ps" in r["Name"]]
for r in rows: print("$lt".ljust(12), r["Name"][:40], "calls", r["Calls"], "avg_us %.1f min %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done
