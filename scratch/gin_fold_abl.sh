#!/bin/bash
# what the last block's BatchNorm fold of gin32_mlp_kernel costs the FORWARD (ablation build: the last block
# leaves without folding -- a stale record, garbage downstream, the same launches) -> gpurun_out/gin_fold/abl.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/gin_fold; mkdir -p $O; : > $O/abl.txt
echo "# cached-layout forward, ms (3 x 30 forwards per process; processes alternate)" >> $O/abl.txt
for rep in 1 2 3; do
  for lt in default NOFOLDWORK; do
    if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
    timeout 200 python scratch/time_fwd_lib.py 2>/dev/null | tail -1 >> $O/abl.txt
  done
done
echo "# rocprofv3 --kernel-trace --stats of 20 forwards: average us per launch in the forward" >> $O/abl.txt
for lt in default NOFOLDWORK; do
  if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
  rm -rf /tmp/gf_$lt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gf_$lt -o t -- python scratch/run_fwd_mode.py 1 20 > /tmp/gf_$lt.log 2>&1
  f=$(find /tmp/gf_$lt -name "*kernel_stats.csv" | head -1)
  python - <<PY >> $O/abl.txt
import csv
for r in csv.DictReader(open("$f")):
    if any(k in r["Name"] for k in ("nnconv32_eg", "gin32_aggregate", "gin32_mlp", "merge_bn1")):
        print("$lt".ljust(13), r["Name"][:40].ljust(40), "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done
cat $O/abl.txt
