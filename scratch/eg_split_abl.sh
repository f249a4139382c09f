#!/bin/bash
# what the row split and the message split of nnconv32_eg_kernel cost INSIDE the forward (ablation builds:
# garbage values, the same instruction stream minus the splits) -> gpurun_out/eg_split/abl.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/eg_split; mkdir -p $O; : > $O/abl.txt
echo "# cached-layout forward, ms (3 x 30 forwards per process; processes alternate)" >> $O/abl.txt
for rep in 1 2 3; do
  for lt in default EGNOSPLIT EGNOMSGSPLIT EGNOBOTH; do
    if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
    timeout 200 python scratch/time_fwd_lib.py 2>/dev/null | tail -1 >> $O/abl.txt
  done
done
echo "# rocprofv3 --kernel-trace --stats of 20 forwards: the NNConv and the kernels beside it, average us per launch in the forward" >> $O/abl.txt
for lt in default EGNOSPLIT EGNOMSGSPLIT EGNOBOTH; do
  if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
  rm -rf /tmp/egs_$lt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/egs_$lt -o t -- python scratch/run_fwd_mode.py 1 20 > /tmp/egs_$lt.log 2>&1
  f=$(find /tmp/egs_$lt -name "*kernel_stats.csv" | head -1)
  python - <<PY >> $O/abl.txt
import csv
for r in csv.DictReader(open("$f")):
    if any(k in r["Name"] for k in ("nnconv32_eg", "gin32_aggregate", "gin32_mlp", "merge_bn1")):
        print("$lt".ljust(13), r["Name"][:40].ljust(40), "calls", r["Calls"], "avg_us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
done
cat $O/abl.txt
