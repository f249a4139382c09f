#!/bin/bash
# rocprofv3 kernel-trace statistics of the cached-layout forward at the benchmark shape under both split modes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for m in 0 1; do
  rm -rf /tmp/km_$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/km_$m -- python scratch/run_fwd_mode.py $m 12 > /tmp/km_$m.log 2>&1
  f=$(find /tmp/km_$m -name "*kernel_stats.csv" | head -1)
  echo "== mode $m"; python scratch/kstats.py "$f" 22 | tee gpurun_out/kstats_mode$m.txt
  t=$(find /tmp/km_$m -name "*kernel_trace.csv" | head -1); cp "$t" gpurun_out/ktrace_mode$m.csv
done
