#!/bin/bash
# HBM-side traffic of the single-op runner (separate --pmc passes, kernel-trace only, each under its own timeout)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
which=$1; kern=$2
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_hbm_$c
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_hbm_$c -- python scratch/run_nnconv_only.py $which > gpurun_out/pmc_hbm_$c.log 2>&1
  echo "rc=$?"
  f=$(find gpurun_out/pmc_hbm_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py $kern $f
done
