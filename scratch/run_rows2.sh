#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/rows2
timeout 600 python scratch/check_rows2.py 2>&1 | tee gpurun_out/rows2/check.txt
rm -rf /tmp/r2ks; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r2ks -- python scratch/check_rows2.py > /dev/null 2>&1
python scratch/kstats.py $(find /tmp/r2ks -name "*kernel_stats.csv" | head -1) 40 | grep -i "dense\|Name" | tee gpurun_out/rows2/kstats.txt
