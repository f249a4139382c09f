#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/tail
AB_REPS=6 timeout 600 python scratch/ab_tail.py 100000 ${1:-0,2} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/tail/ab.txt
rm -rf /tmp/tks; AB_REPS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tks -- python scratch/ab_tail.py 100000 ${1:-0,2} > /dev/null 2>&1
python scratch/kstats.py $(find /tmp/tks -name "*kernel_stats.csv" | head -1) 60 | grep -i "dense\|bn_fin" | tee gpurun_out/tail/kstats.txt
