#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c3
bash scratch/kstats3.sh c3_stats 2>&1 | tee gpurun_out/c3/kernel_stats.txt
timeout 200 python scratch/time_config3.py 2>&1 | grep -i "bfloat16\|ms" | tail -3 | tee -a gpurun_out/c3/kernel_stats.txt
bash scratch/pmc_config3.sh 2>&1 | tee gpurun_out/c3/pmc.txt
