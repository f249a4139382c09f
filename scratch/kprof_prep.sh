#!/bin/bash
# rocprofv3 kernel-trace statistics of prepare_graph at the benchmark shape (6 calls)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/kpp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kpp -- python scratch/run_prep_only.py > /tmp/kpp.log 2>&1
f=$(find /tmp/kpp -name "*kernel_stats.csv" | head -1)
python scratch/kstats.py "$f" 30 | tee gpurun_out/kstats_prep.txt
