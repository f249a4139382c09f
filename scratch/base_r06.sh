#!/bin/bash
# round 6 baseline: bench line + timelines of the forward and the preparation at 100 000 nodes -> gpurun_out/r06_base/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r06_base}; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-train-step --no-extra-sizes > $O/bench.json 2> $O/bench.err
bash scratch/mid_trace.sh 100000 > /dev/null 2>&1; cp gpurun_out/mid_trace_100000/timeline.txt $O/trace_100000.txt
bash scratch/prep_trace_n.sh 100000 > $O/prep_trace_100000.txt 2>&1
tail -c 1500 $O/bench.json
