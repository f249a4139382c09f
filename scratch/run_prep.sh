#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/prep
timeout 900 python -m pytest tests/test_running_stats.py tests/test_hip_parity.py -m gpu -x -q -k "running or csr or prep or dedup or group or structure or full_size or drop_in" 2>&1 | tail -4 | cut -c1-200
AB_REPS=6 timeout 600 python scratch/ab_head.py 100000 0,3 2>&1 | grep -v amdgpu | tail -3
bash scratch/step_trace.sh 100000 prep_step > /dev/null 2>&1; head -34 gpurun_out/prep_step/step_timeline.txt | cut -c1-110
