#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/head
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_gin_fused.py -m gpu -x -q 2>&1 | tail -5 | cut -c1-220 | tee gpurun_out/head/tests.txt
AB_REPS=6 timeout 600 python scratch/ab_head.py 100000 0,3 2>&1 | grep -v amdgpu | tee gpurun_out/head/ab.txt
bash scratch/step_trace.sh 100000 head_step > /dev/null 2>&1; head -40 gpurun_out/head_step/step_timeline.txt | cut -c1-110
