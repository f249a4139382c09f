#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/head
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "sharded_hip_path" 2>&1 | grep "world\|passed\|failed" | cut -c1-220 | tee gpurun_out/head/tests.txt
timeout 600 python scratch/ab_head.py 100000 2>&1 | tee gpurun_out/head/ab.txt
