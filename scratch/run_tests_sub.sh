#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/sub
timeout 1200 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -25 | cut -c1-250 | tee gpurun_out/sub/tests.txt
