#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/suite
( time timeout 1500 python -m pytest tests -m gpu -q --durations=30 ) > gpurun_out/suite/log.txt 2>&1
tail -60 gpurun_out/suite/log.txt | cut -c1-200
