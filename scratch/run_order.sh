#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2 3; do
for bf in 0 1; do
echo "TGNN_BEGIN_FIRST=$bf"; TGNN_BEGIN_FIRST=$bf AB_REPS=4 timeout 600 python scratch/ab_head.py 100000 3,3 2>&1 | grep "^mode" | tail -1
done; done
