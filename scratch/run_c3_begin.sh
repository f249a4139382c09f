#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_bf16_path.py -m gpu -q -x 2>&1 | tail -3
for k in 1 2 3; do
  TGNN_BF16_BEGIN=0 timeout 200 python scratch/time_c3_lib.py 2>&1 | tail -1 | sed "s/^/no begin /"
  timeout 200 python scratch/time_c3_lib.py 2>&1 | tail -1 | sed "s/^/begin    /"
done
