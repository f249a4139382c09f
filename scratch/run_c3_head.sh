#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in 1 2 3; do
  TGNN_LIB_PATH=$PWD/scratch/libs/libtgnn_C3SERIALHEAD.so timeout 200 python scratch/time_c3_lib.py 2>&1 | tail -1
  timeout 200 python scratch/time_c3_lib.py 2>&1 | tail -1
done
timeout 250 python -m pytest tests/test_bf16_path.py -m gpu -q -x 2>&1 | tail -3
