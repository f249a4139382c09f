#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in 1 2 3; do
  timeout 200 python scratch/time_fwd_lib.py 2>&1 | tail -1
  TGNN_LIB_PATH=$PWD/scratch/libs/libtgnn_EGW8.so timeout 200 python scratch/time_fwd_lib.py 2>&1 | tail -1
done
TGNN_LIB_PATH=$PWD/scratch/libs/libtgnn_EGW8.so timeout 250 python -m pytest tests/test_nnconv_eg.py -m gpu -q -x 2>&1 | tail -3
