#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -f /tmp/fold_*.pt
TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_NOFOLD.so python scratch/check_fold.py /tmp/fold 2>&1 | grep -v amdgpu
python scratch/check_fold.py /tmp/fold 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gin_fused.py tests/test_hip_parity.py -m gpu -x -q -k "gin or full_size or reproduc or coll" 2>&1 | tail -3 | cut -c1-200
bash scratch/gin_fold_abl.sh 2>&1 | tail -16
