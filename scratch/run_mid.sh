#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_graph_prep_small.py -m gpu -x -q -k "csr or prep or dedup or group or column or structure" 2>&1 | tail -2 | cut -c1-200
timeout 300 python scratch/time_mid.py 2>&1 | grep "^n"
bash scratch/prep_trace_n.sh 10000 2>&1 | tail -18
