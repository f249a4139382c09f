#!/bin/bash
# scratch/ab_libs.sh <script + args> -- <libtags...>: runs the script once per library build (default first)
cd $GRAFT_REPO_ROOT
cmd=()
while [ "$1" != "--" ]; do cmd+=("$1"); shift; done; shift
for lt in default "$@"; do
  if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
  echo "== $lt"; timeout 300 python "${cmd[@]}" 2>&1 | grep -v amdgpu.ids | tail -3
done
