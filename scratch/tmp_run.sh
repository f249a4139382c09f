for t in "" NN8; do
  if [ -n "$t" ]; then export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$t.so; fi
  echo "== ${t:-default}"; python scratch/time_100k.py 2>&1 | grep "^n "
done
