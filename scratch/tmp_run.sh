for t in ROWS_NOLOOP ROWS_NOMEM ROWS_NOSPLIT; do
  export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$t.so
  echo "== $t"; bash scratch/kstats_dense.sh r4f_dense_$t 2>&1 | grep "rows_kernel"
done
