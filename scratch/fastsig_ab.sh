#!/bin/bash
# Same-box A/B of the collision MLP's output sigmoid: libm expf + IEEE division (default) against exp2 on a two-part product + one
# Newton step on the hardware reciprocal (scratch/libs/libtgnn_FASTSIG.so = gin.hip with -DTGNN_ABL_FASTSIG)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/fastsig; mkdir -p $O; S=$O/summary.txt; : > $S
for rep in 1 2; do
  for lt in default FASTSIG; do
    if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
    echo "== $lt (run $rep)" >> $S
    timeout 300 python scratch/lib_ab.py $O/out_$lt.pt 100000 2>&1 | grep -v amdgpu.ids | tail -1 >> $S
  done
done
for lt in default FASTSIG; do
  if [ $lt = default ]; then unset TGNN_LIB_PATH; else export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_$lt.so; fi
  echo "== kernels inside the forward, $lt" >> $S
  bash scratch/kstats_any.sh fastsig_$lt scratch/run_fwd_groups.py groups > /dev/null 2>&1
  head -8 gpurun_out/fastsig_$lt/stats.txt >> $S
done
python - >> $S <<'PY'
import torch
a, b = torch.load('gpurun_out/fastsig/out_default.pt'), torch.load('gpurun_out/fastsig/out_FASTSIG.pt')
for n in a:
    d = (a[n][0].double() - b[n][0].double()).abs()
    print(f"n {n}: probabilities default vs FASTSIG: max abs diff {float(d.max()):.3e} (max p {float(a[n][0].max()):.3f})")
PY
export TGNN_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libtgnn_FASTSIG.so
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "gin or coll or forward" 2>&1 | tail -4 >> $S
cat $S
