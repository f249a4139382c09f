#!/bin/bash
# usage: pmc_dense.sh <tag> <N>: SQ counters of dense_f16_rows_kernel (kernel-trace only passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; n=$2
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_$i -- python scratch/run_dense_rows_only.py $n > gpurun_out/pmc_${tag}_$i.log 2>&1
  f=$(find gpurun_out/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py dense_f16_rows $f
done
