#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_dense_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_dense_$i -- python scratch/run_dense_only.py > gpurun_out/pmc_dense_$i.log 2>&1
  f=$(find gpurun_out/pmc_dense_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py dense_split $f
done
