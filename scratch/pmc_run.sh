#!/bin/bash
# usage: pmc_run.sh <tag> <kernel-substr> ; PMC passes (kernel-trace only), each wrapped in timeout
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; kern=$2
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${tag}_$i -- python scratch/run_nnconv_only.py nnconv > gpurun_out/pmc_${tag}_$i.log 2>&1
  f=$(find gpurun_out/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py $kern $f
done
