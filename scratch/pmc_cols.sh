#!/bin/bash
# usage: pmc_cols.sh <tag>: SQ / SQC counters of the NNConv column kernel (kernel-trace only, separate passes, own timeouts)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${tag}_$i -- python scratch/time_nnconv.py > /tmp/pmc_${tag}_$i.log 2>&1
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py cols_kernel $f || tail -3 /tmp/pmc_${tag}_$i.log
done
