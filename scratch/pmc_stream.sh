#!/bin/bash
# usage: pmc_stream.sh <tag> [cols|stream] [kernel-name-substring]: SQ counters of one NNConv kernel (kernel-trace only, separate passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; which=${2:-stream}; kn=${3:-stream_kernel}
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${tag}_$i -- python scratch/run_stream_only.py $which > /tmp/pmc_${tag}_$i.log 2>&1
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py $kn $f || tail -3 /tmp/pmc_${tag}_$i.log
done
