#!/bin/bash
# usage: pmc_ps.sh <tag> <kernel-name-substring> <ps|cols_f16>: SQ counters of one NNConv kernel (kernel-trace only, separate passes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; kname=$2; which=$3
mkdir -p gpurun_out/$tag
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_${tag}_$i
  timeout 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${tag}_$i -- python scratch/run_ps_only.py $which > /tmp/pmc_${tag}_$i.log 2>&1
  f=$(find /tmp/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python scratch/pmc.py $kname $f | tee -a gpurun_out/$tag/pmc.txt || tail -3 /tmp/pmc_${tag}_$i.log
done
