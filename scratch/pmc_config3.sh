#!/bin/bash
# rocprofv3 --pmc passes (kernel-trace only, one counter group per pass, each under its own timeout) of the config-3 forward
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_c3_$i
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_c3_$i -- python scratch/run_config3_only.py > /tmp/pmc_c3_$i.log 2>&1
  f=$(find /tmp/pmc_c3_$i -name "*counter_collection.csv" | head -1)
  for k in nnconv64_bf16_eg nnconv64_bf16_cols gin64_bf16_aggregate gin64_bf16_mlp_kernelILi3 bn_apply64_bf16 merge_bf16 dense_bf16_slots; do
    [ -n "$f" ] && python scratch/pmc.py $k $f | sed "s/^/$k  /"
  done
done
