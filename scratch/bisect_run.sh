#!/bin/bash
# runs tests/dist_gpu_worker.py 20000 20 in every bisect worktree and in the live tree (library RCCL on / off)
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
out=gpurun_out/bisect.txt; : > $out
port=29700
for t in scratch/bisect/878d603 scratch/bisect/446bd5b scratch/bisect/028fa06 scratch/bisect/ce05d07 .; do
  for lr in 1 0; do
    port=$((port+1))
    r=$(cd $t && MASTER_PORT=$port TGNN_LIBRARY_RCCL=$lr timeout 300 python tests/dist_gpu_worker.py 20000 20 2>/dev/null | grep '^OK' | tail -1)
    echo "$t rccl=$lr $r" | cut -c1-400 >> $out
  done
done
cat $out
