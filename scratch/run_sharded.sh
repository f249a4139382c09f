#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/sharded
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_dist_gpu.py tests/test_sharded_solve.py -m gpu -x -q -k "shard or nccl or dist" 2>&1 | tail -4 | cut -c1-200
for i in 1 2; do
MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 400 python bench.py --force-sharded --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-extra-sizes 2>/dev/null | tail -1 > gpurun_out/sharded/bench_sharded_world1.json
python -c "
import json; d=json.loads(open('gpurun_out/sharded/bench_sharded_world1.json').read()); print('sharded ms', d['ms_per_step'])"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train-step --no-extra-sizes 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('unsharded ms', d['ms_per_step'])"
done
