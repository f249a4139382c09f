#!/bin/bash
# kernel timeline of two layers of the sharded step at world 1 (side stream on)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/shtr.py <<'PY'
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
dist.init_process_group("nccl", rank=0, world_size=1)
from tilingnn_amd import TilinGNN
from tilingnn_amd.dist import ShardedTilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
sg = make_super_graph(100_000, 1_000_000, 1_250_000, tile_count=2, n_edge_types=13, seed=2)
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
runner = ShardedTilinGNN(net, sg, 0, 1, dev)
runner.fused.two_streams = sys.argv[1] == "1"
for _ in range(4): runner.step()
torch.cuda.synchronize(); time.sleep(0.05)
runner.step(); torch.cuda.synchronize()
dist.destroy_process_group()
PY
for two in 0 1; do
rm -rf gpurun_out/shtr$two; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/shtr$two -- python /tmp/shtr.py $two > gpurun_out/shtr$two.log 2>&1
f=$(find gpurun_out/shtr$two -name "*kernel_trace.csv" | head -1)
python - "$f" $two <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [int(r['Start_Timestamp']) for r in rows]; ends = [int(r['End_Timestamp']) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - max(ends[max(0, i - 60):i]) > 20_000_000: cut = i
sel = rows[cut:]
t0 = int(sel[0]['Start_Timestamp'])
print("== two_streams", sys.argv[2], "kernels", len(sel), "total us", (max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3)
idx = [i for i, r in enumerate(sel) if 'nnconv32_' in r['Kernel_Name']]
a, b = idx[8], idx[10]
prev_end = int(sel[a]['Start_Timestamp'])
for r in sel[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:56]}")
    prev_end = max(prev_end, e)
PY
done
