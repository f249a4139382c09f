#!/bin/bash
# kernel timeline of ONE cached-layout forward at a mid size (argument: nodes): per kernel start, duration, queue, gap to the previous
# kernel on the same queue -- where the ~45 us per layer go when the kernels themselves take 3-8 us
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
N=${1:-20000}; CAPA=${2:-0}; CAPB=${3:-0}
cat > /tmp/mt.py <<PY
import sys, time, torch, ctypes as C
sys.path.insert(0, '.')
from tilingnn_amd import TilinGNN, _lib
_lib.lib.tgnn_debug_set_block_caps($CAPA, $CAPB)
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device('cuda:0')
n = $N
net = TilinGNN(15, 20, 32, node_features_dim=3); net.load_state_dict(make_state_dict(15, 20, 32, 1, 3)); net = net.to(dev).train()
sg = make_super_graph(n, 10 * n, 12 * n + n // 2, tile_count=2, n_edge_types=13, seed=2)
x, adj, attr, col, _ = sg.to_torch(dev)
for _ in range(6): net(x, adj, attr, col)
torch.cuda.synchronize(); time.sleep(0.05)
net(x, adj, attr, col); torch.cuda.synchronize()
PY
rm -rf /tmp/mtr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/mtr -- python /tmp/mt.py > /tmp/mtr.log 2>&1
python - "$(find /tmp/mtr -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [int(r['Start_Timestamp']) for r in rows]; ends = [int(r['End_Timestamp']) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - max(ends[max(0, i - 40):i]) > 20_000_000: cut = i
sel = rows[cut:]
t0 = int(sel[0]['Start_Timestamp'])
last_end = {}
print(f"{len(sel)} kernels, span {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us")
for r in sel:
    q = r['Queue_Id']; s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  q{q:>2}  gap {gap:6.1f}  {r['Kernel_Name'][:60]}")
PY
