#!/bin/bash
# kernel timeline of one forward_many call (three labyrinth-sized layouts, cached preparation)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/mt.py <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tests.test_hip_parity import make_net
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
net, _ = make_net(dev, depth=20)
g = load_labyrinth_graph()
layouts = [tuple(graph_tensors(g, torch.float32, dev)[:4])]
for n, seed in ((1254, 3), (1254, 4)):
    sg = make_super_graph(n, 8502, 10472, tile_count=2, n_edge_types=13, seed=seed)
    layouts.append(tuple(sg.to_torch(dev)[:4]))
for _ in range(4): net.forward_many(layouts)
torch.cuda.synchronize(); time.sleep(0.05)
net.forward_many(layouts); torch.cuda.synchronize()
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
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:8.1f} -> {(e - t0) / 1e3:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:60]}")
PY
