#!/bin/bash
# kernel trace of a few cached-layout forwards at a mid size: scratch/mid_trace.sh <n_nodes> -> gpurun_out/mid_trace_<n>/
n=${1:-10000}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/mid_trace_$n
mkdir -p $out
cat > /tmp/run_mid.py <<PY
import sys, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from tilingnn_amd import TilinGNN
from tilingnn_amd.synth import make_super_graph
from tilingnn_amd.weights import make_state_dict
dev = torch.device("cuda:0")
n = $n
ea, ec = (8 * n, 10 * n) if n == 10000 else (10 * n, 12 * n + n // 2)
sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
net = TilinGNN(adj_edge_features_dim=15, network_depth=20, network_width=32, node_features_dim=3)
net.load_state_dict(make_state_dict(15, 20, 32, 1, 3, seed=0), strict=True)
net = net.to(dev).train()
for _ in range(8):
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python /tmp/run_mid.py > $out/log.txt 2>&1
python - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last forward: from the last forward_scales launch on
idx = [i for i, r in enumerate(rows) if "forward_scales" in r["Kernel_Name"]]
rows = rows[idx[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
with open("$out/timeline.txt", "w") as o:
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        o.write(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?')}  {r['Kernel_Name'][:90]}\n")
print(open("$out/timeline.txt").read())
PY
