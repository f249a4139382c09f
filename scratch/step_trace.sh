#!/bin/bash
# kernel + copy timeline of ONE uncached step (graph preparation + forward, what bench.py times) at <n_nodes>:
# scratch/step_trace.sh <n_nodes> [out-name]  -> gpurun_out/<out-name>/step_timeline.txt
n=${1:-100000}; name=${2:-step_trace_$n}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cat > /tmp/run_step.py <<PY
import sys, time, torch
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
net.cache_graph = False
import os
if os.environ.get("LEAN"):
    from tilingnn_amd._lib import lib
    lib.tgnn_set_lean_head(int(os.environ["LEAN"]))
for _ in range(8):
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    net(x=x, adj_e_index=adj, adj_e_features=attr, col_e_idx=col)
torch.cuda.synchronize()
print("ms_per_step", (time.perf_counter() - t0) / 20 * 1e3)
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o trace -- python /tmp/run_step.py > $out/log.txt 2>&1
grep ms_per_step $out/log.txt
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), r["Kernel_Name"]))
for f in glob.glob("$out/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "cp", "COPY " + r.get("Direction", "")))
rows.sort()
# the last complete step: from the second-to-last "result" fill in front of a dedup/bk_hist to the one after it
starts = [i for i, r in enumerate(rows) if "bk_hist" in r[3]]
a, b = starts[-2], starts[-1]
# walk back from bk_hist to the step's first op (fills / dedup in front of it, at most 6 entries, < 80 us earlier)
def first_of(i):
    j = i
    while j > 0 and rows[i][0] - rows[j - 1][0] < 80_000 and ("fillBuffer" in rows[j - 1][3] or "dedup" in rows[j - 1][3] or "prep_init" in rows[j - 1][3]):
        j -= 1
    return j
a, b = first_of(a), first_of(b)
sel = rows[a:b]
t0 = sel[0][0]
with open("$out/step_timeline.txt", "w") as o:
    prev_end = t0
    for s, e, q, name in sel:
        o.write(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} us  gap {(s-prev_end)/1e3:6.1f}  {q:4s} {name[:100]}\n")
        prev_end = max(prev_end, e)
    o.write(f"step period (start to next step's start): {(rows[b][0]-t0)/1e3:.1f} us\n")
print(open("$out/step_timeline.txt").read())
PY
