#!/bin/bash
# kernel timeline of one prepare_graph of a synthetic layout: scratch/prep_trace_n.sh <n_nodes>  -> stdout
n=${1:-10000}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/prep_n.py <<PY
import sys, time, torch
sys.path.insert(0, '.')
from tilingnn_amd import ops
from tilingnn_amd.synth import make_super_graph
dev = torch.device('cuda:0')
n = $n
ea, ec = (8 * n, 10 * n) if n == 10000 else (10 * n, 12 * n + n // 2)
sg = make_super_graph(n, ea, ec, tile_count=2, n_edge_types=13, seed=1)
x, adj, attr, col, _ = sg.to_torch(dev)
for _ in range(5): ops.prepare_graph(n, adj, attr, col)
torch.cuda.synchronize(); time.sleep(0.05)
ts = []
for _ in range(3):
    t = time.perf_counter(); ops.prepare_graph(n, adj, attr, col); torch.cuda.synchronize(); ts.append(time.perf_counter() - t); time.sleep(0.02)
print("wall us", [round(t * 1e6, 1) for t in ts])
PY
rm -rf /tmp/ptrace_n; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/ptrace_n -- python /tmp/prep_n.py 2>&1 | grep "wall us"
f=$(find /tmp/ptrace_n -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [int(r['Start_Timestamp']) for r in rows]; ends = [int(r['End_Timestamp']) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - max(ends[max(0, i - 50):i]) > 5_000_000: cut = i
sel = rows[cut:]
t0 = int(sel[0]['Start_Timestamp']); prev_end = t0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
print(f"total {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us, {len(sel)} kernels")
PY
