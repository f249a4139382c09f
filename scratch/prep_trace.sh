#!/bin/bash
# kernel timeline of one prepare_graph of the labyrinth layout
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/prep_small.py <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from tests.golden_util import graph_tensors, load_labyrinth_graph
from tilingnn_amd import ops
dev = torch.device('cuda:0')
x, adj, attr, col, _ = graph_tensors(load_labyrinth_graph(), torch.float32, dev)
for _ in range(5): ops.prepare_graph(1254, adj, attr, col)
torch.cuda.synchronize(); time.sleep(0.05)
t = time.perf_counter()
for _ in range(3):
    ops.prepare_graph(1254, adj, attr, col); torch.cuda.synchronize(); time.sleep(0.02)
PY
rm -rf gpurun_out/ptrace; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ptrace -- python /tmp/prep_small.py > gpurun_out/ptrace.log 2>&1
f=$(find gpurun_out/ptrace -name "*kernel_trace.csv" | head -1)
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
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
print(f"total {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us, {len(sel)} kernels")
PY
