#!/bin/bash
# kernel timeline of the last labyrinth forward of scratch/small_trace.py (rocprofv3 kernel trace): start offset, duration, gap
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/strace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/strace -- python scratch/small_trace.py > gpurun_out/strace.log 2>&1
f=$(find gpurun_out/strace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last forward = the kernels after the last long idle gap
starts = [int(r['Start_Timestamp']) for r in rows]; ends = [int(r['End_Timestamp']) for r in rows]
cut = 0
for i in range(1, len(rows)):
    if starts[i] - max(ends[:i][-50:]) > 5_000_000: cut = i
sel = rows[cut:]
t0 = int(sel[0]['Start_Timestamp']); prev_end = t0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?'):>2s}  {r['Kernel_Name'][:60]}")
    prev_end = max(prev_end, e)
print(f"total {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us, {len(sel)} kernels")
PY
