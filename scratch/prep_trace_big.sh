#!/bin/bash
# kernel timeline of one prepare_graph at the benchmark shape (100k nodes)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/ptraceb; timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ptraceb -- python scratch/run_prep_only.py > gpurun_out/ptraceb.log 2>&1
f=$(find gpurun_out/ptraceb -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# last prepare_graph = from the last csr_count pair backwards: take the last 40 kernels
idx = [i for i, nm in enumerate(names) if 'csr_count_kernel' in nm]
start = idx[-2] - 4
sel = rows[start:]
t0 = int(sel[0]['Start_Timestamp']); prev_end = t0
for r in sel:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end, e)
print(f"total {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us, {len(sel)} kernels")
PY
