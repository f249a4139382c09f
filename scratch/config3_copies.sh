#!/bin/bash
# where do the __amd_rocclr_copyBuffer launches of a config-3 run come from?  kernel trace of scratch/run_config3_only.py, copies counted
# before / after the first forward's first kernel -> gpurun_out/<tag>/copies.txt + kernel stats of the 5 forwards
tag=${1:-r5_c3}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python scratch/run_config3_only.py > $out/log.txt 2>&1 )
python - <<PY | tee $out/copies.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "edge_weight" in r["Kernel_Name"] or "nnconv64" in r["Kernel_Name"] or "forward_scales" in r["Kernel_Name"] or "csr" in r["Kernel_Name"].lower() or "bk_hist" in r["Kernel_Name"])
cp_before = sum("copyBuffer" in r["Kernel_Name"] for r in rows[:first])
cp_after = sum("copyBuffer" in r["Kernel_Name"] for r in rows[first:])
print(f"__amd_rocclr_copyBuffer launches: {cp_before} before the first kernel of the library (model upload: .to(device) of 664 state-dict tensors, inputs), {cp_after} after it (5 forwards)")
t0 = int(rows[first]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"the 5 forwards incl. one graph preparation: {(t1 - t0) / 1e3:.0f} us of trace")
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[first:]:
    a = agg[r["Kernel_Name"][:70]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{k:70s} calls {c:4d} avg_us {t / c:7.1f} us_per_forward {t / 5:8.1f}")
PY
