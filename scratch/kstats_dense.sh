#!/bin/bash
# scratch/kstats_dense.sh <tag>: per-kernel times of scratch/time_dense_rows.py
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/scratch/time_dense_rows.py > $out/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(f[0])))
import collections
d=collections.defaultdict(list)
for r in rows:
    nm=r["Kernel_Name"]
    if "dense" in nm: d[(nm[:60], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
with open("$out/stats.txt","w") as o:
    for k,v in sorted(d.items()):
        v=sorted(v); line=f"{k[0].ljust(60)} grid {k[1]:>8} calls {len(v):>4} median_us {v[len(v)//2]:8.1f} min {v[0]:8.1f}"
        print(line); o.write(line+"\n")
PY
tail -12 $out/log.txt
