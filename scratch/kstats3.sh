#!/bin/bash
# scratch/kstats3.sh <tag>: rocprofv3 kernel stats of 5 cached-layout config-3 (width 64, bf16 storage) forwards
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/scratch/run_config3_only.py > $out/log.txt 2>&1
python - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv",recursive=True)
if not f: print(open("$out/log.txt").read()[-2000:]); raise SystemExit
with open("$out/stats.txt","w") as o:
    for r in list(csv.DictReader(open(f[0])))[:18]:
        line=f'{r["Name"][:70].ljust(70)} calls {r["Calls"]:>5} avg_us {float(r["AverageNs"])/1e3:8.1f} total_us_per_fwd {float(r["TotalDurationNs"])/5e3:9.1f} pct {r["Percentage"]}'
        print(line); o.write(line+"\n")
PY
