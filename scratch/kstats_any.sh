#!/bin/bash
# scratch/kstats_any.sh <tag> <python script + args>: rocprofv3 kernel stats of the command -> gpurun_out/<tag>/stats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python "$@" > $out/log.txt 2>&1 )
python - <<PY
import csv,glob
f=glob.glob("$out/**/*kernel_stats.csv",recursive=True)
if not f: print(open("$out/log.txt").read()[-2000:]); raise SystemExit
with open("$out/stats.txt","w") as o:
    for r in list(csv.DictReader(open(f[0])))[:24]:
        line=f'{r["Name"][:80].ljust(80)} calls {r["Calls"]:>5} avg_us {float(r["AverageNs"])/1e3:8.1f} min {float(r["MinNs"])/1e3:8.1f} pct {r["Percentage"]}'
        print(line); o.write(line+"\n")
PY
tail -12 $out/log.txt
