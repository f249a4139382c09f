#!/bin/bash
# the bench line's roofline.avg_launch_us (device-stamped, in the production forward) beside rocprofv3's average for the same command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/bvr
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bvr -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-step --no-extra-sizes > gpurun_out/bvr_bench.json 2> /tmp/bvr.err
f=$(find /tmp/bvr -name "*kernel_stats.csv" | head -1)
python scratch/kstats.py "$f" 12
python - <<'PY'
import json
for l in open('gpurun_out/bvr_bench.json'):
    l = l.strip()
    if l.startswith('{"metric"'):
        r = json.loads(l)["roofline"]
        print("bench: stamped in-forward", round(r["avg_launch_us"], 1), "us frac", round(r["frac"], 4), "| single-stream events", round(r["single_stream"]["avg_launch_us"], 1),
              "| two-stream events", round(r.get("in_forward_events", {}).get("avg_launch_us", 0), 1), "| ms/step", round(json.loads(l)["ms_per_step"], 3))
PY
