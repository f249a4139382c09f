#!/bin/bash
# HBM bytes of the WHOLE cached-layout forward at the benchmark shape, per kernel: FETCH_SIZE and WRITE_SIZE in separate
# kernel-trace-only passes (3 forwards each; the counters serialise the kernels, so these are bytes, not times).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pf_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pf_$c -- python scratch/run_fwd_mode.py 1 3 > /tmp/pf_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pf_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    tot[c] = agg
names = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda k: -(2 * tot["FETCH_SIZE"][k][0] + tot["WRITE_SIZE"][k][0]))
n_fw = 3
print(f"{'kernel':50s} launches/fwd   read MB/fwd  write MB/fwd   (FETCH_SIZE x 2 on gfx950: 32-byte units counted as 64; KB -> MB)")
sr = sw = 0.0
for k in names:
    fr, nf = tot["FETCH_SIZE"][k]; wr, nw = tot["WRITE_SIZE"][k]
    r_mb, w_mb = 2 * fr * 1024 / n_fw / 1e6, wr * 1024 / n_fw / 1e6
    if "at::" in k or "elementwise" in k or "spin" in k: continue
    sr += r_mb; sw += w_mb
    print(f"{k:50s} {max(nf, nw) / n_fw:8.1f}   {r_mb:10.1f}  {w_mb:10.1f}")
print(f"{'whole forward':50s}            {sr:10.1f}  {sw:10.1f}   total {sr + sw:.1f} MB against 2 887 MB algorithmic (SURVEY 8d)")
PY
