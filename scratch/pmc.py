import csv, sys, collections
for f in sys.argv[2:]:
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(list)
    for r in rows:
        if sys.argv[1] in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        print(f"{k:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
