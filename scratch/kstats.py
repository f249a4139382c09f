import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 14]:
    print(r['Name'][:52].ljust(52), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e3:10.1f}us", f"avg {float(r['AverageNs'])/1e3:8.1f}us", f"min {float(r['MinNs'])/1e3:7.1f}", r['Percentage'])
print("total", tot/1e3, "us")
