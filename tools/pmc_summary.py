"""Summarise a rocprofv3 counter_collection.csv: per-kernel mean of each counter per dispatch."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
with open(path) as fh:
    for row in csv.DictReader(fh):
        k = row.get("Kernel_Name", "")
        if pat and not any(x in k for x in pat.split("|")):      # (alternatives: a|b)
            continue
        k = k.split("(")[0][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row.get("Dispatch_Id"))
for k, counters in acc.items():
    n = max(1, len(disp[k]))
    print("%s  dispatches=%d" % (k, n))
    for name, v in sorted(counters.items()):
        print("  %-28s %.6g per dispatch" % (name, v / n))
