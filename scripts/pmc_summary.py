"""per-kernel mean of every counter in rocprofv3 --pmc CSV outputs (one or more directories)"""
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kern, ctr), (n, v) in sorted(acc.items()):
    print(f"{kern:62s} {ctr:24s} launches {n:4d}  mean {v / n:16.1f}")
