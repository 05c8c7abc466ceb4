"""per-kernel mean of every counter in rocprofv3 --pmc CSV outputs (one or more directories).

Launches are keyed on (kernel, grid size): launches of one kernel with different grids are different populations (the
one-ray autofocus launch and the light-grid set-up launches of k_trace against the batched ones) and must not share a mean
(round 1's raybench summary mixed them).  --min-grid N drops groups whose grid is smaller than N work-items;
--only SUBSTR keeps kernels whose name contains SUBSTR."""
import csv, glob, sys
from collections import defaultdict

args = sys.argv[1:]
min_grid, only, dirs = 0, None, []
while args:
    a = args.pop(0)
    if a == "--min-grid": min_grid = int(args.pop(0))
    elif a == "--only": only = args.pop(0)
    else: dirs.append(a)
acc = defaultdict(lambda: [0, 0.0])
seen = set()
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][:60]
            grid = int(r["Grid_Size"])
            if grid < min_grid or (only and only not in name):
                continue
            key = (f, r["Dispatch_Id"], r["Counter_Name"])
            if key in seen:                      # one row per dispatch and counter
                continue
            seen.add(key)
            k = (name, grid, r["Counter_Name"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for (kern, grid, ctr), (n, v) in sorted(acc.items()):
    print(f"{kern:62s} grid {grid:9d} {ctr:34s} launches {n:4d}  mean {v / n:16.1f}")
