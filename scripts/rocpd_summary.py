"""summarise a rocprofv3 rocpd database (kernel trace) as a per-kernel table: calls, total, average, share"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
agg = {}
for name, s, e in rows:
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += e - s
total = sum(v[1] for v in agg.values()) or 1
print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:90]:90s} {n:7d} {t / 1e6:10.3f} {t / n / 1e3:10.2f} {100.0 * t / total:6.1f}%")
print(f"{'TOTAL':90s} {sum(v[0] for v in agg.values()):7d} {total / 1e6:10.3f}")

# bench.py runs: the same table restricted to bench.py's TIMED steps (argv[2] = --steps, argv[3] = --warmup): the frames between
# the (warmup+1)-th and the (warmup+steps+1)-th k_init_image -- what roofline.avg_launch_ms of the bench line is measured over
if len(sys.argv) > 3:
    steps, warmup = int(sys.argv[2]), int(sys.argv[3])
    rows.sort(key=lambda r: r[1])
    inits = [i for i, r in enumerate(rows) if "k_init_image" in r[0]]
    if len(inits) >= warmup + steps + 1:
        sel = rows[inits[warmup]:inits[warmup + steps]]
        for j, r in enumerate(sel):                     # rank 0 takes pt_stddev right after the timed loop
            if "k_stddev" in r[0]:
                sel = sel[:j]
                break
        agg = {}
        for name, s, e in sel:
            a = agg.setdefault(name, [0, 0])
            a[0] += 1; a[1] += e - s
        total = sum(v[1] for v in agg.values()) or 1
        print()
        print(f"timed region only: the {steps} steps after {warmup} warm-up steps (per step: {total / steps / 1e6:.3f} ms of kernel time)")
        print(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
        for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{name[:90]:90s} {n:7d} {t / 1e6:10.3f} {t / n / 1e3:10.2f} {100.0 * t / total:6.1f}%")
