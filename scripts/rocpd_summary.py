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
