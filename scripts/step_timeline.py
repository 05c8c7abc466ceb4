"""per-dispatch timeline of the last full bench step in a rocprofv3 kernel-trace database (rocpd):
python scripts/step_timeline.py <results.db> [steps back from the last one, default 3: bench.py ends with two counting frames]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if 'k_init_image' in r[0]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
i0, i1 = idx[-1 - back], idx[-back]
t0 = rows[i0][1]; prev = None
for name, st, en, g, wg in rows[i0:i1]:
    if 'stddev' in name: break
    short = name.split('(')[0].replace('void pt::', '').replace('pt::', '')[:34]
    gap = (st - prev) / 1e3 if prev else 0.0
    print(f"{short:36s} start {(st - t0) / 1e3:9.1f} us  dur {(en - st) / 1e3:8.1f} us  gap {gap:6.1f}  blocks {g // max(wg, 1)}")
    prev = en
