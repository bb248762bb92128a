#!/usr/bin/env python
"""Per-launch-shape durations of the kernels whose name contains a pattern (rocprofv3 rocpd database):
   python tools/rocpd_by_grid.py db pattern"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
g = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size", "workgroup_size_x", "grid_y", "grid_size_y")]
sel = ", ".join(g) if g else "0"
rows = db.execute(f"select name, {sel}, end - start from kernels where name like ?", (f"%{pat}%",)).fetchall()
agg = {}
for r in rows:
    k = (r[0][:60],) + tuple(r[1:-1])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += r[-1]
print("columns:", g)
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{str(k):110s} calls {n:4d} avg {t/n/1e3:9.1f} us")
