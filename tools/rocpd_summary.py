#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into a per-kernel table:
   python tools/rocpd_summary.py gpurun_out/prof/x_results.db [steps] > profiles/rNN_xxx.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("cbim::", "").replace("(anonymous namespace)::", "")
    if len(name) > 90:
        name = name[:87] + "..."
    a = agg.setdefault(name, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
print(f"# {len(rows)} dispatches, total kernel time {tot/1e6:.3f} ms over {steps} traced step(s) (+warm-up) ")
print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:92s} {n:7d} {t/1e6:10.3f} {t/n/1e3:10.2f} {100*t/tot:6.2f}")
