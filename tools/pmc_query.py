#!/usr/bin/env python
"""Average PMC counters of kernels whose name contains a pattern: python tools/pmc_query.py db pattern [min_us]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; mn = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 0
rows = db.execute("select counter_name, sum(counter_value), count(distinct dispatch_id), avg(duration) from pmc_events "
                  "where name like ? and duration > ? group by counter_name", (f"%{pat}%", mn)).fetchall()
for r in rows:
    print(f"{r[0]:32s} total/dispatch {r[1]/r[2]:.4g}   dispatches {r[2]}  avg {r[3]/1e3:.1f} us")
