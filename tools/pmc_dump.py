#!/usr/bin/env python
"""rocprofv3 --pmc <COUNTER> results.db -> JSON {kernel name: {"sum": counter total, "dispatches": n, "avg_us": ..}}, one file per
counter pass (FETCH_SIZE, WRITE_SIZE): python tools/pmc_dump.py db COUNTER out.json"""
import json, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); counter = sys.argv[2]
rows = db.execute("select name, sum(counter_value), count(distinct dispatch_id), avg(duration) from pmc_events where counter_name = ? group by name",
                  (counter,)).fetchall()
out = {}
for name, tot, n, dur in rows:
    k = re.sub(r"^void ", "", name)
    k = re.sub(r"\(.*$", "", k).replace("cbim::", "").replace("(anonymous namespace)::", "")
    e = out.setdefault(k, {"sum": 0.0, "dispatches": 0, "dur_ns": 0.0})
    e["sum"] += float(tot); e["dispatches"] += int(n); e["dur_ns"] += float(dur) * int(n)
json.dump(out, open(sys.argv[3], "w"), indent=0, sort_keys=True)
print(counter, len(out), "kernels", sum(v["dispatches"] for v in out.values()), "dispatches")
