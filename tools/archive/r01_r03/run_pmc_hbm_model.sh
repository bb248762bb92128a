#!/bin/bash
# Achieved HBM bandwidth per kernel of one model step: gpurun -- bash tools/run_pmc_hbm_model.sh medformer
R=$GRAFT_REPO_ROOT; M=${1:-medformer}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pb_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pb_$c -o f -- python $R/bench.py --model $M --steps 2 --warmup 1 --graph 0 --no-roofline --no-cpu-baseline > /dev/null 2>&1
done
python - <<PY
import sqlite3, re
def q(c):
    db = sqlite3.connect(f"/tmp/pb_{c}/f_results.db")
    return {r[0]: (r[1], r[2], r[3]) for r in db.execute("select name, sum(counter_value), count(distinct dispatch_id), sum(duration) from pmc_events where counter_name=? group by name", (c,))}
f, w = q("FETCH_SIZE"), q("WRITE_SIZE")
rows = []
for k in f:
    if k in w:
        kb = 2 * f[k][0] + w[k][0]          # gfx950: FETCH_SIZE counts 128-B requests as 64 B
        dur = (f[k][2] / max(f[k][1],1) + w[k][2] / max(w[k][1],1)) / 2   # avg ns per dispatch... careful: duration summed per counter row
        rows.append((f[k][2], k, f[k][1], kb * 1024 / f[k][1], ))
rows.sort(reverse=True)
print("kernel | calls | MB/launch | avg us | GB/s")
for tot, k, n, b in rows[:28]:
    us = tot / n / 1e3
    print(f"{re.sub(r'cbim::', '', k)[:70]:70s} {n:5d} {b/1e6:9.1f} {us:9.1f} {b/us/1e3:8.0f}")
PY
