"""Where does a fine-tune micro-step's wall time go?  Reads the rocpd database of `rocprofv3 --kernel-trace -- python bench.py --mode ft ...`
and prints, for the last WINDOW ms of the trace (the timed region), per stream: dispatches, the sum of kernel durations, the gaps between
consecutive dispatches (end -> next start) as a histogram, and the union of all kernels' execution intervals (the time the GPU ran anything).
usage: python scripts/ft_timeline.py RESULTS.db [WINDOW_MS]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 120e6
cur = db.cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view') and name like 'rocpd_kernel_dispatch%'")][0]
rows = list(cur.execute(f"select stream_id, queue_id, start, end from {t} order by start"))
t1 = max(r[3] for r in rows)
rows = [r for r in rows if r[2] >= t1 - win]
t0 = rows[0][2]
by = {}
for sid, q, s, e in rows:
    by.setdefault((sid, q), []).append((s, e))
ev = sorted([(s, 1) for _, _, s, e in rows] + [(e, -1) for _, _, s, e in rows])
busy, depth, last = 0, 0, None
for ts, d in ev:
    if depth > 0:
        busy += ts - last
    depth += d
    last = ts
print(f"window: last {(t1 - t0) / 1e6:.1f} ms, {len(rows)} dispatches on {len(by)} (stream, queue) pairs; some kernel executing during {100 * busy / (t1 - t0):.0f} % of it")
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    ktime = sum(e - s for s, e in v)
    gaps = [max(0, v[i + 1][0] - v[i][1]) for i in range(len(v) - 1)]
    hist = [0] * 7
    for g in gaps:
        hist[0 if g < 2e3 else 1 if g < 5e3 else 2 if g < 10e3 else 3 if g < 20e3 else 4 if g < 50e3 else 5 if g < 200e3 else 6] += 1
    big = sum(g for g in gaps if g >= 20e3)
    print(f"stream {k[0]} (queue {k[1]}): {len(v)} dispatches, kernel time {ktime / 1e6:.1f} ms = {100 * ktime / (t1 - t0):.0f} % of the window; gaps {sum(gaps) / 1e6:.1f} ms "
          f"({big / 1e6:.1f} ms in gaps >= 20 us); gaps <2 / 2-5 / 5-10 / 10-20 / 20-50 / 50-200 / >200 us: {hist}")
