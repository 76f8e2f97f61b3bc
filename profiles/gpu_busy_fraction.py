"""From a rocprofv3 kernel trace (rocpd sqlite): fraction of the traced span with at least one kernel running, and the average number
of kernels in flight.   python profiles/gpu_busy_fraction.py <results.db> [first_fraction last_fraction]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1.0)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
tab = "kernels" if "kernels" in tabs else next(t for t in tabs if "kernel_dispatch" in t)
rows = sorted(db.execute(f"select start, end, name from {tab}").fetchall())
t0, t1 = rows[0][0], max(r[1] for r in rows)
a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
rows = [(max(s, a), min(e, b), n) for s, e, n in rows if e > a and s < b]
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
busy = 0.0; area = 0.0; depth = 0; prev = a
for t, d in ev:
    if depth > 0: busy += t - prev
    area += depth * (t - prev)
    depth += d; prev = t
span = b - a
print(f"span {span / 1e6:.2f} ms, {len(rows)} kernels: busy {busy / span:.3f}, mean kernels in flight {area / span:.2f}")
by = {}
for s, e, n in rows: by[n.split('(')[0][:40]] = by.get(n.split('(')[0][:40], 0) + (e - s)
for n, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]: print(f"   {v / span:6.3f} of the span  {n}")
