"""rocprofv3 (rocpd sqlite output) -> per-kernel summary CSV:  python profiles/summarize_rocprof.py <results.db> [out.csv]
Columns: kernel, calls, total_us, avg_us, min_us, max_us, percent (of the summed kernel time)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels "
                  "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
for name, n, s, a, lo, hi in rows:
    lines.append(f"\"{name}\",{n},{s / 1e3:.2f},{a / 1e3:.2f},{lo / 1e3:.2f},{hi / 1e3:.2f},{100.0 * s / tot:.2f}")
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
