#!/usr/bin/env python
"""Turn a rocprofv3 `--kernel-trace --stats` result database (rocpd sqlite) into the CSV summary kept under profiles/.

    python profiles/summarize_rocprof.py gpurun_out/prof_xxx/name_results.db profiles/r01_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path: str, out_csv: str) -> None:
    con = sqlite3.connect(db_path)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds_bytes",
                    "grid_x", "workgroup_x"])
        for r in rows:
            w.writerow([r[0], r[1], round(r[2] / 1e3, 2), round(r[3] / 1e3, 2), round(r[4] / 1e3, 2), round(r[5] / 1e3, 2),
                        round(100.0 * r[2] / total, 2), r[6], r[7], r[8], r[9], r[10]])
    print(f"wrote {out_csv} ({len(rows)} kernels)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
