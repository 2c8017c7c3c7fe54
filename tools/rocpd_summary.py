#!/usr/bin/env python3
"""Export judge-readable CSV summaries from rocprofv3's rocpd SQLite output.

    python tools/rocpd_summary.py stats  <results.db> <out.csv>     # per kernel x grid: calls, avg/min/max us
    python tools/rocpd_summary.py pmc    <results.db> <out.csv>     # per kernel x grid x counter: avg/min/max value
"""
import csv
import sqlite3
import sys


def main():
    mode, db_path, out = sys.argv[1:4]
    db = sqlite3.connect(db_path)
    c = db.cursor()
    if mode == "stats":
        rows = c.execute("""select name, grid_x, grid_y, workgroup_x, count(*), avg(duration)/1e3, min(duration)/1e3,
                                   max(duration)/1e3, sum(duration)/1e3, max(vgpr_count), max(sgpr_count), max(lds_size)
                            from kernels group by name, grid_x, grid_y, workgroup_x order by sum(duration) desc""").fetchall()
        hdr = ["kernel", "grid_x", "grid_y", "workgroup_x", "calls", "avg_us", "min_us", "max_us", "total_us", "vgpr", "sgpr", "lds_bytes"]
    else:
        rows = c.execute("""select kernel_name, grid_size_x, grid_size_y, counter_name, count(*), avg(value), min(value), max(value)
                            from counters_collection group by kernel_name, grid_size_x, grid_size_y, counter_name
                            order by kernel_name, grid_size_x""").fetchall()
        hdr = ["kernel", "grid_x", "grid_y", "counter", "dispatches", "avg", "min", "max"]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(hdr)
        for r in rows:
            w.writerow([f"{v:.3f}" if isinstance(v, float) else v for v in r])
    print(f"{out}: {len(rows)} rows")


if __name__ == "__main__":
    main()
