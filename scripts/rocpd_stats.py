#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace: per-kernel calls / total / mean.
usage: scripts/rocpd_stats.py <results.db> [out.csv]"""
import csv
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
for name, calls, tot, avg, pct in rows:
    w.writerow([name[:160], calls, f"{tot:.1f}", f"{avg:.3f}", f"{pct:.3f}"])  # top_kernels view is in microseconds

# optional third argument: the same trace grouped by (kernel, grid, workgroup, dynamic LDS) -- one row per launch geometry,
# which separates most layer shapes that share a kernel template
if len(sys.argv) > 3:
    rows = list(con.execute(
        "select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration), sum(duration) "
        "from kernels group by name, grid_x, grid_y, grid_z, workgroup_x, lds_size order by sum(duration) desc limit 60"))
    with open(sys.argv[3], "w", newline="") as f:
        w2 = csv.writer(f)
        w2.writerow(["kernel", "grid", "workgroup", "lds_bytes", "calls", "avg_us", "min_us", "max_us", "total_us"])
        for name, gx, gy, gz, wx, lds, n, avg, mn, mx, tot in rows:   # the kernels view is in nanoseconds
            w2.writerow([name[:120], f"{gx}x{gy}x{gz}", wx, lds, n, f"{avg / 1e3:.3f}", f"{mn / 1e3:.3f}", f"{mx / 1e3:.3f}", f"{tot / 1e3:.1f}"])
