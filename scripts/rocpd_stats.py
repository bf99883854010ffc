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
