#!/usr/bin/env python
"""Average rocprofv3 --pmc counters per kernel from a counter_collection CSV (rocprofv3 --output-format csv).
usage: scripts/pmc_summary.py <counter_collection.csv> [substring filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in rows:
    k = r.get("Kernel_Name", "")
    if flt and flt not in k:
        continue
    k = k[:90]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
for k in acc:
    parts = [f"{c}={acc[k][c] / cnt[k][c]:.4g}" for c in sorted(acc[k])]
    n = max(cnt[k].values())
    print(f"{k} [dispatches {n}]: " + " ".join(parts))
