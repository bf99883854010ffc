#!/usr/bin/env python
"""Idle time between consecutive kernel dispatches of a rocprofv3 trace (rocpd sqlite): for the steady part of the trace,
the mean kernel duration per name and the mean gap from the end of one dispatch to the start of the next.
usage: scripts/rocpd_gaps.py <results.db> [substring of the kernel names to keep, default mpm_]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "mpm_"
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = list(con.execute(f"select name, {start}, {end} from kernels order by {start}"))
rows = [r for r in rows if pat in r[0]]
rows = rows[len(rows) // 4:]      # skip the set-up / warm-up quarter
dur, gaps = {}, {}
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    key = n0.split("(")[0][:50]
    dur.setdefault(key, []).append((e0 - s0) / 1e3)
    gaps.setdefault(key + " -> " + n1.split("(")[0][:30], []).append((s1 - e0) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -len(kv[1]))[:4]:
    print(f"{k:52s} n={len(v):5d}  mean duration {sum(v) / len(v):7.2f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:4]:
    v2 = sorted(v)
    print(f"gap {k:80s} n={len(v):5d}  mean {sum(v) / len(v):6.2f} us  median {v2[len(v2) // 2]:6.2f} us")
span = (rows[-1][2] - rows[0][1]) / 1e3
busy = sum((e - s_) / 1e3 for _, s_, e in rows)
allgaps = [(b[1] - a[2]) / 1e3 for a, b in zip(rows, rows[1:])]
pos = sorted(g for g in allgaps if g > 0)
print(f"span {span:.1f} us over {len(rows)} dispatches; kernels busy {busy:.1f} us ({100 * busy / span:.1f} %); "
      f"gaps: mean {sum(allgaps) / max(len(allgaps), 1):.2f} us, median {sorted(allgaps)[len(allgaps) // 2]:.2f} us, "
      f"p90 {sorted(allgaps)[int(0.9 * len(allgaps))]:.2f} us")
