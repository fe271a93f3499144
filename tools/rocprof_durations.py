#!/usr/bin/env python3
"""Durations (us) of consecutive launches of one kernel from a rocprofv3 kernel trace (rocpd .db), with the gap to the
previous launch: usage: rocprof_durations.py <results.db> [kernel substring] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "sd_demod_kernel"
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
rows = db.execute("select start, end from kernels where name like ? order by start", (f"%{pat}%",)).fetchall()
rows = rows[-cnt:]
prev_end = None
out = []
for s, e in rows:
    out.append(f"{(e - s) / 1000.0:.1f}" + (f"(+{(s - prev_end) / 1000.0:.1f})" if prev_end is not None else ""))
    prev_end = e
print(" ".join(out))
d = sorted((e - s) / 1000.0 for s, e in rows)
print("n", len(d), "min", d[0], "median", d[len(d) // 2], "p90", d[int(len(d) * 0.9)], "max", d[-1], "mean", sum(d) / len(d))
