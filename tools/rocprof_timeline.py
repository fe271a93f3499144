#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 kernel trace (rocpd .db): for the last steps of the run, every kernel's start and
end relative to the first kernel of its step (a step = the kernels between two launches of the anchor kernel).
usage: rocprof_timeline.py <results.db> [anchor substring] [steps]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "sd_demod_kernel"
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = db.execute("select name, start, end, stream_id, queue_id from kernels where name like '%sd_%' order by start").fetchall()
rows = rows[len(rows) // 2:]                      # steady state
steps, cur = [], []
for r in rows:
    if anchor in r[0] and cur and any(anchor in x[0] for x in cur) and r[1] > max(x[2] for x in cur if anchor in x[0]):
        steps.append(cur)
        cur = []
    cur.append(r)
acc = {}
for st in steps[-(nsteps + 40):]:
    t0 = min(x[1] for x in st)
    for name, s, e, sid, qid in st:
        short = name.split("(")[0].replace("void ", "")
        a = acc.setdefault(short, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (s - t0) / 1000.0; a[2] += (e - t0) / 1000.0
print("kernel,calls,avg_start_us,avg_end_us (relative to the first kernel start of the step)")
for k, (n, s, e) in sorted(acc.items(), key=lambda kv: kv[1][1] / kv[1][0]):
    print(f"{k},{n},{s / n:.1f},{e / n:.1f}")
for st in steps[-nsteps:]:
    t0 = min(x[1] for x in st)
    print("--- step")
    for name, s, e, sid, qid in st:
        print(f"  {name.split('(')[0].replace('void ', ''):44s} q{qid} {(s - t0) / 1000.0:8.1f} -> {(e - t0) / 1000.0:8.1f} us")
