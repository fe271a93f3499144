#!/usr/bin/env python3
"""Summarise rocprofv3 output (rocpd sqlite .db) for this repo's kernels into a small CSV.

usage: rocprof_summary.py <results.db> [<results.db> ...] > profiles/<name>.csv
Prints per kernel: calls, average/min/max duration (us), and the average of every PMC counter
collected in that pass.  FETCH_SIZE/WRITE_SIZE are reported raw (KiB); the gfx950 x2 correction
for wide coalesced reads (MI355X_MICROARCH.md, HBM section) is applied in an extra column.
"""
import sqlite3
import sys


def main():
    print("db,kernel,calls,avg_us,min_us,max_us,counter,counter_avg,corrected_bytes")
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        cur = db.cursor()
        rows = cur.execute("select name, count(*), avg(end-start)/1000.0, min(end-start)/1000.0, max(end-start)/1000.0 "
                           "from kernels where name like '%sd_%' group by name").fetchall()
        pmc = {}
        try:
            for k, c, v in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                       "where kernel_name like '%sd_%' group by kernel_name, counter_name"):
                pmc.setdefault(k, []).append((c, v))
        except sqlite3.OperationalError:
            pass
        for name, calls, avg, mn, mx in rows:
            short = name.split("(")[0].replace("void ", "")
            cs = pmc.get(name) or [("", "")]
            for c, v in cs:
                corr = ""
                if c == "FETCH_SIZE":
                    corr = f"{v * 1024 * 2:.0f}"
                elif c == "WRITE_SIZE":
                    corr = f"{v * 1024:.0f}"
                vs = f"{v:.1f}" if v != "" else ""
                print(f"{path.split('/')[-2]},{short},{calls},{avg:.2f},{mn:.2f},{mx:.2f},{c},{vs},{corr}")


if __name__ == "__main__":
    main()
