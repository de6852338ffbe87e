#!/usr/bin/env python3
"""Dispatch timeline of the last N kernel launches in a rocprofv3 --kernel-trace rocpd database.
usage: python tools/rocpd_timeline.py <results.db> [N]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = con.execute("select * from kernels limit 1")
cols = [d[0] for d in cur.description]
s, e = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = con.execute(f"select name, {s}, {e} from kernels order by {s} desc limit {n}").fetchall()[::-1]
t0 = rows[0][1]
prev_end = t0
for name, a, b in rows:
    print(f"{(a - t0) / 1e3:10.1f} us  +gap {(a - prev_end) / 1e3:7.1f}  dur {(b - a) / 1e3:8.1f} us  {name[:70]}")
    prev_end = b
