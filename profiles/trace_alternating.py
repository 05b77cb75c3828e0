#!/usr/bin/env python3
"""Average duration of every k-th dispatch of one kernel in a rocprofv3 kernel trace (rocpd sqlite):
trace_alternating.py <db> <kernel substring> <period>   -- e.g. seg_kernel 2 separates the two chain segments of bench.py"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select duration from kernels where name like ? order by start", ("%" + sys.argv[2] + "%",)).fetchall()
period = int(sys.argv[3])
d = [r[0] / 1e3 for r in rows]
d = d[len(d) % period:]
for k in range(period):
    sel = d[k::period][-20:]
    print("%s dispatch %d mod %d: n=%d avg %.1f us min %.1f max %.1f" % (sys.argv[2], k, period, len(sel), sum(sel) / len(sel), min(sel), max(sel)))
