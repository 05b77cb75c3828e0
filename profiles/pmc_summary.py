#!/usr/bin/env python3
"""Print per-kernel averages of every counter in a rocprofv3 (rocpd sqlite) PMC run:  pmc_summary.py <db> [kernel substring]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
pat = "%" + (sys.argv[2] if len(sys.argv) > 2 else "") + "%"
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                   "group by kernel_name, counter_name order by kernel_name, counter_name", (pat,)).fetchall()
for k, c, v, n in rows:
    print("%-40s %-28s %16.1f  (n=%d)" % (k.split("(")[0][:40], c, v, n))
