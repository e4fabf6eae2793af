#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd sqlite database (kernel-trace) like `--stats` does:
per kernel calls / total / avg / min / max duration and % of GPU kernel time."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-64s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "%"))
for n, c, t, a, mn, mx in rows:
    n = re.sub(r"\(.*", "", n)
    print("%-64s %6d %12.3f %12.4f %12.4f %12.4f %6.2f" % (n[:64], c, t / 1e6, a / 1e6, mn / 1e6, mx / 1e6, 100.0 * t / tot))
