#!/usr/bin/env python3
"""Per-solve summary of a rocprofv3 --kernel-trace CSV of bench.py: busy time by kernel and idle gaps.
usage: tools/trace_gaps.py <kernel_trace.csv>"""
import csv, re, sys
from collections import defaultdict
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void bsn::", "").replace("bsn::", "")))
rows.sort()
# a solve starts at k_random
solves, cur = [], None
for st, en, name in rows:
    if name.startswith("k_random"):
        cur = []
        solves.append(cur)
    if cur is not None:
        cur.append((st, en, name))
for i, s in enumerate(solves):
    t0, t1 = s[0][0], s[-1][1]
    busy = defaultdict(float)
    gap_after = defaultdict(float)
    for j, (st, en, name) in enumerate(s):
        short = re.sub(r"<.*", "", name)
        busy[short] += (en - st) / 1e6
        if j + 1 < len(s):
            g = (s[j + 1][0] - en) / 1e6
            if g > 0:
                gap_after[short] += g
    tot_busy = sum(busy.values())
    print("solve %d: span %.1f ms, busy %.1f ms, idle %.1f ms, %d kernels" % (i, (t1 - t0) / 1e6, tot_busy, (t1 - t0) / 1e6 - tot_busy, len(s)))
    print("   busy: " + ", ".join("%s %.2f" % (k, v) for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:8]))
    print("   idle after: " + ", ".join("%s %.2f" % (k, v) for k, v in sorted(gap_after.items(), key=lambda kv: -kv[1])[:8]))
