#!/usr/bin/env python3
"""Averages rocprofv3 counter_collection CSVs per kernel: tools/pmc_summary.py <dir> [filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "k_prod|k_cprod|k_counts"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        if not re.search(flt, k):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*_kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        if re.search(flt, k):
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k in acc:
    print("== %s  (avg %.3f ms over %d dispatches, profiled)" % (k, sum(dur[k]) / max(len(dur[k]), 1), len(dur[k])))
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
