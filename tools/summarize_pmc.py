#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC counter_collection) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(n):
    n = n.replace("void ", "")
    return n[:70]


for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:40]:
        print("  %-72s calls %6s  total %10.3f ms  avg %9.1f us  %5s%%" % (
            short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    print("== counters:", f)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:14]:
        n = max(cnt[(k, c)] for c in acc[k])
        print("  %-72s dispatches %d" % (k, n))
        for c, v in sorted(acc[k].items()):
            print("      %-28s %.4g per dispatch" % (c, v / n))
