#!/usr/bin/env python3
"""Kernel census of ONE PPO minibatch step from a rocprofv3 kernel trace: everything between two
consecutive adam_dev_kernel launches (name, count, total us)."""
import csv
import glob
import sys
from collections import OrderedDict

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_dev_kernel")]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
agg = OrderedDict()
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"][:70]
    d = agg.setdefault(n, [0, 0.0])
    d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3
print("kernels in one step: %d, sum of durations %.1f us, wall span %.1f us" % (b - a, sum(v[1] for v in agg.values()), span))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4d  %8.1f us  %s" % (c, t, n))
