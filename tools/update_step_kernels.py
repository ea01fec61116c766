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
# the update phase of one round: period between consecutive Adam launches, and what runs between the round's last
# encoder kernel and its first update step (bootstrap values, GAE, advantage normalisation, sampler)
ends = [int(rows[i]["End_Timestamp"]) for i in idx]
per = [(ends[i + 1] - ends[i]) / 1e3 for i in range(len(ends) - 1)]
print("Adam-to-Adam periods (us):", " ".join("%.0f" % p for p in per[-24:]))
if len(sys.argv) > 2:                                  # gap analysis of the last round
    last8 = idx[-8:]
    first = last8[0]
    # walk back from the first Adam of the round to the previous round's last Adam
    prev = idx[-9] if len(idx) >= 9 else 0
    t0 = int(rows[prev]["End_Timestamp"])
    busy, last_end = 0.0, t0
    names = OrderedDict()
    for r in rows[prev + 1:first + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += (e - s) / 1e3
        d = names.setdefault(r["Kernel_Name"][:60], [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
    print("between the previous round's last Adam and this round's first: %.0f us wall, %.0f us of kernels" % ((int(rows[first]["End_Timestamp"]) - t0) / 1e3, busy))
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:12]:
        print("   %4d  %9.1f us  %s" % (c, t, n))
    # idle time inside the update phase of the round
    a0, a1 = first, last8[-1]
    lo = a0
    while lo > 0 and not rows[lo - 1]["Kernel_Name"].startswith("adam_dev") and (int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])) < 200000:
        lo -= 1
    tot = (int(rows[a1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
    kb = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[lo:a1 + 1])
    gaps = sorted(((int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3, rows[i]["Kernel_Name"][:40], rows[i + 1]["Kernel_Name"][:40]) for i in range(lo, a1))
    print("contiguous kernel run ending at the round's last Adam: %.0f us wall, %.0f us of kernels, largest gaps:" % (tot, kb))
    for g_, x, y in gaps[-8:]:
        print("   %8.1f us  after %s  before %s" % (g_, x, y))
