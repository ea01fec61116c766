#!/usr/bin/env python3
"""Where do the __amd_rocclr_copyBuffer launches of a learner round sit?  Prints, for the last round of a rocprofv3 kernel trace, every
copy with the kernels right before and after it.   python tools/dbg/copy_sites.py <trace dir>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "stem_pool_kernel" in r["Kernel_Name"]]
a = stems[-2] if len(stems) > 1 else 0
b = stems[-1]
seg = rows[a:b]
print("kernels in the round:", len(seg), " copies:", sum(r["Kernel_Name"].startswith("__amd_rocclr_copyBuffer") for r in seg))
from collections import Counter
ctx = Counter()
for i, r in enumerate(seg):
    if r["Kernel_Name"].startswith("__amd_rocclr_copyBuffer"):
        p = seg[i - 1]["Kernel_Name"][:48] if i else "-"
        n = seg[i + 1]["Kernel_Name"][:48] if i + 1 < len(seg) else "-"
        ctx[(p, n)] += 1
for (p, n), c in ctx.most_common():
    print("%4d  after %-50s before %s" % (c, p, n))
