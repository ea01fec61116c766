#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
joined with the kernel trace of the same pass:
    util = SQ_VALU_MFMA_BUSY_CYCLES / (duration_ns * f_clk * 1024 SIMDs)
with f_clk estimated as GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md 'DVFS give-back')."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
cnt = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[(r["Dispatch_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0])][r["Counter_Name"]] += float(r["Counter_Value"])
dur = {}
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for (did, name), c in cnt.items():
    if did not in dur or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    a = agg[name]
    a[0] += c["SQ_VALU_MFMA_BUSY_CYCLES"]; a[1] += c.get("GRBM_GUI_ACTIVE", 0.0); a[2] += dur[did]; a[3] += 1
print("== MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (time x clock x 1024 SIMDs)); clock = GRBM_GUI_ACTIVE/8/time")
for name, (busy, gui, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:8]:
    if busy <= 0 or ns <= 0:
        continue
    clk = gui / 8.0 / ns if gui > 0 else 2.4          # GHz
    print("  %-40s dispatches %4d  avg %8.1f us  clock %.2f GHz  MFMA busy %.1f %% of SIMD-cycles" %
          (name, n, ns / n / 1e3, clk, 100.0 * busy / (ns * clk * 1024)))
