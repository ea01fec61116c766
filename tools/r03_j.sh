#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
timeout 600 python tools/enc_kernel_times.py --dtype bf16 --frames 2048 --passes 4 > $O/enc_bf16_2048.txt 2>&1; cat $O/enc_bf16_2048.txt | cut -c1-170
timeout 600 python tools/enc_kernel_times.py --dtype f32 --frames 1024 --passes 3 > $O/enc_f32_1024.txt 2>&1; cat $O/enc_f32_1024.txt | cut -c1-170
OUT=$O/act_trace; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/act_latency.py > $OUT/out.txt 2>&1
python3 - <<'PY'
import csv, glob
f=sorted(glob.glob("gpurun_out/r03j/act_trace/**/*kernel_stats.csv", recursive=True))
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
    print("act trace: %d kernel launches, %.1f ms total device time -> %.1f us per launch" % (calls, tot/1e6, tot/calls/1e3))
    for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:14]:
        print("  %6d calls %9.1f us avg  %5.1f%%  %s" % (int(r["Calls"]), float(r["AverageNs"])/1e3, float(r["Percentage"]), r["Name"][:80]))
PY
find $O -name "*kernel_trace.csv" -delete
