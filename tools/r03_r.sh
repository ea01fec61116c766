#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03r; mkdir -p $O
for r in 0 1; do
timeout 200 python tools/ring_trace.py --mode 1 --dtype bf16 --shape 2048 72 72 64 64 --resid $r >> $O/trace_l1.txt 2>&1
timeout 200 python tools/ring_trace.py --mode 3 --dtype bf16 --shape 2048 72 72 64 64 --resid $r >> $O/trace_l1.txt 2>&1
done
timeout 200 python tools/ring_trace.py --mode 1 --dtype bf16 --shape 2048 36 36 128 128 --resid 1 >> $O/trace_l1.txt 2>&1
cat $O/trace_l1.txt
