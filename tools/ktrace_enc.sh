#!/bin/bash
# rocprofv3 kernel trace + stats of one encoder timing run: tools/ktrace_enc.sh f32|bf16 [extra args]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
DT=${1:-f32}; shift
OUT=gpurun_out/ktrace_enc_$DT
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/enc_kernel_times.py --dtype $DT --passes 2 "$@" > $OUT/run.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/trace
cut -d, -f1-4 $OUT/kernel_stats.csv | cut -c1-150 | head -${LINES_OUT:-14}
