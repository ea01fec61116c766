#!/bin/bash
# rocprofv3 kernel trace + stats of a C3 (bf16 encoder) bench run; summary -> gpurun_out/ktrace_c3/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/ktrace_c3
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
head -40 $OUT/kernel_stats.csv
cat $OUT/bench.json
