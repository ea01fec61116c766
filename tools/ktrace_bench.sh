cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/ktrace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline $* > $OUT/trace.json 2> $OUT/trace.err
f=$(find $OUT -name "*kernel_stats.csv" | head -1); head -22 $f | cut -c1-150
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
