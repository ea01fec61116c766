cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/ktrace2; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-peaks --no-c3 "$@" > $OUT/trace.json 2> $OUT/trace.err
python3 tools/update_step_kernels.py $OUT/trace
find $OUT -name "*kernel_trace.csv" -delete
