#!/bin/bash
# round-3 first GPU pass: GPU test suite, bench line, kernel census of one update step at B=64 (C2) and B=256 (C3)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/tests.log
tail -5 gpurun_out/r03a/tests.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
for cfg in C2 C3; do
  OUT=gpurun_out/r03a/ktrace_$cfg; rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline --no-peaks --no-c3 > $OUT/trace.json 2> $OUT/trace.err
  python3 tools/update_step_kernels.py $OUT/trace > $OUT/step_census.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  head -40 $OUT/step_census.txt
done
