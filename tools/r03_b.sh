#!/bin/bash
# fused LSTM step kernels: unit tests, learner parity, then the full suite and the bench + census
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "lstm_step" > $O/t_step.log 2>&1; tail -15 $O/t_step.log
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_timed_shapes_gpu.py -x -q > $O/t_learner.log 2>&1; tail -8 $O/t_learner.log
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03b/bench.json"))
print("C2", d["value"], d["ms_per_step"], d["t_update_ms"], d["update_roofline"]["ms_per_step"], d["update_roofline"]["kernels_per_update"])
c=d["c3"]; print("C3", c["value"], c["ms_per_step"], c["t_update_ms"], c["update_roofline"]["ms_per_step"])
PY
for cfg in C2 C3; do
  OUT=$O/ktrace_$cfg; rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline --no-peaks --no-c3 > $OUT/trace.json 2> $OUT/trace.err
  python3 tools/update_step_kernels.py $OUT/trace > $OUT/step_census.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  head -30 $OUT/step_census.txt
done
