#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03n; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "lstm" > $O/t_lstm.log 2>&1; tail -8 $O/t_lstm.log
for rt in 1 2; do echo "CADRE_LSTM_RT=$rt"; CADRE_LSTM_RT=$rt timeout 200 python tools/lstm_step_bench.py 2>&1 | grep "B=" | tee -a $O/step_bench.txt; done
timeout 300 python tools/lstm_trace.py > $O/lstm_trace.txt 2>&1; grep -E "launch|MFMA|issue|cell|wall" $O/lstm_trace.txt
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_timed_shapes_gpu.py -q -x > $O/t_learner.log 2>&1; tail -5 $O/t_learner.log
for cfg in C2 C3; do
  OUT=$O/ktrace_$cfg; rm -rf $OUT; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --config $cfg --steps 2 --warmup 2 --no-cpu-baseline --no-peaks --no-c3 > $OUT/trace.json 2> $OUT/trace.err
  python3 tools/update_step_kernels.py $OUT/trace > $OUT/step_census.txt 2>&1
  find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
  head -8 $OUT/step_census.txt
done
