#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
timeout 300 python tools/lstm_trace.py > $O/lstm_trace.txt 2>&1; cat $O/lstm_trace.txt
timeout 900 python -m pytest tests/test_learner_gpu.py tests/test_timed_shapes_gpu.py tests/test_encoder_gpu.py -q -x > $O/t_learner.log 2>&1; tail -5 $O/t_learner.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2500 $O/bench.json
